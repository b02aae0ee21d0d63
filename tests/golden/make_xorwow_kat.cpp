// Generator of tests/golden/xorwow_rocrand_kat.json: known answers from rocRAND's HOST XORWOW engine
// (/opt/rocm/include/rocrand/rocrand_xorwow.h -- a ROCm library, not the reference).  rocRAND shares
// the engine, the 2^67 subsequence stride and the offset semantics with cuRAND but salts the seed with
// different constants, so these answers pin the oracle's engine + jump matrices via salt=1.
// Build+run: hipcc -O1 -o kat make_xorwow_kat.cpp && ./kat > xorwow_rocrand_kat.json
#include <cstdio>
#include <rocrand/rocrand_xorwow.h>
int main() {
    unsigned long long seeds[] = {1227ULL, 0ULL, 0x123456789abcdefULL};
    unsigned long long seqs[] = {0, 1, 2, 3, 7, 255, 256, 61439, 65536, 921599, 2073599, 4194303, 16777216ULL, 4000000000ULL};
    unsigned long long offs[] = {0, 1, 5, 1000, 123456789ULL};
    printf("[\n");
    bool first = true;
    for (auto seed : seeds) for (auto seq : seqs) for (auto off : offs) {
        if (seed != 1227ULL && (seq > 70000 || off > 1000)) continue;
        rocrand_state_xorwow st;
        rocrand_init(seed, seq, off, &st);
        unsigned int v[4];
        for (int i = 0; i < 4; ++i) v[i] = rocrand(&st);
        printf("%s{\"seed\": %llu, \"subsequence\": %llu, \"offset\": %llu, \"u32\": [%u, %u, %u, %u]}", first ? "" : ",\n", seed, seq, off, v[0], v[1], v[2], v[3]);
        first = false;
    }
    printf("\n]\n");
}
