// CPU sanitizer driver (test infrastructure): the oracle, the host-side library and the PPM writers
// compiled into ONE executable with -fsanitize=address,undefined and exercised through their C
// interfaces.  SURVEY.md §5 asks for this because the reference itself carries latent UB on this
// path (an uninitialised sphere slot, hittable.h:34; the writer's int(256*x), main.cu:374).
// Built and run by tests/test_sanitizers.py (`make -C oracle asan`); exits non-zero on any finding
// (-fno-sanitize-recover) or on a self-check mismatch.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../oracle/rtiow_oracle.cpp"                         // the oracle, as a translation unit
#include "../../raytracingincuda_amd/csrc/host/rtiow_host.cpp"   // the product's host library

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); ++fails; } } while (0)

template <class T> static void scenes_and_cameras(int prec) {
    for (int scene = 1; scene <= 3; ++scene) {
        const int slots = oracle_scene_slots(scene);
        CHECK(slots == rtiow_host_scene_slots(scene));
        std::vector<T> cr(4 * slots), af(4 * slots), ri(slots), cr2(4 * slots), af2(4 * slots), ri2(slots);
        std::vector<int> ty(slots), va(slots), ty2(slots), va2(slots);
        CHECK(oracle_build_scene(scene, prec, cr.data(), af.data(), ri.data(), ty.data(), va.data()) == slots);
        CHECK(rtiow_host_build_scene(scene, prec, cr2.data(), af2.data(), ri2.data(), ty2.data(), va2.data()) == slots);
        for (int i = 0; i < slots; ++i) {
            CHECK(va[i] == va2[i]);
            if (!va[i]) continue;                                   // skipped grid cell: contents are unspecified
            CHECK(std::memcmp(&cr[4 * i], &cr2[4 * i], 4 * sizeof(T)) == 0);
            CHECK(std::memcmp(&af[4 * i], &af2[4 * i], 4 * sizeof(T)) == 0);
            CHECK(ty[i] == ty2[i]);
        }
    }
    struct Geo { int W, H, S, B; } geos[] = {{320, 192, 10, 25}, {1280, 720, 100, 50}, {1920, 1080, 500, 50}, {1, 1, 1, 1}, {7, 3, 2, 0}};
    for (const Geo& g : geos) {
        int ints[4]; T flat[20];
        CHECK(oracle_camera_init(prec, g.W, g.H, g.S, g.B, ints, flat) == 0);
        CHECK(ints[0] == g.W && ints[1] == g.H);
        if (prec == 32) { rtiow_camera_f32 c; CHECK(rtiow_host_camera(32, g.W, g.H, g.S, g.B, &c) == 0); CHECK(c.img_width == g.W); }
        else { rtiow_camera_f64 c; CHECK(rtiow_host_camera(64, g.W, g.H, g.S, g.B, &c) == 0); CHECK(c.img_height == g.H); }
    }
}

template <class T> static void renders_and_writers(int prec, const char* tmpdir) {
    const int W = 24, H = 14, S = 3, B = 12;
    for (int scene = 1; scene <= 3; ++scene) {
        const int slots = oracle_scene_slots(scene);
        std::vector<T> cr(4 * slots), af(4 * slots), ri(slots);
        std::vector<int> ty(slots), va(slots);
        oracle_build_scene(scene, prec, cr.data(), af.data(), ri.data(), ty.data(), va.data());
        int n = 0;                                                  // compact, as the tests do
        for (int i = 0; i < slots; ++i) if (va[i]) { std::memmove(&cr[4 * n], &cr[4 * i], 4 * sizeof(T)); std::memmove(&af[4 * n], &af[4 * i], 4 * sizeof(T)); ri[n] = ri[i]; ty[n] = ty[i]; ++n; }
        int ints[4]; T flat[20];
        oracle_camera_init(prec, W, H, S, B, ints, flat);
        std::vector<T> img((size_t)W * H * 3, (T)-1);
        unsigned long long st[4] = {0, 0, 0, 0};
        for (int form = 0; form < 2; ++form)
            for (int sky = 0; sky < 2; ++sky)
                CHECK(oracle_render_modes(prec, n, cr.data(), af.data(), ri.data(), ty.data(), ints, flat, 1227ull, 0, H, form, sky, img.data(), st, nullptr) == 0);
        CHECK(st[0] == (unsigned long long)W * H * S && st[1] >= st[0]);
        for (T v : img) CHECK(v >= 0 && v <= (T)1.0001 && std::isfinite((double)v));
        // the writers, P3 into memory and both formats to disk; values outside [0,1) and NaN included
        img[0] = (T)-0.25; img[1] = (T)7; img[2] = std::nan("");
        size_t len = 0;
        std::vector<char> text((size_t)W * H * 12 + 64);
        CHECK(rtiow_host_format_ppm(prec, W, H, img.data(), text.data(), text.size(), &len) == 0);
        CHECK(len > 0 && len <= text.size() && std::memcmp(text.data(), "P3\n", 3) == 0);
        CHECK(rtiow_host_format_ppm(prec, W, H, img.data(), text.data(), 8, &len) != 0);       // buffer too small must be refused
        char name[256];
        CHECK(rtiow_host_ppm_filename(prec, scene, W, H, S, B, 8, name, sizeof name) == 0);
        const std::string p3 = std::string(tmpdir) + "/" + name, p6 = p3 + ".p6";
        CHECK(rtiow_host_write_ppm(p3.c_str(), prec, W, H, img.data()) == 0);
        CHECK(rtiow_host_write_ppm_binary(p6.c_str(), prec, W, H, img.data()) == 0);
        CHECK(rtiow_host_write_ppm((std::string(tmpdir) + "/no/such/dir/x.ppm").c_str(), prec, W, H, img.data()) != 0);
    }
    // the threaded writer (65536 pixels or more): ranges formatted concurrently from tight buffers, NaN levels in the
    // first and the last range force the long form
    for (int prec : {32, 64}) {
        const int W = 311, H = 277;
        std::vector<double> img64((size_t)W * H * 3);
        std::vector<float> img32(img64.size());
        for (size_t k = 0; k < img64.size(); ++k) { img64[k] = (double)((k * 2654435761u) % 1300u) / 1000.0 - 0.15; img32[k] = (float)img64[k]; }
        img64[1] = img64[img64.size() - 2] = std::nan(""); img32[1] = img32[img32.size() - 2] = std::nanf("");
        const void* data = prec == 32 ? (const void*)img32.data() : (const void*)img64.data();
        size_t len = 0;
        CHECK(rtiow_host_format_ppm(prec, W, H, data, nullptr, 0, &len) == 0);
        std::vector<char> text(len);
        CHECK(rtiow_host_format_ppm(prec, W, H, data, text.data(), text.size(), &len) == 0 && len == text.size());
        CHECK(text.back() == '\n' && std::memcmp(text.data(), "P3\n311 277\n255\n", 15) == 0);
        CHECK(rtiow_host_write_ppm((std::string(tmpdir) + "/threaded.ppm").c_str(), prec, W, H, data) == 0);
    }
    // serial semantics (the CPU baseline path), small frame
    unsigned long long st[4];
    std::vector<char> p3(1 << 20);
    CHECK(oracle_render_serial(3, 32, 18, 2, 8, p3.data(), (long long)p3.size(), st) > 0);
}

static void rng_and_shards() {
    unsigned int s[6];
    for (unsigned long long seq : {0ull, 1ull, 61439ull, 2073599ull, 0xffffffffull}) { oracle_xorwow_init(1227ull, seq, 0ull, 0, s); oracle_xorwow_init(1227ull, seq, 5ull, 1, s); }
    int r[16]; oracle_glibc_rand(16, r);
    CHECK(r[0] == 1804289383);
    for (int H : {1, 7, 8, 1080}) for (int N : {1, 2, 3, 8}) for (int strip : {1, 2, 8}) {
        int total = 0;
        std::vector<char> seen(H, 0);
        for (int rank = 0; rank < N; ++rank) {
            std::vector<int32_t> rows(H);
            const int k = rtiow_host_shard_rows(H, rank, N, strip, rows.data());
            CHECK(k >= 0);
            for (int i = 0; i < k; ++i) { CHECK(rows[i] >= 0 && rows[i] < H && !seen[rows[i]]); seen[rows[i]] = 1; }
            total += k;
        }
        CHECK(total == H);
    }
}

int main(int argc, char** argv) {
    const char* tmpdir = argc > 1 ? argv[1] : "/tmp";
    oracle_set_threads(2);
    scenes_and_cameras<float>(32); scenes_and_cameras<double>(64);
    renders_and_writers<float>(32, tmpdir); renders_and_writers<double>(64, tmpdir);
    rng_and_shards();
    std::printf("sanitize_main: %d self-check failure(s)\n", fails);
    return fails ? 1 : 0;
}
