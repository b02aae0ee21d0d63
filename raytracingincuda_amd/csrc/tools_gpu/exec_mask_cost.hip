// exec_mask_cost.hip -- does a wave64 vector instruction cost less issue time when only some lanes are enabled?
//
// The render path is bound by vector issue and its divergent blocks (rejection rounds, grid steps, IEEE tails) run
// with ~11 of 64 lanes enabled (profiles/archive/r02_path_stats.json; SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU says a third
// of the lanes on average).  A SIMD-32 executes a wave64 instruction in two passes of 32 lanes; if the hardware
// skipped a pass whose 32 lanes are all disabled, packing the live lanes of a divergent block into one half of the
// wave would halve its cost.  This probe times 16 independent v_fma_f32 / v_xor_b32 / v_pk_fma_f32 chains per lane
// under different EXEC masks at 1, 2 and 5 waves per SIMD and prints cycles per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o exec_mask_cost exec_mask_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2048, UNROLL = 16;

// MASK: 0 all 64 lanes, 1 lanes 0-31, 2 lanes 32-63, 3 lane 0 only, 4 even lanes, 5 lanes 0-15, 6 lanes 0-10 and 32-42
template <int MASK> __device__ __forceinline__ bool lane_on(int lane) {
    switch (MASK) {
        case 0: return true;
        case 1: return lane < 32;
        case 2: return lane >= 32;
        case 3: return lane == 0;
        case 4: return (lane & 1) == 0;
        case 5: return lane < 16;
        default: return (lane & 31) < 11;
    }
}

template <int MASK, int OP>
__global__ void __launch_bounds__(256) mask_kernel(float* out, float seed) {
    const int lane = threadIdx.x & 63;
    const float a = seed + threadIdx.x * 1e-7f, b = 0.999f;
    float acc[UNROLL];
    f2 acc2[UNROLL];
    for (int k = 0; k < UNROLL; ++k) { acc[k] = a + k; acc2[k] = {a + k, a - k}; }
    const f2 a2 = {a, a}, b2 = {b, b};
    if (lane_on<MASK>(lane)) {
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
                if (OP == 1) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(acc[k]) : "v"(a));
                if (OP == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[k]) : "v"(a2), "v"(b2));
                if (OP == 3) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(acc[k]));
            }
        }
    }
    float s = 0;
    for (int k = 0; k < UNROLL; ++k) s += acc[k] + acc2[k].x + acc2[k].y;
    if (s == 12345.678f) out[0] = s;
}

template <int MASK, int OP>
double run(int blocks_per_cu, int cus, double clock_hz) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = cus * blocks_per_cu;
    mask_kernel<MASK, OP><<<blocks, 256>>>(d, 1.0f);
    hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mask_kernel<MASK, OP><<<blocks, 256>>>(d, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)ITERS * UNROLL * blocks_per_cu;      // each block = 4 waves = one per SIMD
        const double cyc = ms * 1e-3 * clock_hz / per_simd;
        if (cyc < best) best = cyc;
    }
    hipFree(d);
    return best;
}

template <int OP> void row(const char* name, int bpc, int cus, double hz, bool first) {
    printf("%s\"%s\": {\"all64\": %.2f, \"lanes0_31\": %.2f, \"lanes32_63\": %.2f, \"lane0\": %.2f, \"even\": %.2f, \"lanes0_15\": %.2f, \"11_per_half\": %.2f}",
           first ? "" : ", ", name, run<0, OP>(bpc, cus, hz), run<1, OP>(bpc, cus, hz), run<2, OP>(bpc, cus, hz), run<3, OP>(bpc, cus, hz),
           run<4, OP>(bpc, cus, hz), run<5, OP>(bpc, cus, hz), run<6, OP>(bpc, cus, hz));
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double hz = p.clockRate * 1e3;
    printf("{\"device\": \"%s\", \"unit\": \"nominal-clock cycles per wave-instruction per SIMD, by EXEC mask\"", p.gcnArchName);
    for (int bpc : {1, 2, 5}) {
        printf(", \"waves_per_simd_%d\": {", bpc);
        row<0>("v_fma_f32", bpc, cus, hz, true);
        row<1>("v_xor_b32", bpc, cus, hz, false);
        row<2>("v_pk_fma_f32", bpc, cus, hz, false);
        row<3>("v_sqrt_f32", bpc, cus, hz, false);
        printf("}");
    }
    printf("}\n");
    return 0;
}
