// valu_peak.hip -- measures the vector-ALU issue rates that bound the render kernel on this
// GPU (the roofline denominator in bench.py / DESIGN.md): v_fma_f32, v_pk_fma_f32, v_fma_f64,
// and the v_mul_f32+v_add_f32 pair, as chip-wide TFLOP/s, at several waves per SIMD.
// Prints one JSON object.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;   // independent accumulators per lane

template <int KIND>
__global__ void __launch_bounds__(256) peak_kernel(float* out, float seed) {
    const float a = seed + threadIdx.x * 1e-7f, b = 0.999f;
    float acc[UNROLL];
    f2 acc2[UNROLL];
    double accd[UNROLL];
    for (int k = 0; k < UNROLL; ++k) { acc[k] = a + k; acc2[k] = {a + k, a - k}; accd[k] = a + k; }
    const f2 a2 = {a, a}, b2 = {b, b};
    const double ad = a, bd = b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[k]) : "v"(a2), "v"(b2));
            if (KIND == 2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(accd[k]) : "v"(ad), "v"(bd));
            if (KIND == 3) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(acc[k]) : "v"(b), "v"(a));
            if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(acc2[k]) : "v"(b2), "v"(a2));
            if (KIND == 5) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[k]) : "v"(a), "v"(b));
            if (KIND == 6) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc[k]) : "v"(a));
            if (KIND == 7) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "s"(seed));
            if (KIND == 8) asm volatile("v_sub_f32_e32 %0, %1, %0\n\tv_mul_f32_e32 %0, %0, %0\n\tv_fmac_f32_e32 %0, %1, %1" : "+v"(acc[k]) : "v"(a));
        }
    }
    float s = 0;
    for (int k = 0; k < UNROLL; ++k) s += acc[k] + acc2[k].x + acc2[k].y + (float)accd[k];
    if (s == 12345.678f) out[0] = s;   // never true; keeps the chains alive
}

template <int KIND>
double run(int blocks_per_cu, int cus, double flops_per_inst_lane) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = cus * blocks_per_cu;
    peak_kernel<KIND><<<blocks, 256>>>(d, 1.0f);
    hipDeviceSynchronize();
    double best = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        peak_kernel<KIND><<<blocks, 256>>>(d, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)ITERS * UNROLL * ((KIND == 3 || KIND == 4) ? 2 : (KIND == 8 ? 3 : 1));
        const double tf = insts * flops_per_inst_lane * 256.0 * blocks / (ms * 1e-3) / 1e12;
        if (tf > best) best = tf;
    }
    hipFree(d);
    return best;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d", p.gcnArchName, cus, p.clockRate / 1000);
    for (int bpc : {1, 2, 4, 8}) {   // 256-thread blocks per CU = waves per SIMD
        printf(", \"waves_per_simd_%d\": {\"v_fma_f32_tflops\": %.2f, \"v_pk_fma_f32_tflops\": %.2f, \"v_fma_f64_tflops\": %.2f, "
               "\"v_mul_add_f32_tflops\": %.2f, \"v_pk_mul_add_f32_tflops\": %.2f, \"v_fmac_f32_e32_tflops\": %.2f, "
               "\"v_fma_f32_2src_tflops\": %.2f, \"v_fma_f32_sgpr_tflops\": %.2f, \"sub_mul_fmac_mix_Tinst\": %.2f}",
               bpc, run<0>(bpc, cus, 2), run<1>(bpc, cus, 4), run<2>(bpc, cus, 2), run<3>(bpc, cus, 1), run<4>(bpc, cus, 2),
               run<5>(bpc, cus, 2), run<6>(bpc, cus, 2), run<7>(bpc, cus, 2), run<8>(bpc, cus, 1));
    }
    printf("}\n");
    return 0;
}
