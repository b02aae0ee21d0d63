// valu_cost.hip -- issue cost (cycles per wave-instruction on one SIMD, all SIMDs busy) of the
// vector opcodes the render kernel is made of, at 1/2/4/8 waves per SIMD.  The kernel is bound
// by vector issue, so these are the weights of its instruction budget.
// Each opcode runs in 16 independent chains per lane; cycles = SIMD-cycles / wave-instructions at
// the clock the chip holds during the run (s_memtime is not used: wall time x nominal clock is
// reported together with the v_fma_f32 reference so ratios are clock-independent).
// Prints one JSON object.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int ITERS = 2048;
constexpr int UNROLL = 16;

#define OPS(X) \
    X(0,  v_fma_f32,        "v_fma_f32 %0, %2, %3, %0",                         1) \
    X(1,  v_mul_f32,        "v_mul_f32_e32 %0, %2, %0",                         1) \
    X(2,  v_add_f32,        "v_add_f32_e32 %0, %2, %0",                         1) \
    X(3,  v_fmac_f32,       "v_fmac_f32_e32 %0, %2, %3",                        1) \
    X(4,  v_pk_fma_f32,     "v_pk_fma_f32 %1, %4, %5, %1",                      1) \
    X(5,  v_pk_fma_f32_sgpr,"v_pk_fma_f32 %1, %4, %6, %1",                      1) \
    X(6,  v_pk_fma_f32_sq,  "v_pk_fma_f32 %1, %4, %4, %1",                      1) \
    X(7,  v_pk_mul_f32,     "v_pk_mul_f32 %1, %4, %1",                          1) \
    X(8,  v_pk_add_f32,     "v_pk_add_f32 %1, %4, %1",                          1) \
    X(9,  v_xor_b32,        "v_xor_b32_e32 %0, %2, %0",                         1) \
    X(10, v_lshlrev_b32,    "v_lshlrev_b32_e32 %0, 1, %0",                      1) \
    X(11, v_add_u32,        "v_add_u32_e32 %0, %2, %0",                         1) \
    X(12, v_add3_u32,       "v_add3_u32 %0, %2, %3, %0",                        1) \
    X(13, v_mov_b32,        "v_mov_b32_e32 %0, %2",                             1) \
    X(14, v_cndmask_b32,    "v_cndmask_b32_e32 %0, %2, %0, vcc",                1) \
    X(15, v_cvt_f32_u32,    "v_cvt_f32_u32_e32 %0, %0",                         1) \
    X(16, v_max3_f32,       "v_max3_f32 %0, %2, %3, %0",                        1) \
    X(17, v_max_f32,        "v_max_f32_e32 %0, %2, %0",                         1) \
    X(18, v_cmp_lt_f32,     "v_cmp_lt_f32_e32 vcc, %2, %0",                     1) \
    X(19, v_rcp_f32,        "v_rcp_f32_e32 %0, %0",                             1) \
    X(20, v_sqrt_f32,       "v_sqrt_f32_e32 %0, %0",                            1) \
    X(21, v_rsq_f32,        "v_rsq_f32_e32 %0, %0",                             1) \
    X(22, v_div_scale_f32,  "v_div_scale_f32 %0, vcc, %0, %2, %0",              1) \
    X(23, v_div_fmas_f32,   "v_div_fmas_f32 %0, %0, %2, %3",                    1) \
    X(24, v_div_fixup_f32,  "v_div_fixup_f32 %0, %0, %2, %3",                   1) \
    X(25, v_fma_f64,        "v_fma_f64 %7, %8, %8, %7",                         1) \
    X(26, v_mul_f64,        "v_mul_f64 %7, %8, %7",                             1) \
    X(27, v_add_f64,        "v_add_f64 %7, %8, %7",                             1) \
    X(28, v_sqrt_f64,       "v_sqrt_f64_e32 %7, %7",                            1) \
    X(29, v_rcp_f64,        "v_rcp_f64_e32 %7, %7",                             1) \
    X(30, v_lshl_add_u32,   "v_lshl_add_u32 %0, %0, 1, %2",                     1) \
    X(31, v_xad_u32,        "v_xad_u32 %0, %0, %2, %3",                         1) \
    X(32, v_and_or_b32,     "v_and_or_b32 %0, %0, %2, %3",                      1) \
    X(33, v_readlane,       "v_readlane_b32 s20, %0, 3",                        1) \
    X(34, v_mul_add_pair,   "v_mul_f32_e32 %0, %2, %0\n\tv_add_f32_e32 %0, %3, %0", 2) \
    X(35, pk_fma_then_xor,  "v_pk_fma_f32 %1, %4, %5, %1\n\tv_xor_b32_e32 %0, %2, %0", 2) \
    X(36, fma_then_xor,     "v_fma_f32 %0, %2, %3, %0\n\tv_xor_b32_e32 %0, %2, %0", 2) \
    X(37, v_cndmask_e64_sgpr, "v_cndmask_b32_e64 %0, %2, %0, s[22:23]",          1) \
    X(38, v_cndmask_e32_other, "v_cndmask_b32_e32 %0, %2, %3, vcc",              1) \
    X(39, cmp_then_cndmask,  "v_cmp_lt_f32_e32 vcc, %2, %0\n\tv_cndmask_b32_e32 %0, %2, %0, vcc", 2) \
    X(40, v_and_b32,        "v_and_b32_e32 %0, %2, %0",                          1) \
    X(41, v_or_b32,         "v_or_b32_e32 %0, %2, %0",                           1) \
    X(42, v_sub_f32,        "v_sub_f32_e32 %0, %2, %0",                          1) \
    X(43, v_lshrrev_b32,    "v_lshrrev_b32_e32 %0, 2, %0",                       1) \
    X(44, v_mul_u32_u24,    "v_mul_u32_u24_e32 %0, 16, %0",                      1) \
    X(45, v_min_f32,        "v_min_f32_e32 %0, %2, %0",                          1) \
    X(46, v_fma_f32_neg,    "v_fma_f32 %0, -%2, %3, %0",                         1) \
    X(47, v_mad_u32_u24,    "v_mad_u32_u24 %0, %0, 16, %2",                      1) \
    X(48, v_bfe_u32,        "v_bfe_u32 %0, %0, 2, 30",                           1) \
    X(49, v_alignbit_b32,   "v_alignbit_b32 %0, %2, %0, 2",                      1) \
    X(50, cmp_then_3_cndmask, "v_cmp_lt_f32_e32 vcc, %2, %0\n\tv_cndmask_b32_e32 %0, %2, %0, vcc\n\tv_cndmask_b32_e32 %0, %3, %0, vcc\n\tv_cndmask_b32_e32 %0, %2, %0, vcc", 4) \
    X(51, v_cndmask_e64_vcc, "v_cndmask_b32_e64 %0, %2, %0, vcc",                1) \
    X(52, salu_vcc_then_cndmask, "s_mov_b64 vcc, s[22:23]\n\ts_nop 4\n\tv_cndmask_b32_e32 %0, %2, %0, vcc", 1) \
    X(53, cmp_e64_then_cndmask_e64, "v_cmp_lt_f32_e64 s[22:23], %2, %0\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %2, %0, s[22:23]", 2) \
    X(54, v_bitop3_b32,     "v_bitop3_b32 %0, %2, %0, %3 bitop3:0x96",           1) \
    X(55, v_cndmask_same_src, "v_cndmask_b32_e32 %0, %0, %0, vcc",                1) \
    X(56, cmp_nop_cndmask_e32, "v_cmp_lt_f32_e32 vcc, %2, %0\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %2, %0, vcc", 2) \
    X(57, cmp_nop_cndmask_e64, "v_cmp_lt_f32_e32 vcc, %2, %0\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %2, %0, vcc", 2) \
    X(58, cmp_nop_3_cndmask_e32, "v_cmp_lt_f32_e32 vcc, %2, %0\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %2, %0, vcc\n\tv_cndmask_b32_e32 %0, %3, %0, vcc\n\tv_cndmask_b32_e32 %0, %2, %0, vcc", 4) \
    X(59, cmp_nop_3_cndmask_e64, "v_cmp_lt_f32_e32 vcc, %2, %0\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %2, %0, vcc\n\tv_cndmask_b32_e64 %0, %3, %0, vcc\n\tv_cndmask_b32_e64 %0, %2, %0, vcc", 4) \
    X(60, cmp_2xor_cndmask_e32, "v_cmp_lt_f32_e32 vcc, %2, %0\n\tv_xor_b32_e32 %0, %3, %0\n\tv_xor_b32_e32 %0, %2, %0\n\tv_cndmask_b32_e32 %0, %2, %0, vcc", 4) \
    X(61, cmp_2xor_cndmask_e64, "v_cmp_lt_f32_e32 vcc, %2, %0\n\tv_xor_b32_e32 %0, %3, %0\n\tv_xor_b32_e32 %0, %2, %0\n\tv_cndmask_b32_e64 %0, %2, %0, vcc", 4) \
    X(62, cmp_e32_alone_nop,  "v_cmp_lt_f32_e32 vcc, %2, %0\n\ts_nop 1\n\tv_xor_b32_e32 %0, %3, %0", 2) \
    X(63, v_add_u32_self,     "v_add_u32_e32 %0, %0, %0",                          1)

template <int KIND>
__global__ void __launch_bounds__(256) cost_kernel(float* out, float seed, f2 sg) {
    const float a = seed + threadIdx.x * 1e-7f, b = 0.999f;
    float acc[UNROLL];
    f2 acc2[UNROLL];
    double accd[UNROLL];
    for (int k = 0; k < UNROLL; ++k) { acc[k] = a + k; acc2[k] = {a + k, a - k}; accd[k] = 1.0 + 1e-3 * k + a * 1e-3; }
    const f2 a2 = {a, a}, b2 = {b, b};
    const double bd = 1.0000001;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
#define X(ID, NAME, TEXT, N) if (KIND == ID) asm volatile(TEXT : "+v"(acc[k]), "+v"(acc2[k]) : "v"(a), "v"(b), "v"(a2), "v"(b2), "s"(sg), "v"(accd[k]), "v"(bd) : "vcc", "s20", "s22", "s23");
            OPS(X)
#undef X
        }
    }
    float s = 0;
    for (int k = 0; k < UNROLL; ++k) s += acc[k] + acc2[k].x + acc2[k].y + (float)accd[k];
    if (s == 12345.678f) out[0] = s;
}

// v_fma_f64 & co write accd through a read-only "v" constraint above would be wrong: give the
// f64 kinds their own kernel with the accumulator as the read-write operand.
template <int KIND>
__global__ void __launch_bounds__(256) cost_kernel_f64(float* out, float seed) {
    double accd[UNROLL];
    const double bd = 1.0000001 + seed * 1e-9;
    for (int k = 0; k < UNROLL; ++k) accd[k] = 1.0 + 1e-3 * k + threadIdx.x * 1e-9;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            if (KIND == 25) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(accd[k]) : "v"(bd));
            if (KIND == 26) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(accd[k]) : "v"(bd));
            if (KIND == 27) asm volatile("v_add_f64 %0, %1, %0" : "+v"(accd[k]) : "v"(bd));
            if (KIND == 28) asm volatile("v_sqrt_f64_e32 %0, %0" : "+v"(accd[k]));
            if (KIND == 29) asm volatile("v_rcp_f64_e32 %0, %0" : "+v"(accd[k]));
        }
    }
    double s = 0;
    for (int k = 0; k < UNROLL; ++k) s += accd[k];
    if (s == 12345.678) out[0] = (float)s;
}

template <int KIND>
double run(int blocks_per_cu, int cus, int insts_per_slot, double clock_hz) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = cus * blocks_per_cu;
    const f2 sg = {0.5f, 0.25f};
    auto launch = [&]() {
        if (KIND >= 25 && KIND <= 29) cost_kernel_f64<KIND><<<blocks, 256>>>(d, 1.0f);
        else cost_kernel<KIND><<<blocks, 256>>>(d, 1.0f, sg);
    };
    launch();
    hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions per SIMD: each block has 4 waves = one per SIMD
        const double per_simd = (double)ITERS * UNROLL * insts_per_slot * blocks_per_cu;
        const double cyc = ms * 1e-3 * clock_hz / per_simd;
        if (cyc < best) best = cyc;
    }
    hipFree(d);
    return best;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double clock_hz = p.clockRate * 1e3;
    printf("{\"device\": \"%s\", \"cus\": %d, \"nominal_clock_mhz\": %d, \"unit\": \"nominal-clock cycles per wave-instruction per SIMD\"", p.gcnArchName, cus, p.clockRate / 1000);
    for (int bpc : {1, 2, 5, 8}) {
        printf(", \"waves_per_simd_%d\": {", bpc);
        bool first = true;
#define X(ID, NAME, TEXT, N) printf("%s\"" #NAME "\": %.2f", first ? "" : ", ", run<ID>(bpc, cus, N, clock_hz)); first = false;
        OPS(X)
#undef X
        printf("}");
    }
    printf("}\n");
    return 0;
}
