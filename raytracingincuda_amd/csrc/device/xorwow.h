// xorwow.h -- XORWOW draws (curand_init / curand_uniform semantics, rtweekend.h:32-50), Real<T>, rng_init_kernel
// Part of the single gfx950 translation unit rtiow_hip.hip (included there, in this order; internal linkage).
#pragma once

namespace {

// =====================================================================================
// XORWOW: state in registers; skip-ahead matrices built on the host.
// =====================================================================================
struct Rng { uint32_t v0, v1, v2, v3, v4, d; };

__device__ __forceinline__ uint32_t rng_next(Rng& s) {
    uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1; s.v1 = s.v2; s.v2 = s.v3; s.v3 = s.v4;
    // gfx950's three-input bit operation (truth table 0x96 = a^b^c) takes one of the four xors: 6 instead
    // of 7 vector instructions per draw (profiles/archive/r02_ab_xorwow_bitop3.jsonl: -2.4 % SQ_INSTS_VALU, -1.2 % time)
    // t << 1 as t + t: on gfx950 v_lshlrev_b32 issues at 5.3 cycles per wave-instruction, v_add_u32 at 3.7
    // (bin/valu_cost); the compiler turns a source-level t + t back into the shift, hence the one-line asm.
    uint32_t t2;
    asm("v_add_u32 %0, %1, %1" : "=v"(t2) : "v"(t));
    s.v4 = __builtin_amdgcn_bitop3_b32(s.v4, s.v4 << 4, t, 0x96) ^ t2;
    s.d += 362437u;
    return s.v4 + s.d;
}

// Two / three draws at once with the state rotated IN PLACE (fp32 rejection loops).  A loop whose round
// draws k numbers rotates the five state words by k places per trip; the compiler materialises that as five
// register copies at the back edge.  Tied operands leave nothing to copy at the back edge, and inside the block
// a rotation by three takes two moves and one by two takes three: 20 instead of 23 vector instructions for
// three draws, 15 instead of 17 for two.  Draw i of the block is then (new word) + d + i * 362437.
// (t << 1 is written t + t: v_add_u32 issues faster than v_lshlrev_b32, see rng_next.)
__device__ __forceinline__ void rng_step3(Rng& s) {      // afterwards the draws are v2 + d1, v3 + d2, v4 + d3
    uint32_t t1, t2, t3, c;
    asm("v_lshrrev_b32 %5, 2, %0\n\t"
        "v_lshrrev_b32 %6, 2, %1\n\t"
        "v_lshrrev_b32 %7, 2, %2\n\t"
        "v_xor_b32 %5, %5, %0\n\t"
        "v_xor_b32 %6, %6, %1\n\t"
        "v_xor_b32 %7, %7, %2\n\t"
        "v_mov_b32 %0, %3\n\t"
        "v_mov_b32 %1, %4\n\t"
        "v_lshlrev_b32 %8, 4, %4\n\t"
        "v_bitop3_b32 %2, %4, %8, %5 bitop3:0x96\n\t"
        "v_add_u32 %5, %5, %5\n\t"
        "v_xor_b32 %2, %2, %5\n\t"
        "v_lshlrev_b32 %8, 4, %2\n\t"
        "v_bitop3_b32 %3, %2, %8, %6 bitop3:0x96\n\t"
        "v_add_u32 %6, %6, %6\n\t"
        "v_xor_b32 %3, %3, %6\n\t"
        "v_lshlrev_b32 %8, 4, %3\n\t"
        "v_bitop3_b32 %4, %3, %8, %7 bitop3:0x96\n\t"
        "v_add_u32 %7, %7, %7\n\t"
        "v_xor_b32 %4, %4, %7"
        : "+v"(s.v0), "+v"(s.v1), "+v"(s.v2), "+v"(s.v3), "+v"(s.v4), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(c));
}
__device__ __forceinline__ void rng_step2(Rng& s) {      // afterwards the draws are v3 + d1, v4 + d2
    uint32_t t1, t2, c;
    asm("v_lshrrev_b32 %5, 2, %0\n\t"
        "v_lshrrev_b32 %6, 2, %1\n\t"
        "v_xor_b32 %5, %5, %0\n\t"
        "v_xor_b32 %6, %6, %1\n\t"
        "v_mov_b32 %0, %2\n\t"
        "v_mov_b32 %1, %3\n\t"
        "v_lshlrev_b32 %7, 4, %4\n\t"
        "v_bitop3_b32 %3, %4, %7, %5 bitop3:0x96\n\t"
        "v_add_u32 %5, %5, %5\n\t"
        "v_mov_b32 %2, %4\n\t"
        "v_xor_b32 %3, %3, %5\n\t"
        "v_lshlrev_b32 %7, 4, %3\n\t"
        "v_bitop3_b32 %4, %3, %7, %6 bitop3:0x96\n\t"
        "v_add_u32 %6, %6, %6\n\t"
        "v_xor_b32 %4, %4, %6"
        : "+v"(s.v0), "+v"(s.v1), "+v"(s.v2), "+v"(s.v3), "+v"(s.v4), "=&v"(t1), "=&v"(t2), "=&v"(c));
}

template <class T> struct Real;
template <> struct Real<float> {
    // curand_uniform: (0,1]
    static __device__ __forceinline__ float uniform(Rng& s) {
        uint32_t x = rng_next(s);
        return __builtin_fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    }
    static __device__ __forceinline__ float from_u32(uint32_t x) {
        return __builtin_fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    }
    static __device__ __forceinline__ void uniform2(Rng& s, float& a, float& b) {
#ifdef RTIOW_NO_INPLACE_RNG
        a = uniform(s); b = uniform(s);
#else
        rng_step2(s);
        a = from_u32(s.v3 + (s.d + 362437u)); s.d += 2u * 362437u; b = from_u32(s.v4 + s.d);
#endif
    }
    static __device__ __forceinline__ void uniform3(Rng& s, float& a, float& b, float& c) {
#ifdef RTIOW_NO_INPLACE_RNG
        a = uniform(s); b = uniform(s); c = uniform(s);
#else
        rng_step3(s);
        a = from_u32(s.v2 + (s.d + 362437u)); b = from_u32(s.v3 + (s.d + 2u * 362437u)); s.d += 3u * 362437u; c = from_u32(s.v4 + s.d);
#endif
    }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float sqrt(float a) { return __builtin_sqrtf(a); }
    static __device__ __forceinline__ float fmin(float a, float b) { return __builtin_fminf(a, b); }
    static __device__ __forceinline__ float fmax(float a, float b) { return __builtin_fmaxf(a, b); }
    static __device__ __forceinline__ float fabs(float a) { return __builtin_fabsf(a); }
    static constexpr float near_zero = 1e-6f;   // vec3.h:50
    static constexpr float ruv_eps = 1e-8f;     // vec3.h:124
};
template <> struct Real<double> {
    // curand_uniform_double (XORWOW): 53 bits from two draws
    static __device__ __forceinline__ double uniform(Rng& s) {
        uint32_t x = rng_next(s);
        uint32_t y = rng_next(s);
        uint64_t z = (uint64_t)x ^ ((uint64_t)y << 21);
        return __builtin_fma((double)z, 1.1102230246251565e-16, 5.5511151231257827e-17);
    }
    static __device__ __forceinline__ double from_u32(uint32_t x) {
        return __builtin_fma((double)x, 1.1102230246251565e-16, 5.5511151231257827e-17);
    }
    static __device__ __forceinline__ void uniform2(Rng& s, double& a, double& b) { a = uniform(s); b = uniform(s); }
    static __device__ __forceinline__ void uniform3(Rng& s, double& a, double& b, double& c) { a = uniform(s); b = uniform(s); c = uniform(s); }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ double sqrt(double a) { return __builtin_sqrt(a); }
    static __device__ __forceinline__ double fmin(double a, double b) { return __builtin_fmin(a, b); }
    static __device__ __forceinline__ double fmax(double a, double b) { return __builtin_fmax(a, b); }
    static __device__ __forceinline__ double fabs(double a) { return __builtin_fabs(a); }
    static constexpr double near_zero = 1e-8;   // GlobalDouble vec3.h:50
    static constexpr double ruv_eps = 1e-160;   // GlobalDouble vec3.h:125
};

constexpr int XW_BITS = 160;
constexpr int XW_WORDS = 5;
constexpr int XW_JUMPS = 32;                         // subsequence index bits supported
constexpr size_t XW_MAT_WORDS = (size_t)XW_BITS * XW_WORDS;

// jump: [XW_JUMPS][160][5]; column `bit` of matrix b is the state reached from basis bit.
// All lanes walk the same (b, bit) order, so the column reads are wave-uniform scalar loads.
// One product of a jump matrix with the lane's state: 160 x 5 masked xors (the whole cost of curand_init).
__device__ __forceinline__ void xw_apply_jump(const uint32_t* __restrict__ m, uint32_t (&v)[XW_WORDS]) {
    uint32_t o[XW_WORDS] = {0, 0, 0, 0, 0};
    for (int w = 0; w < XW_WORDS; ++w) {
        const uint32_t word = v[w];
        for (int bit = 0; bit < 32; ++bit) {
            const uint32_t mask = 0u - ((word >> bit) & 1u);
            const uint32_t* c = m + (size_t)(w * 32 + bit) * XW_WORDS;
#pragma unroll
            for (int k = 0; k < XW_WORDS; ++k) o[k] ^= c[k] & mask;
        }
    }
#pragma unroll
    for (int k = 0; k < XW_WORDS; ++k) v[k] = o[k];
}

// curand_init(seed, seq, 0) = J^seq * s0 with J the 2^67-step matrix.  seq = hi * 2^XW_LOW_BITS + lo and the powers commute, so the
// state is J^(hi << XW_LOW_BITS) * (J^lo * s0): xw_low_table_kernel computes the 2^XW_LOW_BITS states J^lo * s0 once per seed (one thread each,
// <= XW_LOW_BITS products), and rng_init_kernel starts every pixel from its table entry and applies only the HIGH bits' matrices -- which
// consecutive pixels share, so a wave executes a product only for the set bits of its common high part.  1920 x 1080: 4.5 products per wave on
// average instead of 13.5 (the six lowest bits differ between the lanes of a wave, so every one of their matrices was applied by every wave):
// 1.26 -> 0.61 ms (rtiow_stats.rng_init_ms, profiles/r04/bench_n1.json).  GF(2) linear algebra: the same states bit for bit (tests/test_gpu_parity.py, rocRAND known answers).
constexpr int XW_LOW_BITS = 12;
__global__ void __launch_bounds__(256)
xw_low_table_kernel(uint32_t* __restrict__ table, const uint32_t* __restrict__ jump, uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3, uint32_t s4) {
    const uint32_t lo = blockIdx.x * blockDim.x + threadIdx.x;
    if (lo >= (1u << XW_LOW_BITS)) return;
    uint32_t v[XW_WORDS] = {s0, s1, s2, s3, s4};
    for (int b = 0; b < XW_LOW_BITS; ++b)
        if ((lo >> b) & 1u) xw_apply_jump(jump + (size_t)b * XW_MAT_WORDS, v);
#pragma unroll
    for (int k = 0; k < XW_WORDS; ++k) table[(size_t)lo * XW_WORDS + k] = v[k];
}

__global__ void __launch_bounds__(256)
rng_init_kernel(uint32_t* __restrict__ states, const uint32_t* __restrict__ jump, const uint32_t* __restrict__ low_table, uint32_t d0,
                int W, int H, int local_rows, int rank, int nranks, int strip_rows) {
    const int npix = W * local_rows;
    const int lp = blockIdx.x * blockDim.x + threadIdx.x;
    if (lp >= npix) return;
    const int jl = lp / W, i = lp - jl * W;
    const int j = ((jl / strip_rows) * nranks + rank) * strip_rows + (jl % strip_rows);
    const uint32_t seq = (uint32_t)(j * W + i);       // camera.h:134 pixel_index, rtweekend.h:49
    uint32_t v[XW_WORDS];
    const uint32_t* t = low_table + (size_t)(seq & ((1u << XW_LOW_BITS) - 1u)) * XW_WORDS;
#pragma unroll
    for (int k = 0; k < XW_WORDS; ++k) v[k] = t[k];
    for (int b = XW_LOW_BITS; b < XW_JUMPS; ++b) {
        if (!((seq >> b) & 1u)) continue;
        xw_apply_jump(jump + (size_t)b * XW_MAT_WORDS, v);
    }
    // SoA so that the render kernel's 6 loads per lane are coalesced.
    states[0 * (size_t)npix + lp] = v[0];
    states[1 * (size_t)npix + lp] = v[1];
    states[2 * (size_t)npix + lp] = v[2];
    states[3 * (size_t)npix + lp] = v[3];
    states[4 * (size_t)npix + lp] = v[4];
    states[5 * (size_t)npix + lp] = d0;               // 2^67*k draws leave the Weyl counter unchanged
    (void)H;
}

}  // namespace
