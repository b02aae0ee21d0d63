// vecmath.h -- vec3 helpers (vec3.h) and the IEEE sqrt / divide sequences without range handling
// Part of the single gfx950 translation unit rtiow_hip.hip (included there, in this order; internal linkage).
#pragma once
#include "probes.h"

namespace {

template <class T> __device__ __forceinline__ T dot3(V3<T> u, V3<T> v) {   // vec3.h:93-97
    return RT_FMA(u.z, v.z, RT_FMA(u.y, v.y, u.x * v.x));
}
template <class T> __device__ __forceinline__ V3<T> madd3(T t, V3<T> v, V3<T> w) {   // w + t*v
    return {RT_FMA(t, v.x, w.x), RT_FMA(t, v.y, w.y), RT_FMA(t, v.z, w.z)};
}
template <class T> __device__ __forceinline__ V3<T> scale3(T t, V3<T> v) { return {t * v.x, t * v.y, t * v.z}; }
__device__ __forceinline__ float inv_sqrt_accepted(float x);
__device__ __forceinline__ double inv_sqrt_accepted(double x);
template <class T> __device__ __forceinline__ V3<T> unit3(V3<T> v) {       // vec3.h:105-107, 89-91
    const T dd = dot3(v, v);
    T inv;
    // when every lane here has |v|^2 in [2^-80, 2^80] the wave takes 1/sqrt without the range handling
    // (inv_sqrt_accepted: fp32 16 instead of 26 instructions, same bits); one lane outside and all take the long form
    if (__builtin_amdgcn_ballot_w64(!(dd >= (T)0x1p-80 && dd <= (T)0x1p80)) == 0) inv = inv_sqrt_accepted(dd);
    else inv = (T)1 / Real<T>::sqrt(dd);
    return scale3(inv, v);
}
template <class T> __device__ __forceinline__ V3<T> reflect3(V3<T> v, V3<T> n) {   // vec3.h:129-131
    T k = (T)2 * dot3(v, n);
    return madd3(-k, n, v);
}

// 1 / sqrt(lensq) of an ACCEPTED candidate (vec3.h:126: p / sqrt(lensq)), i.e. the IEEE square root followed by the
// IEEE reciprocal, for an operand known to lie in (1e-8, 1].  The compiler's correctly rounded sequences (15 + 11
// instructions) spend 10 of them on what such an operand never needs: the 2^32 pre-scaling of a square root below
// 2^-96 with its un-scaling, the zero / infinity pass-through, v_div_scale on both operands (no scaling for a
// numerator 1 and a denominator in [1e-4, 1]: the scaled values ARE the operands and the flag is clear, so
// v_div_fmas is a plain fma) and v_div_fixup (specials only).  What is left is those sequences' own arithmetic,
// instruction for instruction: raw v_sqrt_f32 (<= 1 ulp) corrected by the two residual tests against its neighbours,
// then raw v_rcp_f32 with one Newton step and the two quotient refinements.  Same bits as
// `1.0f / sqrtf(lensq)` (the full-frame goldens compare every pixel); the fp64 twin follows below.
// The same holds for any operand in [2^-80, 2^80] (square root in [2^-40, 2^40]: v_sqrt_f32 needs no scaling
// from 2^-96 up, v_div_scale none while the exponents of 1 and the root differ by less than 96): gen_primary
// uses it for |D|^2 of the primary rays when the host has bounded that for the whole frame.
// The square-root half on its own: correctly rounded sqrt of a normal x in [2^-90, 2^90] (ieee_roots, behind a wave-wide range test).
__device__ __forceinline__ float sqrt_in_range(float x) {
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s0) - 1u), sp = __uint_as_float(__float_as_uint(s0) + 1u);
    const float rm = __builtin_fmaf(-sm, s0, x), rp = __builtin_fmaf(-sp, s0, x);
    const float s = rm <= 0.0f ? sm : s0;
    return rp > 0.0f ? sp : s;
}
// fp64: the compiler's correctly rounded sqrt is v_rsq_f64 and nine multiply-adds (Goldschmidt with two residual
// corrections) wrapped in a 2^256 pre-scaling of operands below 2^-767 and the zero / infinity pass-through: eight of
// its eighteen instructions.  The ten in the middle, as emitted:
__device__ __forceinline__ double sqrt_in_range(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, x);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ float inv_sqrt_accepted(float x) {
#ifdef RTIOW_GENERIC_RUV_NORMALISATION
    return 1.0f / __builtin_sqrtf(x);
#else
    const float s = sqrt_in_range(x);
    float r = __builtin_amdgcn_rcpf(s);
    r = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
    float q = r;                                             // numerator 1: q = 1 * r
    q = __builtin_fmaf(__builtin_fmaf(-s, q, 1.0f), r, q);
    return __builtin_fmaf(__builtin_fmaf(-s, q, 1.0f), r, q);
#endif
}
// fp64 (accepted lensq in (1e-160, 1], and any operand in [2^-90, 2^90]): the ten instructions above, then 1 / s as
// the division's own arithmetic with a numerator of 1 (v_rcp_f64, two Newton steps, q = 1 * r, one refinement).
__device__ __forceinline__ double inv_sqrt_accepted(double x) {
#ifdef RTIOW_GENERIC_RUV_NORMALISATION
    return 1.0 / __builtin_sqrt(x);
#else
    const double s = sqrt_in_range(x);
    double r = __builtin_amdgcn_rcp(s);
    r = __builtin_fma(r, __builtin_fma(-s, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-s, r, 1.0), r);
    return __builtin_fma(__builtin_fma(-s, r, 1.0), r, r);
#endif
}

// sqrt(x) for every live lane of the wave: the short form when all their operands are normal and mid-range (one
// ballot), else the compiler's sequence for everyone -- the same bits either way.
template <class T> __device__ __forceinline__ T sqrt_wave_checked(T x) {
    if (__builtin_amdgcn_ballot_w64(!(x >= (T)0x1p-90 && x <= (T)0x1p90)) == 0) return sqrt_in_range(x);
    return Real<T>::sqrt(x);
}

}  // namespace
