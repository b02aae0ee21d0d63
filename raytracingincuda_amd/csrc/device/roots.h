// roots.h -- second half of hit_sphere (hittable.h:50-57): pre-test, shared reciprocal, IEEE roots
// Part of the single gfx950 translation unit rtiow_hip.hip (included there, in this order; internal linkage).
#pragma once
#include "vecmath.h"

namespace {

// raw hardware square root (v_sqrt_f32 / v_sqrt_f64, error <= 2^-22 relative): used ONLY inside the
// conservative pre-test below, never for a value that reaches the image.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double fast_sqrt(double x) { return __builtin_amdgcn_sqrt(x); }

// Second half of hit_sphere (hittable.h:50-57) once the discriminant is known to be >= 0.
//
// Most spheres that get here are then rejected (behind the origin, or farther than the hit
// already found), after an IEEE sqrt and up to two IEEE divisions (~56 instructions).  A cheap
// pre-test drops a sphere ONLY when the exact code below provably would: all roots of one ray
// share the divisor a > 0, so they order like their numerators n = h -+ sqrt(disc);
//   (behind) n2 + e <  tmin*a*(1-2^-20)  =>  both exact roots <= tmin;
//   (far)    n1 - e >  closest*a*(1+2^-20) =>  exact near root >= closest, hence the far one too.
// e bounds |n_fast - n_ref|: the raw sqrt is within 1 ulp (2^-23 relative) and the reference's
// IEEE sqrt within half an ulp, so the two square roots differ by at most 0.75 * 2^-22 * sqrt;
// the additions h -+ sqrt round once on each side (2^-24 |n| each, |n| <= |h| + sqrt).  Hence
// e = 2^-22 (|h| + sqrt) + 2^-60 (the constant covers a raw sqrt that flushes a denormal
// discriminant to zero).  Rays leaving the ground sphere (|h|, sqrt ~ 1000, far root = rounding
// noise ~1e-4) are what the tight bound is for: with 2^-20 every one of them fell through to
// the IEEE code.  In doubt the exact code runs, so the result is unchanged.
//
// ANYORDER: the caller does not visit the spheres in index order (hit_world_grid).  The reference's
// loop keeps the FIRST sphere among equal roots (`root < closest_so_far` is strict, hittable.h:54-56),
// i.e. its result is the lexicographic minimum of (t, index); out of order that is `root < closest,
// or root == closest and a lower index`.  Testing a sphere twice changes nothing.
template <class T>
__device__ __forceinline__ bool root_pretest_rejects(T h, T disc, T a, T closest) {
    const T tmin = (T)0.001;
    const T kappa = (T)2.384185791015625e-07;                      // 2^-22
    const T sq_approx = fast_sqrt(disc);
    const T e = RT_FMA(kappa, Real<T>::fabs(h) + sq_approx, (T)8.673617379884035e-19);   // + 2^-60
    const T behind_bound = (tmin * a) * (T)0.99999904632568359375;  // tmin*a*(1-2^-20)
    const T far_bound = (closest * a) * (T)1.00000095367431640625; // closest*a*(1+2^-20); inf while nothing is hit
    return (int)((h + sq_approx) + e < behind_bound) | (int)((h - sq_approx) - e > far_bound);   // one branch, not two
}

// All roots of one ray are divided by the same a = d.d.  The correctly rounded fp32 division the compiler emits
// is   d' = div_scale(a), n' = div_scale(n);  r = rcp(d'); r = fma(fma(-d', r, 1), r, r);        <- a only
//      q = n' r; q = fma(fma(-d', q, n'), r, q); q = div_fmas(fma(-d', q, n'), r, q); div_fixup   <- per quotient
// and for operands that need no scaling (d' = a, n' = n, flag clear: div_fmas is an fma, div_fixup the identity)
// its first line depends on the ray alone.  hit_world_grid computes r once per segment (refined_reciprocal) and every
// quotient of the segment is the second line's five instructions instead of eleven -- the same instructions on the
// same values, hence the same bits.  "No scaling" is guaranteed, not tested per quotient: v_div_scale_f32 leaves
// its operands alone while a is normal, 1/a is normal and -126 < exponent(n) - exponent(a) < 96.  The host vouches
// for the scene (range_flags bit 1: every coordinate of spheres and lens below 2^18, so |oc| < 2^21), the wave
// checks a in [2^-40, 2^40] for all its lanes (else the whole wave divides the long way for that segment), which
// bounds |n| = |h -+ sqrt(disc)| by 2^43 and the exponent difference by 83.  A quotient so small that the IEEE
// sequence would scale it is < 2^-80 on both paths and fails `tmin < root` either way; only accepted roots are stored.
template <class T> struct FastDiv { T ra; bool on; };          // on is wave-uniform
__device__ __forceinline__ float shared_rcp_quotient(float n, float a, float ra) {
    float q = n * ra;
    q = __builtin_fmaf(__builtin_fmaf(-a, q, n), ra, q);
    return __builtin_fmaf(__builtin_fmaf(-a, q, n), ra, q);
}
// fp64: the compiler's sequence is  d' = div_scale(a), n' = div_scale(n); r = rcp(d'); twice r = fma(r, fma(-d', r, 1), r);
//                                    q = n' r; div_fixup(div_fmas(fma(-d', q, n'), r, q))
// -- the same split: six instructions (one of them v_rcp_f64, 16 cycles) per ray, three per quotient.  No scaling
// while the exponents of n and a differ by less than 768; the bounds above leave 83.
__device__ __forceinline__ double shared_rcp_quotient(double n, double a, double ra) {
    const double q = n * ra;
    return __builtin_fma(__builtin_fma(-a, q, n), ra, q);
}

__device__ __forceinline__ float refined_reciprocal(float a) {        // the divisor-only half of the fp32 sequence
    const float r = __builtin_amdgcn_rcpf(a);
    return __builtin_fmaf(__builtin_fmaf(-a, r, 1.0f), r, r);
}
__device__ __forceinline__ double refined_reciprocal(double a) {      // ... and of the fp64 sequence
    double r = __builtin_amdgcn_rcp(a);
    r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
    return __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
}

template <class T, bool ANYORDER = false>
__device__ __forceinline__ void ieee_roots(int s, T h, T disc, T a, T& closest, int& hit, FastDiv<T> fd);

template <class T, bool ANYORDER = false>
__device__ __forceinline__ void finish_sphere_test(int s, T h, T disc, T a, T& closest, int& hit, FastDiv<T> fd = FastDiv<T>{(T)0, false}) {
    PATH_STAT(PS_FINISH_CALL);
    // The pre-test pays in front of the full IEEE sequences (two divisions of 11 instructions, a square root of 15).  With the
    // segment's shared reciprocal (fd.on: hit_world_grid) the block it guards is 26 straight-line instructions -- short square
    // root, both quotients, the interval tests as selects -- and the 13 of the pre-test plus its divergent region cost more than
    // they save: -2.5 % vector and -6 % scalar instructions, -2.8 % time on the headline frame, -3.9 % on the 487-sphere scene
    // (profiles/r04/ab_second_root.jsonl; round 2 measured "no pre-test" at equal time when the block still branched around
    // its second quotient).  -DRTIOW_R03_ROOT_FINISH restores round 3's form for A/B runs.
#ifdef RTIOW_R03_ROOT_FINISH
    const bool pretest = true;
#else
    const bool pretest = !fd.on;
#endif
    if (pretest && root_pretest_rejects<T>(h, disc, a, closest)) return;
    ieee_roots<T, ANYORDER>(s, h, disc, a, closest, hit, fd);
}

// hittable.h:50-57 proper.
template <class T, bool ANYORDER>
__device__ __forceinline__ void ieee_roots(int s, T h, T disc, T a, T& closest, int& hit, FastDiv<T> fd) {
    const T tmin = (T)0.001;
    PATH_STAT(PS_IEEE_BLOCK);
    T sq;                                                           // :50
    if (fd.on && __builtin_amdgcn_ballot_w64(!(disc >= (T)0x1p-90 && disc <= (T)0x1p90)) == 0)
        sq = sqrt_in_range(disc);      // every lane here has a normal discriminant well above 2^-96: the IEEE sequence without its range handling
    else
        sq = Real<T>::sqrt(disc);
    T root = fd.on ? shared_rcp_quotient(h - sq, a, fd.ra) : (h - sq) / a;   // :53
    auto inside = [&](T r) {
        // bitwise on purpose: `&&` / `||` here compile to nested divergent regions (a dozen scalar instructions per evaluation),
        // the comparisons have no side effects
        if (ANYORDER) return (bool)((int)(tmin < r) & ((int)(r < closest) | ((int)(r == closest) & (int)((unsigned)s < (unsigned)hit))));
        return (bool)((int)(tmin < r) & (int)(r < closest));
    };
    bool ok = inside(root);                                         // :54
#ifndef RTIOW_R03_ROOT_FINISH
    if (fd.on) {
        // with the shared reciprocal the far root is five instructions: computed for every lane and selected, instead of a
        // divergent region (exec-mask save / branch / restore) around them
        const T far_root = shared_rcp_quotient(h + sq, a, fd.ra);   // :55
        const bool far_ok = inside(far_root);                       // :56
        root = ok ? root : far_root;
        ok = (bool)((int)ok | (int)far_ok);
    } else
#endif
    if (!ok) {
        PATH_STAT(PS_SECOND_DIV);
        root = fd.on ? shared_rcp_quotient(h + sq, a, fd.ra) : (h + sq) / a;   // :55
        ok = inside(root);                                          // :56
    }
    if (ok) { closest = root; hit = s; }                            // hittable.h:88-92
}

}  // namespace
