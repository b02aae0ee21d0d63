// sampling.h -- random_unit_vector, primary ray generation (vec3.h:109-127, camera.h:73-76, :145-155)
// Part of the single gfx950 translation unit rtiow_hip.hip (included there, in this order; internal linkage).
#pragma once
#include "vecmath.h"

namespace {

template <class T> __device__ __forceinline__ V3<T> random_unit_vector(Rng& s) {   // vec3.h:117-127
    // The rejection loop only finds the accepted candidate; its normalisation (an IEEE sqrt and
    // divide, ~30 instructions) runs once after the loop instead of in every round the wave
    // executes for its slowest lane.  Same draws, same arithmetic on the accepted candidate.
    T x, y, z, lensq;
    PATH_STAT(PS_RUV_CALL);
    REGION_BEGIN(ruv);
    for (;;) {
        PATH_STAT(PS_RUV_ROUND);
        T u0, u1, u2;
        Real<T>::uniform3(s, u0, u1, u2);
        x = RT_FMA(u0, (T)2, (T)-1);
        y = RT_FMA(u1, (T)2, (T)-1);
        z = RT_FMA(u2, (T)2, (T)-1);
        lensq = RT_FMA(z, z, RT_FMA(y, y, x * x));
        if (Real<T>::ruv_eps < lensq && lensq <= (T)1) break;
    }
    REGION_END(ruv, RG_RUV_ROUNDS);
    const T inv = inv_sqrt_accepted(lensq);
    return {inv * x, inv * y, inv * z};
}

// The same rejection loop, at most `rounds` rounds of it (shade_step<T, true>): a lane that has not found its candidate
// yet returns false with the generator advanced by the rounds it drew, and goes on in the wave's next iteration.
// Accepted: x, y, z, lensq hold the candidate (the caller normalises it once, vec3.h:126).
template <class T> __device__ __forceinline__ bool random_unit_vector_rounds(Rng& s, int rounds, T& x, T& y, T& z, T& lensq) {
    for (int r = 0; r < rounds; ++r) {
        PATH_STAT(PS_RUV_ROUND);
        T u0, u1, u2;
        Real<T>::uniform3(s, u0, u1, u2);
        x = RT_FMA(u0, (T)2, (T)-1);
        y = RT_FMA(u1, (T)2, (T)-1);
        z = RT_FMA(u2, (T)2, (T)-1);
        lensq = RT_FMA(z, z, RT_FMA(y, y, x * x));
        if (Real<T>::ruv_eps < lensq && lensq <= (T)1) return true;
    }
    return false;
}

// Pins wave-uniform values in scalar registers at this point of the program (the loads that produce them are issued before it).
__device__ __forceinline__ void keep_scalar(float& a, float& b, float& c, float& d, float& e, float& f) { asm volatile("" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f)); }
__device__ __forceinline__ void keep_scalar(double& a, double& b, double& c, double& d, double& e, double& f) { asm volatile("" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f)); }

// ---- The rotated trip of persistent_body (RTIOW_MERGED_ROUNDS, fp64): gen_primary and shade_step cut at their rejection loops, so that ONE loop
// serves both.  A lane is looking for a unit-vector candidate (vec3.h:117-125: three numbers a round) or for a lens sample (vec3.h:109-115: two
// numbers a round).  Both form x = 2 u0 - 1, y = 2 u1 - 1 and fma(y, y, x x) the same way, so a round draws two numbers for every open lane and
// the third, z and fma(z, z, .) for the unit-vector lanes only -- and a trip runs max(rounds) of them instead of the sum of two loops' rounds.
// At most `rounds` rounds; a lane still open keeps its kind.  A pixel's draws and their order are unchanged: the same steps of its generator, the
// same arithmetic on the same draws.  (fp32 was built too -- a fused in-place generator block with the third step under a narrowed exec mask --
// and measured -1.8 % vector instructions, +-0.5 % time: profiles/experiments/r05_merged_rejection_rounds.md; the fp32 kernels keep one loop.)
enum { MR_CLOSED = 0, MR_DISK = 1, MR_RUV = 2 };       // what a lane is looking for (ONE variable: two flags updated in the arms of an if went through scratch memory)
__device__ __forceinline__ void merged_rounds(Rng&, int, int&, float&, float&, float&, float&) {}   // never called: persistent_body takes the rotated trip for fp64 only
// fp64: curand_uniform_double takes two steps of the generator per number (xorwow.h), so a lens round is four steps and a unit-vector round six -- the
// rounds are the heaviest blocks of the fp64 trip (72 and 48 vector instructions); merged, four of the six steps and two of the three conversions are shared.
__device__ __forceinline__ void merged_rounds(Rng& s, int rounds, int& kind, double& x, double& y, double& z, double& lensq) {
    typedef double T;
#pragma unroll 3        // (rolled: the same time; two or four rounds per trip: +3.4 / +2.6 %, profiles/r05/merged_rounds_f64_variants.jsonl)
    for (int r = 0; r < rounds; ++r) {
        if (__builtin_amdgcn_ballot_w64(kind != MR_CLOSED) == 0) break;
        if (kind != MR_CLOSED) {
            const bool three = kind == MR_RUV;
            if (three) PATH_STAT(PS_RUV_ROUND); else PATH_STAT(PS_DISK_ROUND);
            const double u0 = Real<double>::uniform(s);
            const double u1 = Real<double>::uniform(s);
            x = RT_FMA(u0, 2.0, -1.0);
            y = RT_FMA(u1, 2.0, -1.0);
            const double l2 = RT_FMA(y, y, x * x);
            bool accepted = l2 < 1.0;                                               // vec3.h:113
            if (three) {
                const double u2 = Real<double>::uniform(s);
                z = RT_FMA(u2, 2.0, -1.0);
                lensq = RT_FMA(z, z, l2);
                accepted = Real<double>::ruv_eps < lensq && lensq <= 1.0;             // vec3.h:125
            }
            kind = accepted ? MR_CLOSED : kind;
        }
    }
}
// the lens loop of a lane that is still open behind the merged rounds (one lane in a hundred after three rounds)
template <class T> __device__ __forceinline__ void disk_candidate(Rng& s, T& px, T& py) {
    for (;;) {
        PATH_STAT(PS_DISK_ROUND);
        T u0, u1;
        Real<T>::uniform2(s, u0, u1);
        px = RT_FMA((T)2, u0, (T)-1);
        py = RT_FMA((T)2, u1, (T)-1);
        if (RT_FMA(py, py, px * px) < (T)1) break;
    }
}
// camera.h:145-146 (the two jitter draws, first argument first) and camera.h:147-155 once the lens sample is known.
template <class T> __device__ __forceinline__ void primary_jitter(Rng& s, T& ox, T& oy) {
    ox = Real<T>::uniform(s) - (T)0.5;
    oy = Real<T>::uniform(s) - (T)0.5;
}
template <class T>
__device__ __forceinline__ void primary_finish(const RenderParams<T>& p, int i, int j, T ox, T oy, bool defocus, T px, T py, V3<T>& O, V3<T>& D, T& sky_uy) {
    const auto& c = cam_of(p);
    const T fi = (T)i + ox, fj = (T)j + oy;
    const V3<T> ps = madd3(fj, V3<T>{c.dv.x, c.dv.y, c.dv.z}, madd3(fi, V3<T>{c.du.x, c.du.y, c.du.z}, V3<T>{c.pixel00.x, c.pixel00.y, c.pixel00.z}));
    const V3<T> ctr = {c.center.x, c.center.y, c.center.z};
    V3<T> org = ctr;
    if (defocus) org = madd3(py, V3<T>{c.ddv.x, c.ddv.y, c.ddv.z}, madd3(px, V3<T>{c.ddu.x, c.ddu.y, c.ddu.z}, ctr));
    O = org;
    D = {ps.x - org.x, ps.y - org.y, ps.z - org.z};
    const T dd = dot3(D, D);
    T inv;
    if (p.range_flags & 1) inv = inv_sqrt_accepted(dd);   // wave-uniform choice, same bits
    else inv = (T)1 / Real<T>::sqrt(dd);
    sky_uy = inv * D.y;
}

// One primary ray: camera.h:145-155 (+ :73-76, vec3.h:109-115).  Also returns the y
// component of the PRIMARY ray's unit direction, all the sky term needs (camera.h:121).
template <class T>
__device__ __forceinline__ void gen_primary(const RenderParams<T>& p, int i, int j, Rng& s,
                                            V3<T>& O, V3<T>& D, T& sky_uy) {
    PATH_STAT(PS_GEN_PRIMARY);
    const auto& c = cam_of(p);                   // scalar loads from the kernarg segment, here
#ifndef RTIOW_CAMERA_ONE_FETCH
#define RTIOW_CAMERA_ONE_FETCH 1
#endif
    // The lens vectors are needed only behind the disk's rejection loop, and the compiler fetched them there: a SECOND scalar-memory round trip
    // per primary ray, ~250 cycles that nothing hides when the wave holds one ray (a lone trip waits 1140 of its 3080 cycles, two of its
    // waits are these: profiles/r05/lone_trip_counters_scene3.json).  Read here, with the rest of the camera, they arrive while the jitter and the
    // disk loop draw: six scalar registers live across that loop instead of a stall behind it.
    T ddux = c.ddu.x, dduy = c.ddu.y, dduz = c.ddu.z, ddvx = c.ddv.x, ddvy = c.ddv.y, ddvz = c.ddv.z;
    if (RTIOW_CAMERA_ONE_FETCH) keep_scalar(ddux, dduy, dduz, ddvx, ddvy, ddvz);
    T ox = Real<T>::uniform(s) - (T)0.5;
    T oy = Real<T>::uniform(s) - (T)0.5;
    T fi = (T)i + ox, fj = (T)j + oy;
    V3<T> ps = madd3(fj, V3<T>{c.dv.x, c.dv.y, c.dv.z}, madd3(fi, V3<T>{c.du.x, c.du.y, c.du.z}, V3<T>{c.pixel00.x, c.pixel00.y, c.pixel00.z}));
    const V3<T> ctr = {c.center.x, c.center.y, c.center.z};
    V3<T> org = ctr;
    if (!((T)c.defocus_angle <= (T)0)) {
        T px, py;
        for (;;) {
            PATH_STAT(PS_DISK_ROUND);
            T u0, u1;
            Real<T>::uniform2(s, u0, u1);
            px = RT_FMA((T)2, u0, (T)-1);
            py = RT_FMA((T)2, u1, (T)-1);
            if (RT_FMA(py, py, px * px) < (T)1) break;
        }
        if (RTIOW_CAMERA_ONE_FETCH) org = madd3(py, V3<T>{ddvx, ddvy, ddvz}, madd3(px, V3<T>{ddux, dduy, dduz}, ctr));
        else org = madd3(py, V3<T>{c.ddv.x, c.ddv.y, c.ddv.z}, madd3(px, V3<T>{c.ddu.x, c.ddu.y, c.ddu.z}, ctr));
    }
    O = org;
    D = {ps.x - org.x, ps.y - org.y, ps.z - org.z};
    const T dd = dot3(D, D);
    T inv;
    if (p.range_flags & 1) inv = inv_sqrt_accepted(dd);   // wave-uniform choice, same bits
    else inv = (T)1 / Real<T>::sqrt(dd);
    sky_uy = inv * D.y;
}

}  // namespace
