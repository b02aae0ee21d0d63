// shade.h -- one path segment after hit_world: sky, hit record, scatter (camera.h:88-124, material.h:38-89)
// Part of the single gfx950 translation unit rtiow_hip.hip (included there, in this order; internal linkage).
#pragma once
#include "sampling.h"
#include "hit_grid.h"

namespace {

// Per-lane path state of the flattened samples x bounces loop.
template <class T> struct PathState {
    V3<T> O, D, atten, acc;
    T sky_uy;
    int sample, depth;
    Rng rs;
};

// camera.h:121-123: `double a = 0.5 * (unit_direction.y() + 1.0)` and the two blend weights `(T)(1.0 - a)`, `(T)a`.
// fp32 build: y is a float in [-1, 1], so y + 1.0 and 1.0 - y are exact in double, halving is exact, and the only rounding is the
// conversion back to float -- which is what the float additions (1 + y), (1 - y) round to, halved exactly (both results are 0 or
// >= 2^-25, no denormals).  Four float operations instead of two conversions up, three double operations and two conversions
// down; the same bits (the oracle computes the double form; the full-frame goldens compare every pixel).
__device__ __forceinline__ void sky_weights(float uy, float& w1, float& w2) {
#ifdef RTIOW_SKY_BLEND_IN_DOUBLE
    const double a_sky = 0.5 * ((double)uy + 1.0);
    w1 = (float)(1.0 - a_sky); w2 = (float)a_sky;
#else
    w2 = 0.5f * (uy + 1.0f);
    w1 = 0.5f * (1.0f - uy);
#endif
}
__device__ __forceinline__ void sky_weights(double uy, double& w1, double& w2) {
    const double a_sky = 0.5 * (uy + 1.0);
    w1 = 1.0 - a_sky; w2 = a_sky;
}

// Everything after hit_world in one trip of the loop at camera.h:84: sky on a miss (camera.h:120-124), else hit
// record + scatter (camera.h:88-117).  Returns true when the path ended; `col` is then its colour.
//
// BOUNDED ("retry" form, fp64 persistent kernels): a lambertian or metal lane whose random_unit_vector has not found its
// candidate after `rounds` rounds returns false with retry = true and NOTHING but the generator advanced -- the caller
// keeps (closest, hit), skips hit_world for the lane and calls again in the wave's next iteration, where the hit record
// is recomputed (the wave computes it for its other lanes anyway) and the drawing goes on.  A wave's rejection loop is
// as long as its slowest lane's (5.5 rounds per iteration for ~35 lanes that scatter diffusely, at 11 live lanes), but
// the candidates a lane draws, and their order, are its own: cutting the loop and resuming it changes nothing for the pixel.
template <class T, bool BOUNDED>
__device__ __forceinline__ bool shade_step(const RenderParams<T>& p, const T* lds_shade, PathState<T>& st, T closest, int hit, V3<T>& col, int rounds, bool& retry) {
    retry = false;
    col = {0, 0, 0};
    const V3<T> O = st.O, D = st.D;
    if (hit < 0) {
        PATH_STAT(PS_SKY);
        T w1, w2;
        sky_weights(st.sky_uy, w1, w2);                                      // camera.h:120-124, from the PRIMARY ray
        const V3<T> sky = {RT_FMA(w2, (T)0.5, w1), RT_FMA(w2, (T)0.7, w1), RT_FMA(w2, (T)1.0, w1)};
        col = {st.atten.x * sky.x, st.atten.y * sky.y, st.atten.z * sky.z};
        return true;
    }
    PATH_STAT(PS_SHADE_HIT);
    T rec[12];
    if (p.shade_in_lds) {
#pragma unroll
        for (int k = 0; k < 12; ++k) rec[k] = lds_shade[12 * hit + k];
    } else {
        const T* tbl = screen_of(p).shade_tbl;
#pragma unroll
        for (int k = 0; k < 12; ++k) rec[k] = tbl[12 * (size_t)hit + k];
    }
    const V3<T> C = {rec[0], rec[1], rec[2]};
    const T inv_r = rec[3];
    const V3<T> P = madd3(closest, D, O);                                    // hittable.h:59-63, :21-26
    const V3<T> outward = {inv_r * (P.x - C.x), inv_r * (P.y - C.y), inv_r * (P.z - C.z)};
    const bool front = dot3(D, outward) < (T)0;
    const V3<T> nrm = front ? outward : V3<T>{-outward.x, -outward.y, -outward.z};
    const int mtype = (int)rec[10];
    V3<T> nd;
    V3<T> att = {rec[4], rec[5], rec[6]};
    bool ok = true;
    if (mtype == RTIOW_DIELECTRIC) {                                     // material.h:68-89
        PATH_STAT(PS_DIELECTRIC);
        att = {1, 1, 1};
        const T ri = front ? rec[9] : rec[8];
        const V3<T> ud = unit3(D);
        const T cos_theta = Real<T>::fmin(-dot3(ud, nrm), (T)1);
        const T sin_theta = sqrt_wave_checked(RT_FMA(-cos_theta, cos_theta, (T)1));
        bool reflect_it = ri * sin_theta > (T)1;
        if (!reflect_it) {
            const T r0 = front ? rec[4] : rec[5];
            const float x = (float)((T)1 - cos_theta);
            const float x2 = x * x;
            const float p5 = (x2 * x2) * x;
            PATH_STAT(PS_SCHLICK_DRAW);
            const T refl = RT_FMA((T)1 - r0, (T)p5, r0);
            reflect_it = refl > Real<T>::uniform(st.rs);
        }
        if (reflect_it) {
            nd = reflect3(ud, nrm);
        } else {
            const V3<T> perp = scale3(ri, madd3(cos_theta, nrm, ud));
            const T k = -sqrt_wave_checked(Real<T>::fabs((T)1 - dot3(perp, perp)));
            nd = madd3(k, nrm, perp);
        }
    } else {
        V3<T> ruv;
        if (BOUNDED) {
            T ux, uy, uz, lensq;
            RT_PROBE_RUV(T, st.rs, rounds);
            PATH_STAT(PS_RUV_CALL);
            if (!random_unit_vector_rounds<T>(st.rs, rounds, ux, uy, uz, lensq)) { retry = true; return false; }
            const T inv = inv_sqrt_accepted(lensq);                      // the accepted candidate, normalised once (vec3.h:126)
            ruv = {inv * ux, inv * uy, inv * uz};
        } else {
            RT_PROBE_RUV(T, st.rs, 0x7fffffff);
            ruv = random_unit_vector<T>(st.rs);
        }
        if (mtype == RTIOW_LAMBERTIAN) {                                 // material.h:38-49
            nd = {nrm.x + ruv.x, nrm.y + ruv.y, nrm.z + ruv.z};
            const T e = Real<T>::near_zero;
            if ((int)(Real<T>::fabs(nd.x) < e) & (int)(Real<T>::fabs(nd.y) < e) & (int)(Real<T>::fabs(nd.z) < e)) nd = nrm;
        } else {                                                         // material.h:51-59
            PATH_STAT(PS_METAL);
            const V3<T> ur = unit3(reflect3(D, nrm));
            nd = madd3(rec[7], ruv, ur);
            ok = dot3(nd, nrm) > (T)0;
        }
    }
    if (!ok) return true;                                                // camera.h:117
    st.atten = {st.atten.x * att.x, st.atten.y * att.y, st.atten.z * att.z};   // camera.h:110-115
    st.O = P; st.D = nd;
    ++st.depth;
    return false;
}

// ---- shade_step cut at random_unit_vector (rotated trip of persistent_body, RTIOW_MERGED_ROUNDS): shade_front runs everything that precedes
// the rejection loop (camera.h:88-105 + the whole dielectric), shade_back everything that follows it (vec3.h:126, material.h:41-58).  A lambertian
// or metal lane leaves shade_front with its path state UNTOUCHED (the bounded form calls it again in the next trip, from the same closest, hit);
// shade_back recomputes the hit point (three FMAs) and reads the albedo again.  Same operations on the same values as shade_step.
enum { SF_TERMINATED = 0, SF_CONTINUE = 1, SF_NEED_RUV = 2 };
template <class T> struct ShadeCarry { V3<T> nrm; T fuzz; int mtype; };

template <class T>
__device__ __forceinline__ int shade_front(const RenderParams<T>& p, const T* lds_shade, PathState<T>& st, T closest, int hit, V3<T>& col, ShadeCarry<T>& sc) {
    col = {0, 0, 0};
    const V3<T> O = st.O, D = st.D;
    if (hit < 0) {
        PATH_STAT(PS_SKY);
        T w1, w2;
        sky_weights(st.sky_uy, w1, w2);                                      // camera.h:120-124, from the PRIMARY ray
        const V3<T> sky = {RT_FMA(w2, (T)0.5, w1), RT_FMA(w2, (T)0.7, w1), RT_FMA(w2, (T)1.0, w1)};
        col = {st.atten.x * sky.x, st.atten.y * sky.y, st.atten.z * sky.z};
        return SF_TERMINATED;
    }
    PATH_STAT(PS_SHADE_HIT);
    T rec[12];
    if (p.shade_in_lds) {
#pragma unroll
        for (int k = 0; k < 12; ++k) rec[k] = lds_shade[12 * hit + k];
    } else {
        const T* tbl = screen_of(p).shade_tbl;
#pragma unroll
        for (int k = 0; k < 12; ++k) rec[k] = tbl[12 * (size_t)hit + k];
    }
    const V3<T> C = {rec[0], rec[1], rec[2]};
    const T inv_r = rec[3];
    const V3<T> P = madd3(closest, D, O);                                    // hittable.h:59-63, :21-26
    const V3<T> outward = {inv_r * (P.x - C.x), inv_r * (P.y - C.y), inv_r * (P.z - C.z)};
    const bool front = dot3(D, outward) < (T)0;
    const V3<T> nrm = front ? outward : V3<T>{-outward.x, -outward.y, -outward.z};
    const int mtype = (int)rec[10];
    if (mtype == RTIOW_DIELECTRIC) {                                     // material.h:68-89
        PATH_STAT(PS_DIELECTRIC);
        V3<T> nd;
        const T ri = front ? rec[9] : rec[8];
        const V3<T> ud = unit3(D);
        const T cos_theta = Real<T>::fmin(-dot3(ud, nrm), (T)1);
        const T sin_theta = sqrt_wave_checked(RT_FMA(-cos_theta, cos_theta, (T)1));
        bool reflect_it = ri * sin_theta > (T)1;
        if (!reflect_it) {
            const T r0 = front ? rec[4] : rec[5];
            const float x = (float)((T)1 - cos_theta);
            const float x2 = x * x;
            const float p5 = (x2 * x2) * x;
            PATH_STAT(PS_SCHLICK_DRAW);
            const T refl = RT_FMA((T)1 - r0, (T)p5, r0);
            reflect_it = refl > Real<T>::uniform(st.rs);
        }
        if (reflect_it) {
            nd = reflect3(ud, nrm);
        } else {
            const V3<T> perp = scale3(ri, madd3(cos_theta, nrm, ud));
            const T k = -sqrt_wave_checked(Real<T>::fabs((T)1 - dot3(perp, perp)));
            nd = madd3(k, nrm, perp);
        }
        st.atten = {st.atten.x * (T)1, st.atten.y * (T)1, st.atten.z * (T)1};   // camera.h:110-115 with attenuation (1,1,1)
        st.O = P; st.D = nd;
        ++st.depth;
        return SF_CONTINUE;
    }
    sc.nrm = nrm; sc.fuzz = rec[7]; sc.mtype = mtype;
    return SF_NEED_RUV;
}

// (x, y, z), lensq: the accepted candidate of random_unit_vector and its squared length (vec3.h:123).  False: the metal absorbed the ray (camera.h:117).
template <class T>
__device__ __forceinline__ bool shade_back(const RenderParams<T>& p, const T* lds_shade, PathState<T>& st, const ShadeCarry<T>& sc, T closest, int hit, T x, T y, T z, T lensq) {
    const T inv = inv_sqrt_accepted(lensq);                              // vec3.h:126
    const V3<T> ruv = {inv * x, inv * y, inv * z};
    const V3<T> nrm = sc.nrm;
    V3<T> nd;
    bool ok = true;
    if (sc.mtype == RTIOW_LAMBERTIAN) {                                  // material.h:38-49
        nd = {nrm.x + ruv.x, nrm.y + ruv.y, nrm.z + ruv.z};
        const T e = Real<T>::near_zero;
        if ((int)(Real<T>::fabs(nd.x) < e) & (int)(Real<T>::fabs(nd.y) < e) & (int)(Real<T>::fabs(nd.z) < e)) nd = nrm;
    } else {                                                             // material.h:51-59
        PATH_STAT(PS_METAL);
        const V3<T> ur = unit3(reflect3(st.D, nrm));
        nd = madd3(sc.fuzz, ruv, ur);
        ok = dot3(nd, nrm) > (T)0;
    }
    if (!ok) return false;                                               // camera.h:117
    T a0, a1, a2;
    if (p.shade_in_lds) { a0 = lds_shade[12 * hit + 4]; a1 = lds_shade[12 * hit + 5]; a2 = lds_shade[12 * hit + 6]; }
    else { const T* tbl = screen_of(p).shade_tbl; a0 = tbl[12 * (size_t)hit + 4]; a1 = tbl[12 * (size_t)hit + 5]; a2 = tbl[12 * (size_t)hit + 6]; }
    st.atten = {st.atten.x * a0, st.atten.y * a1, st.atten.z * a2};      // camera.h:110-115
    st.O = madd3(closest, st.D, st.O);                                   // the hit point again (hittable.h:59), from the unchanged ray
    st.D = nd;
    ++st.depth;
    return true;
}

// One path segment (one trip of the loop at camera.h:84) done by the lane alone.
template <class T, int SRC>
__device__ __forceinline__ bool segment_step(const RenderParams<T>& p, const T* lds_geom, const T* lds_shade, PathState<T>& st, V3<T>& col) {
    if (st.depth >= p.B) { col = {0, 0, 0}; return true; }   // camera.h:127 (also B <= 0)
    // ---------------- hit_world (hittable.h:80-98), nearest (t, index) only
    T closest = __builtin_huge_val();
    int hit = -1;
    const T a = dot3(st.D, st.D);                 // hittable.h:43, ray-invariant
    hit_world<T, SRC>(p, lds_geom, st.O, st.D, a, closest, hit);
    bool retry;
    return shade_step<T, false>(p, lds_shade, st, closest, hit, col, 0, retry);
}

}  // namespace
