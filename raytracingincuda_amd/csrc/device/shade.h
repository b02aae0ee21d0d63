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

// What a lane keeps while it waits for its random_unit_vector (carry-over form of the persistent kernels,
// random_unit_vector_rounds): the normal and the fuzz -- negative for a lambertian, which has none.  A metal's unit
// reflected direction waits in st.D (the incoming direction is spent once the hit record is complete).
template <class T> struct PendingScatter { V3<T> nrm; T fuzz; };

enum { SHADE_CONTINUES = 0, SHADE_ENDED = 1, SHADE_NEEDS_UNIT_VECTOR = 2 };

// Everything after hit_world in one trip of the loop at camera.h:84, up to the material's random_unit_vector: sky
// on a miss (camera.h:120-124: SHADE_ENDED, `col` is the path's colour), else hit record + scatter
// (camera.h:88-117).  A dielectric is complete here (SHADE_CONTINUES: the next segment is in st); a lambertian or a
// metal returns SHADE_NEEDS_UNIT_VECTOR with origin and attenuation already advanced -- attenuation is applied to a
// path that continues (camera.h:110-115), and a metal that ends instead (material.h:58) ends black, whatever it holds.
template <class T>
__device__ __forceinline__ int shade_begin(const RenderParams<T>& p, const T* lds_shade, PathState<T>& st, T closest, int hit, V3<T>& col, PendingScatter<T>& pend) {
    col = {0, 0, 0};
    const V3<T> O = st.O, D = st.D;
    if (hit < 0) {
        PATH_STAT(PS_SKY);
        // ------------ sky, from the PRIMARY ray (camera.h:120-124)
        const double a_sky = 0.5 * ((double)st.sky_uy + 1.0);
        const T w1 = (T)(1.0 - a_sky), w2 = (T)a_sky;
        const V3<T> sky = {RT_FMA(w2, (T)0.5, w1), RT_FMA(w2, (T)0.7, w1), RT_FMA(w2, (T)1.0, w1)};
        col = {st.atten.x * sky.x, st.atten.y * sky.y, st.atten.z * sky.z};
        return SHADE_ENDED;
    }
    // ------------ complete the hit record (hittable.h:59-63, :21-26)
    // one 12-word record per sphere; LDS copy when it fits (no global-load latency on the
    // critical path of the drain tail), else through L1/L2
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
    const V3<T> P = madd3(closest, D, O);
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
            const T r0 = front ? rec[4] : rec[5];                       // material.h:62-66: ((1 - ri) / (1 + ri))^2, computed by upload_scene in T
            const float x = (float)((T)1 - cos_theta);
            const float x2 = x * x;
            const float p5 = (x2 * x2) * x;                              // powf(x,5), see DESIGN.md
            PATH_STAT(PS_SCHLICK_DRAW);
            const T refl = RT_FMA((T)1 - r0, (T)p5, r0);
            reflect_it = refl > Real<T>::uniform(st.rs);
        }
        if (reflect_it) {
            nd = reflect3(ud, nrm);
        } else {                                                         // vec3.h:133-138
            const V3<T> perp = scale3(ri, madd3(cos_theta, nrm, ud));
            const T k = -sqrt_wave_checked(Real<T>::fabs((T)1 - dot3(perp, perp)));
            nd = madd3(k, nrm, perp);
        }
        st.O = P; st.D = nd;                                             // camera.h:110-115 with attenuation (1,1,1): x * 1 is x, bit for bit
        ++st.depth;
        return SHADE_CONTINUES;
    }
    st.atten = {st.atten.x * rec[4], st.atten.y * rec[5], st.atten.z * rec[6]};   // camera.h:110-115
    st.O = P;
    pend.nrm = nrm;
    if (mtype == RTIOW_LAMBERTIAN) {
        pend.fuzz = (T)-1;
    } else {                                                             // material.h:51-59: unit_vector(reflect(...)) does not depend on the draws
        PATH_STAT(PS_METAL);
        pend.fuzz = rec[7];                                              // material.h:29-30: in [0, 1]
        st.D = unit3(reflect3(D, nrm));
    }
    return SHADE_NEEDS_UNIT_VECTOR;
}

// The rest of lambertian_scatter / metal_scatter once the unit vector is there (material.h:38-59).  Returns true when
// the path ended (a metal scattering below the surface: black, camera.h:117).
template <class T>
__device__ __forceinline__ bool shade_finish(PathState<T>& st, const PendingScatter<T>& pend, V3<T> ruv) {
    const V3<T> nrm = pend.nrm;
    V3<T> nd;
    if (pend.fuzz < (T)0) {                                              // material.h:38-49
        nd = {nrm.x + ruv.x, nrm.y + ruv.y, nrm.z + ruv.z};
        const T e = Real<T>::near_zero;
        if (Real<T>::fabs(nd.x) < e && Real<T>::fabs(nd.y) < e && Real<T>::fabs(nd.z) < e) nd = nrm;
    } else {                                                             // material.h:51-59
        nd = madd3(pend.fuzz, ruv, st.D);
        if (!(dot3(nd, nrm) > (T)0)) return true;
    }
    st.D = nd;
    ++st.depth;
    return false;
}

// The whole step in one go (static schedule, one lane = one pixel: nothing to carry over).
template <class T>
__device__ __forceinline__ bool shade_step(const RenderParams<T>& p, const T* lds_shade, PathState<T>& st, T closest, int hit, V3<T>& col) {
    PendingScatter<T> pend;
    const int r = shade_begin<T>(p, lds_shade, st, closest, hit, col, pend);
    if (r != SHADE_NEEDS_UNIT_VECTOR) return r == SHADE_ENDED;
    RT_PROBE_RUV(T, st.rs, 0x7fffffff);
    return shade_finish<T>(st, pend, random_unit_vector<T>(st.rs));
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
    return shade_step<T>(p, lds_shade, st, closest, hit, col);
}

}  // namespace
