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

// Everything after hit_world in one trip of the loop at camera.h:84: sky on a miss
// (camera.h:120-124), else hit record + scatter (camera.h:88-117).  Returns true when the
// path ended; `col` is then its colour.
template <class T>
__device__ __forceinline__ bool shade_step(const RenderParams<T>& p, const T* lds_shade, PathState<T>& st, T closest, int hit, V3<T>& col) {
    col = {0, 0, 0};
    const V3<T> O = st.O, D = st.D;
    if (hit < 0) {
        PATH_STAT(PS_SKY);
        // ------------ sky, from the PRIMARY ray (camera.h:120-124)
        const double a_sky = 0.5 * ((double)st.sky_uy + 1.0);
        const T w1 = (T)(1.0 - a_sky), w2 = (T)a_sky;
        const V3<T> sky = {RT_FMA(w2, (T)0.5, w1), RT_FMA(w2, (T)0.7, w1), RT_FMA(w2, (T)1.0, w1)};
        col = {st.atten.x * sky.x, st.atten.y * sky.y, st.atten.z * sky.z};
        return true;
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
    } else {
        RT_PROBE_RUV(T, st.rs);
        const V3<T> ruv = random_unit_vector<T>(st.rs);
        if (mtype == RTIOW_LAMBERTIAN) {                                 // material.h:38-49
            nd = {nrm.x + ruv.x, nrm.y + ruv.y, nrm.z + ruv.z};
            const T e = Real<T>::near_zero;
            if (Real<T>::fabs(nd.x) < e && Real<T>::fabs(nd.y) < e && Real<T>::fabs(nd.z) < e) nd = nrm;
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
