// hit_loop.h -- brute-force hit_world (hittable.h:80-98): packed discriminants, exact loop, screened loop
// Part of the single gfx950 translation unit rtiow_hip.hip (included there, in this order; internal linkage).
#pragma once
#include "roots.h"

namespace {

// First half of hit_sphere (hittable.h:42-47) for the four spheres s..s+3 of one trip:
// h = d.oc and disc = h*h - a*c, each element with exactly the reference's operation sequence.
//
// fp32: the table is PAIR-INTERLEAVED -- {cxA,cxB, cyA,cyB, czA,czB, r2A,r2B} per pair of
// spheres -- so the twelve operations run as v_pk_add/mul/fma_f32 on two spheres at once:
// 24 packed VALU per trip instead of 48 (per-element IEEE results are unchanged).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <class T> struct Trip { T h0, h1, h2, h3, d0, d1, d2, d3; };

// The ray as the sphere loop wants it.  The packed instructions take the ray as their second operand for BOTH spheres of a pair:
// RTIOW_RAY_BROADCAST (default, round 5) names the component's own register as the low half of a 64-bit operand whose high half is never
// read -- `op_sel_hi` 0 makes the instruction use the low half for both results -- so no copy of the ray is built at all.  The compiler
// folds a splat like that by itself when it is made in the basic block that uses it; here the splats are loop-invariant, get hoisted in
// front of the sphere loop and were materialised there with 14-16 v_mov_b32 per path segment (7 to pin the components, 7 for the high
// halves: RTIOW_RAY_BROADCAST=0, the form of rounds 1-4).  fp64 keeps plain scalars.
#ifndef RTIOW_RAY_BROADCAST
#define RTIOW_RAY_BROADCAST 1
#endif
template <class T> struct LoopRay;
#if RTIOW_RAY_BROADCAST
template <> struct LoopRay<float> { float ox, oy, oz, dx, dy, dz, a; };
__device__ __forceinline__ LoopRay<float> make_loop_ray(float ox, float oy, float oz, float dx, float dy, float dz, float a) {
    return {ox, oy, oz, dx, dy, dz, a};
}
// x in the low half of a register pair, the high half undefined (never read)
__device__ __forceinline__ v2f low_half(float x) {
    v2f t;          // .y deliberately left unset
    t.x = x;
    return t;
}
// {c.x - s, c.y - s}, {s * v.x, s * v.y}, {fma(s, v.x, w.x), fma(s, v.y, w.y)}: one IEEE operation per element, as the plain forms
__device__ __forceinline__ v2f pk_sub_lane(v2f c, float s) {
    v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(c), "v"(low_half(s))); return r;
}
__device__ __forceinline__ v2f pk_mul_lane(float s, v2f v) {
    v2f r; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(low_half(s)), "v"(v)); return r;
}
__device__ __forceinline__ v2f pk_fma_lane(float s, v2f v, v2f w) {
    v2f r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(low_half(s)), "v"(v), "v"(w)); return r;
}
// The trip's ray as six opaque registers, pinned IN PLACE where the trip's ray is final (persistent_body): without it the vectoriser keeps
// O and D as <3 x float> values, and taking one component of those as the low half of a pair goes through scratch memory.  The pinned
// values replace the state's own, so the pin costs no copy (make_loop_ray's pin of rounds 1-4 kept both alive: 7 v_mov per trip).
__device__ __forceinline__ void pin_ray(V3<float>& O, V3<float>& D) { asm volatile("" : "+v"(O.x), "+v"(O.y), "+v"(O.z), "+v"(D.x), "+v"(D.y), "+v"(D.z)); }
__device__ __forceinline__ void pin_ray(V3<double>&, V3<double>&) {}
#else
__device__ __forceinline__ void pin_ray(V3<float>&, V3<float>&) {}
__device__ __forceinline__ void pin_ray(V3<double>&, V3<double>&) {}
template <> struct LoopRay<float> { v2f ox, oy, oz, dx, dy, dz, aa; float a; };
__device__ __forceinline__ LoopRay<float> make_loop_ray(float ox, float oy, float oz, float dx, float dy, float dz, float a) {
    // The empty asm makes each component an opaque VGPR value, so the splats are built with
    // register moves (hipcc otherwise round-trips the ray through scratch to form the pairs).
    asm volatile("" : "+v"(ox), "+v"(oy), "+v"(oz), "+v"(dx), "+v"(dy), "+v"(dz), "+v"(a));
    LoopRay<float> r;
    r.ox.x = ox; r.ox.y = ox; r.oy.x = oy; r.oy.y = oy; r.oz.x = oz; r.oz.y = oz;
    r.dx.x = dx; r.dx.y = dx; r.dy.x = dy; r.dy.y = dy; r.dz.x = dz; r.dz.y = dz;
    r.aa.x = a; r.aa.y = a; r.a = a;
    return r;
}
#endif
template <> struct LoopRay<double> { double ox, oy, oz, dx, dy, dz, a; };
__device__ __forceinline__ LoopRay<double> make_loop_ray(double ox, double oy, double oz, double dx, double dy, double dz, double a) {
    return {ox, oy, oz, dx, dy, dz, a};
}

__device__ __forceinline__ void pair_discriminants(v4f lo, v4f hi, const LoopRay<float>& r, v2f& hh, v2f& dd) {
    const v2f cx = {lo.x, lo.y}, cy = {lo.z, lo.w}, cz = {hi.x, hi.y}, r2 = {hi.z, hi.w};
#if RTIOW_RAY_BROADCAST
    const v2f ocx = pk_sub_lane(cx, r.ox), ocy = pk_sub_lane(cy, r.oy), ocz = pk_sub_lane(cz, r.oz);  // :42
    hh = pk_fma_lane(r.dz, ocz, pk_fma_lane(r.dy, ocy, pk_mul_lane(r.dx, ocx)));                       // :44
    const v2f c = __builtin_elementwise_fma(ocz, ocz, __builtin_elementwise_fma(ocy, ocy, ocx * ocx)) - r2;   // :45
    dd = __builtin_elementwise_fma(hh, hh, -pk_mul_lane(r.a, c));                                     // :47
#else
    const v2f ocx = cx - r.ox, ocy = cy - r.oy, ocz = cz - r.oz;                                    // :42
    hh = __builtin_elementwise_fma(r.dz, ocz, __builtin_elementwise_fma(r.dy, ocy, r.dx * ocx));    // :44
    const v2f c = __builtin_elementwise_fma(ocz, ocz, __builtin_elementwise_fma(ocy, ocy, ocx * ocx)) - r2;   // :45
    dd = __builtin_elementwise_fma(hh, hh, -(r.aa * c));                                            // :47
#endif
}

__device__ __forceinline__ Trip<float> trip_discriminants(const float* g, int s, const LoopRay<float>& r) {
    const v4f* g4 = reinterpret_cast<const v4f*>(g + 4 * s);
    const v4f p0 = g4[0], p1 = g4[1], p2 = g4[2], p3 = g4[3];
    v2f ha, da, hb, db;
    pair_discriminants(p0, p1, r, ha, da);
    pair_discriminants(p2, p3, r, hb, db);
    return {ha.x, ha.y, hb.x, hb.y, da.x, da.y, db.x, db.y};
}

// fp64: plain {cx,cy,cz,r2} per sphere (no packed f64 on gfx950).
__device__ __forceinline__ void sphere_discriminant(const double* g, int s, const LoopRay<double>& r, double& h, double& disc) {
    const double cx = g[4 * s + 0], cy = g[4 * s + 1], cz = g[4 * s + 2], r2 = g[4 * s + 3];
    const double ocx = cx - r.ox, ocy = cy - r.oy, ocz = cz - r.oz;                    // :42
    h = __builtin_fma(r.dz, ocz, __builtin_fma(r.dy, ocy, r.dx * ocx));                // :44
    const double c = __builtin_fma(ocz, ocz, __builtin_fma(ocy, ocy, ocx * ocx)) - r2; // :45
    disc = __builtin_fma(h, h, -(r.a * c));                                            // :47
}
__device__ __forceinline__ Trip<double> trip_discriminants(const double* g, int s, const LoopRay<double>& r) {
    Trip<double> t;
    sphere_discriminant(g, s + 0, r, t.h0, t.d0);
    sphere_discriminant(g, s + 1, r, t.h1, t.d1);
    sphere_discriminant(g, s + 2, r, t.h2, t.d2);
    sphere_discriminant(g, s + 3, r, t.h3, t.d3);
    return t;
}

// One trip = four spheres: discriminants, ONE wave-level branch on max(disc0..3) >= 0 (a lane
// reaches a sphere's line in only ~4 % of the trips for the reference scenes), and the IEEE
// sqrt/divide tail only inside it, in index order.
template <class T>
__device__ __forceinline__ void sphere_trip(const T* g, int s, const LoopRay<T>& r, T& closest, int& hit) {
    const Trip<T> t = trip_discriminants(g, s, r);
    const T m = Real<T>::fmax(Real<T>::fmax(t.d0, t.d1), Real<T>::fmax(t.d2, t.d3));
    if (m >= (T)0) {                                                          // :48 for any of the four
        if (t.d0 >= (T)0) finish_sphere_test<T>(s + 0, t.h0, t.d0, r.a, closest, hit);
        if (t.d1 >= (T)0) finish_sphere_test<T>(s + 1, t.h1, t.d1, r.a, closest, hit);
        if (t.d2 >= (T)0) finish_sphere_test<T>(s + 2, t.h2, t.d2, r.a, closest, hit);
        if (t.d3 >= (T)0) finish_sphere_test<T>(s + 3, t.h3, t.d3, r.a, closest, hit);
    }
}

// hit_world (hittable.h:80-98): every sphere tested exactly, in index order.  The table is
// padded to a multiple of 4 with never-hit entries (r^2 = -1e12 => disc < 0).
template <class T, int SRC>
__device__ __forceinline__ void hit_world_direct(const RenderParams<T>& p, const T* lds_geom, V3<T> O, V3<T> D, T a,
                                                 T& closest, int& hit) {
    const T* g = (SRC == RTIOW_SCENE_LDS) ? lds_geom : p.geom_a;
    const LoopRay<T> r = make_loop_ray(O.x, O.y, O.z, D.x, D.y, D.z, a);
    for (int s = 0; s < p.n_padded; s += 4) sphere_trip<T>(g, s, r, closest, hit);
}

// hit_sphere for ONE sphere, scalar, exactly the reference's arithmetic (used by the screened
// loop for its rare candidates; the table is the pair-interleaved fp32 one).
__device__ __forceinline__ void exact_sphere_test_f32(const float* g, int s, V3<float> O, V3<float> D, float a, float& closest, int& hit) {
    PATH_STAT(PS_EXACT_BLOCK);
    const int base = (s >> 1) * 8 + (s & 1);
    const float cx = g[base], cy = g[base + 2], cz = g[base + 4], r2 = g[base + 6];
    const float ocx = cx - O.x, ocy = cy - O.y, ocz = cz - O.z;                                   // :42
    const float h = __builtin_fmaf(D.z, ocz, __builtin_fmaf(D.y, ocy, D.x * ocx));                 // :44
    const float c = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, ocx * ocx)) - r2;            // :45
    const float disc = __builtin_fmaf(h, h, -(a * c));                                            // :47
    if (disc >= 0.0f) finish_sphere_test<float>(s, h, disc, a, closest, hit);                      // :48-57
}

// hit_world with an 8-operation SCREEN (packed fp32) in front of the reference's 12-operation test.
//
// In exact arithmetic h = d.(C-O) = d.C - d.O and c = |C-O|^2 - r^2 = (|C|^2 - r^2) + |O|^2 - 2 O.C,
// so with the per-ray constants k1 = d.O', k2 = |O'|^2, m = -2 O' (O' = O - centre) and the
// per-sphere constant q = |C'|^2 - r^2 (C' = C - centre, precomputed) a sphere costs
//     h~ = fma(dz,Cz', fma(dy,Cy', fma(dx,Cx', -k1)))      3
//     c~ = fma(mz,Cz', fma(my,Cy', fma(mx,Cx', q + k2)))   4
//     disc~ = fma(h~,h~, -c~)        (d pre-scaled to unit length, so a = 1)   1      = 8 instead of 12
// (8 v_pk per PAIR of spheres).  The unit direction uses the raw v_rsq (2^-22): only the sign of
// disc~ matters and disc/a has the same sign as disc, the rsq error is covered by the margin.
// disc~ is NOT the reference's discriminant (different roundings, cancellation), so it only
// SCREENS: with E = 2^-18 a ((|C'|+|O'|)^2 + r^2) bounding |disc~ - Disc| + |disc_ref - Disc|
// (derivation in DESIGN.md, constant 45u of slack-free bound vs 64u used), twice that margin is
// subtracted from c~: the sphere's share 2^-17(|C'|^2 + r^2) is baked into q' by the host, the
// ray's share 2^-17(2 Cmax |O'| + |O'|^2) (Cmax = max |C'| over the screened spheres) is folded
// into k2, so that        disc_ref >= 0   =>   disc~' >= 0        for every ray and sphere.
// A sphere with disc~' < 0 therefore fails the reference's `discriminant < 0` test (hittable.h:48)
// and is skipped like there; every other sphere is re-tested with the reference's exact
// arithmetic (exact_sphere_test_f32), in index order.  Spheres the bound would make useless
// (|C'| > 64: the ground) get q' = -1e30 and are always re-tested.  Result: bit-identical.
__device__ __forceinline__ void exact_sphere_test_f64(const double* g, int s, V3<double> O, V3<double> D, double a, double& closest, int& hit) {
    const double cx = g[4 * s], cy = g[4 * s + 1], cz = g[4 * s + 2], r2 = g[4 * s + 3];
    const double ocx = cx - O.x, ocy = cy - O.y, ocz = cz - O.z;                                   // :42
    const double h = __builtin_fma(D.z, ocz, __builtin_fma(D.y, ocy, D.x * ocx));                  // :44
    const double c = __builtin_fma(ocz, ocz, __builtin_fma(ocy, ocy, ocx * ocx)) - r2;             // :45
    const double disc = __builtin_fma(h, h, -(a * c));                                             // :47
    if (disc >= 0.0) finish_sphere_test<double>(s, h, disc, a, closest, hit);                      // :48-57
}

__device__ __forceinline__ void exact_sphere_test(const float* g, int s, V3<float> O, V3<float> D, float a, float& closest, int& hit) {
    exact_sphere_test_f32(g, s, O, D, a, closest, hit);
}
__device__ __forceinline__ void exact_sphere_test(const double* g, int s, V3<double> O, V3<double> D, double a, double& closest, int& hit) {
    exact_sphere_test_f64(g, s, O, D, a, closest, hit);
}

// The screen itself always runs in packed fp32, for both precisions: it only has to be
// conservative.  fp64 rays are rounded to fp32 first (one more 2^-24 relative perturbation of O'
// and d, of the kind the margin already covers for the recentring), and the fp64 reference
// discriminant carries ~2^-53 instead of 18 * 2^-24 of rounding, so the fp32 margins hold a
// fortiori; candidates are re-tested with the exact fp64 arithmetic.  21 instead of 36 issue
// cycles per sphere (v_pk_fma_f32 vs v_fma_f64, bin/valu_cost).
template <class T>
__device__ __forceinline__ void hit_world_screened(const RenderParams<T>& p, const T* lds_exact, const float* lds_screen,
                                                   V3<T> O, V3<T> D, T a, T& closest, int& hit) {
    const auto& sc = screen_of(p);
    float ox = (float)(O.x - (T)sc.ctr_x), oy = (float)(O.y - (T)sc.ctr_y), oz = (float)(O.z - (T)sc.ctr_z);
    float dx = (float)D.x, dy = (float)D.y, dz = (float)D.z;
    const float af = sizeof(T) == 4 ? (float)a : __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    const float rs = __builtin_amdgcn_rsqf(af);   // screen only: |d^| = 1 +- 2^-22
    dx *= rs; dy *= rs; dz *= rs;
    float nk1 = -__builtin_fmaf(dz, oz, __builtin_fmaf(dy, oy, dx * ox));
    float k2 = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
    k2 = k2 - 7.62939453125e-06f * __builtin_fmaf((float)(T)sc.omax2, fast_sqrt(k2), k2);   // - 2^-17 (2 Cmax |O'| + |O'|^2); omax2 holds 2 Cmax (1 + 2^-20): raw sqrt
    float mx = -2.0f * ox, my = -2.0f * oy, mz = -2.0f * oz;
    asm volatile("" : "+v"(nk1), "+v"(k2), "+v"(mx), "+v"(my), "+v"(mz), "+v"(dx), "+v"(dy), "+v"(dz));
    const v2f vnk1 = {nk1, nk1}, vk2 = {k2, k2}, vmx = {mx, mx}, vmy = {my, my}, vmz = {mz, mz};
    const v2f vdx = {dx, dx}, vdy = {dy, dy}, vdz = {dz, dz};
    for (int s = 0; s < p.n_padded; s += 4) {
        const v4f* g4 = reinterpret_cast<const v4f*>(lds_screen + 4 * s);
        const v4f p0 = g4[0], p1 = g4[1], p2 = g4[2], p3 = g4[3];
        v2f dsc[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const v4f lo = q ? p2 : p0, hi = q ? p3 : p1;
            const v2f cx = {lo.x, lo.y}, cy = {lo.z, lo.w}, cz = {hi.x, hi.y}, qq = {hi.z, hi.w};
            const v2f hh = __builtin_elementwise_fma(vdz, cz, __builtin_elementwise_fma(vdy, cy, __builtin_elementwise_fma(vdx, cx, vnk1)));
            const v2f cc = __builtin_elementwise_fma(vmz, cz, __builtin_elementwise_fma(vmy, cy, __builtin_elementwise_fma(vmx, cx, qq + vk2)));
            dsc[q] = __builtin_elementwise_fma(hh, hh, -cc);
        }
        const float m = __builtin_fmaxf(__builtin_fmaxf(dsc[0].x, dsc[0].y), __builtin_fmaxf(dsc[1].x, dsc[1].y));
        if (!(m < 0.0f)) {                        // some sphere of the trip may pass hittable.h:48 (NaNs are kept)
            if (!(dsc[0].x < 0.0f)) exact_sphere_test(lds_exact, s + 0, O, D, a, closest, hit);
            if (!(dsc[0].y < 0.0f)) exact_sphere_test(lds_exact, s + 1, O, D, a, closest, hit);
            if (!(dsc[1].x < 0.0f)) exact_sphere_test(lds_exact, s + 2, O, D, a, closest, hit);
            if (!(dsc[1].y < 0.0f)) exact_sphere_test(lds_exact, s + 3, O, D, a, closest, hit);
        }
    }
}

}  // namespace
