// hit_grid.h -- hit_world over the uniform grid (RTIOW_SCENE_GRID) and the hit_world dispatch
// Part of the single gfx950 translation unit rtiow_hip.hip (included there, in this order; internal linkage).
#pragma once
#include "hit_loop.h"

namespace {

// =====================================================================================
// hit_world over a uniform grid (RTIOW_SCENE_GRID, the default).
//
// The screen above still costs every ray 8 operations per sphere.  The host therefore also bins
// the SMALL spheres of the scene into a 2-D grid of cells over x/z (one slab in y), at most four
// per cell (build_grid_tables); a lane walks only the cells its own ray crosses while it is inside
// the slab and tests their spheres with the reference's exact arithmetic.  Spheres that do not fit
// a cell (the ground, the three unit spheres) or overflow a full one form the DIRECT list, which
// every ray tests exactly in packed trips first.  Measured on the headline scene a wave walks 1.7
// cells per iteration (its longest lane) instead of screening 125 spheres.
//
// Why the result is unchanged.  The reference's nearest hit is the lexicographic minimum of
// (t, index) over the spheres whose hit_sphere succeeds; a sphere's own candidate root does not
// depend on the others (finish_sphere_test).  It therefore suffices that every sphere the
// reference COULD accept is tested, with the reference's arithmetic and the ANYORDER tie rule:
//  * hit_sphere can only succeed if its computed discriminant is >= 0, and that discriminant
//    differs from the real-number one by at most E = 18u(|oc|^2 + r^2) (u = 2^-24),
//    so the ray's LINE passes within sqrt(r^2 + E) of the centre, and the point at the accepted
//    root lies inside that inflated ball (its squared distance from the centre is r^2 + (computed
//    - real discriminant));
//  * for origins within sqrt(far2) of the scene centre the host bounds E once and registers
//    sphere i in every cell that its bounding square inflated to sqrt(r_i^2 + E) + eps touches;
//    eps (2^-16 of the largest coordinate in play, >= 25x the rounding of the walk below) lets the
//    walk be computed in plain fp32 with raw reciprocals: the cells it visits stay within eps of
//    the true ray, and every point of the true ray inside an inflated ball has that sphere
//    registered in every cell within eps of it;
//  * the walk is clipped to the box of the registered (inflated) spheres and to t >= 0 (a sphere
//    behind the origin has both roots < tmin unless the origin is inside it, and then it is
//    registered in the origin's cell);
//  * the walk stops once the next cell boundary lies beyond the nearest accepted root: every
//    sphere not registered in a visited cell has all its candidate points more than eps beyond
//    that boundary, so its root is larger;
//  * rays that start FARTHER away (a bounce off the ground plane hundreds of units out: E grows
//    with |oc|^2) are clipped against the box of the gridded CENTRES inflated by their own
//    sqrt(rmax^2 + E(ray)): if the line misses it no gridded sphere can be accepted, otherwise
//    (a far ray skimming the scene, < 0.01 % of the rays) the whole wave takes the screened
//    brute-force loop above for this one segment.  NaN / zero / huge rays go the same way.
// Tested bit for bit against the exact loop on full frames of every scene, both precisions, and
// on random scenes (tests/test_gpu_parity.py).
// =====================================================================================
template <class T>
__device__ __forceinline__ void direct_trip(const T* g, const int* ids, int s, const LoopRay<T>& r, T& closest, int& hit, FastDiv<T> fd) {
    const Trip<T> t = trip_discriminants(g, s, r);
    // no common guard: some lane has a candidate on the direct list (the ground) in nearly every trip
    if (t.d0 >= (T)0) finish_sphere_test<T, true>(ids[s + 0], t.h0, t.d0, r.a, closest, hit, fd);
    if (t.d1 >= (T)0) finish_sphere_test<T, true>(ids[s + 1], t.h1, t.d1, r.a, closest, hit, fd);
    if (t.d2 >= (T)0) finish_sphere_test<T, true>(ids[s + 2], t.h2, t.d2, r.a, closest, hit, fd);
    if (t.d3 >= (T)0) finish_sphere_test<T, true>(ids[s + 3], t.h3, t.d3, r.a, closest, hit, fd);
}

// {cx, cy, cz, r*r} of sphere i for the per-lane gathers of the walk: fp32 from the AoS copy in the
// grid blob (one ds_read_b128), fp64 from geom_a, which is AoS already.
__device__ __forceinline__ void load_sphere(const float* aos, int i, float& cx, float& cy, float& cz, float& r2) {
    const v4f c = reinterpret_cast<const v4f*>(aos)[i];
    cx = c.x; cy = c.y; cz = c.z; r2 = c.w;
}
__device__ __forceinline__ void load_sphere(const double* aos, int i, double& cx, double& cy, double& cz, double& r2) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d lo = reinterpret_cast<const v2d*>(aos)[2 * i], hi = reinterpret_cast<const v2d*>(aos)[2 * i + 1];
    cx = lo.x; cy = lo.y; cz = hi.x; r2 = hi.y;
}

// The (up to) four spheres of one cell, hittable.h:42-57 each, for this lane's own ray.
// Cells fill from slot 0 and pad with the never-hit entry (index n, `pad2` = n in both halves of a word): 35 % of the headline
// scene's occupied cells hold one sphere, 43 % two, 18 % three, 5 % four (29 / 43 / 24 / 4 % on the 487-sphere scene).  Slots 2-3 are
// therefore tested behind ONE wave-level branch: only when some lane of this step holds a third sphere -- 24 operations and
// two gathers less for the late steps of a walk, which a handful of lanes run (RTIOW_CELL_PAIRS, default on; 0 = every slot always).
#ifndef RTIOW_CELL_PAIRS
#define RTIOW_CELL_PAIRS 2
#endif
template <class T>
__device__ __forceinline__ void cell_tests(const T* aos, unsigned rec_lo, unsigned rec_hi, unsigned pad2, V3<T> O, V3<T> D, T a, T& closest, int& hit, FastDiv<T> fd) {
    const int id[4] = {(int)(rec_lo & 0xffffu), (int)(rec_lo >> 16), (int)(rec_hi & 0xffffu), (int)(rec_hi >> 16)};
    T h[4], disc[4];
    auto slot = [&](int k) __attribute__((always_inline)) {
        T cx, cy, cz, r2;
        load_sphere(aos, id[k], cx, cy, cz, r2);
        const T ocx = cx - O.x, ocy = cy - O.y, ocz = cz - O.z;                         // :42
        h[k] = RT_FMA(D.z, ocz, RT_FMA(D.y, ocy, D.x * ocx));                            // :44
        const T c = RT_FMA(ocz, ocz, RT_FMA(ocy, ocy, ocx * ocx)) - r2;                  // :45
        disc[k] = RT_FMA(h[k], h[k], -(a * c));                                          // :47
    };
    if (RTIOW_CELL_PAIRS == 2) {
        // pair after pair, each with its own candidate branch: two discriminants live at a time instead of four
        auto pair = [&](int k) __attribute__((always_inline)) {
            slot(k); slot(k + 1);
            if (Real<T>::fmax(disc[k], disc[k + 1]) >= (T)0) {                              // :48 for either of the two
                PATH_STAT(PS_EXACT_BLOCK);
                if (disc[k] >= (T)0) finish_sphere_test<T, true>(id[k], h[k], disc[k], a, closest, hit, fd);
                if (disc[k + 1] >= (T)0) finish_sphere_test<T, true>(id[k + 1], h[k + 1], disc[k + 1], a, closest, hit, fd);
            }
        };
        pair(0);
        if (__builtin_amdgcn_ballot_w64(rec_hi != pad2) != 0) { PATH_STAT(PS_CELL_PAIR2); pair(2); }
        return;
    }
    slot(0); slot(1);
    if (RTIOW_CELL_PAIRS) {
        h[2] = h[3] = (T)0; disc[2] = disc[3] = (T)-1;                                    // what a pad entry amounts to: never a candidate
        if (__builtin_amdgcn_ballot_w64(rec_hi != pad2) != 0) { PATH_STAT(PS_CELL_PAIR2); slot(2); slot(3); }
    } else { slot(2); slot(3); }
    const T m = Real<T>::fmax(Real<T>::fmax(disc[0], disc[1]), Real<T>::fmax(disc[2], disc[3]));
    if (m >= (T)0) {                                                                    // :48 for any of the four
        PATH_STAT(PS_EXACT_BLOCK);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (disc[k] >= (T)0) finish_sphere_test<T, true>(id[k], h[k], disc[k], a, closest, hit, fd);
    }
}

// v_min_f32 without the canonicalising v_max_f32 x, x the compiler puts in front of fminf's operands (IEEE minNum quiets signalling NaNs): the walk's
// parameters are never signalling NaNs, and a quiet NaN is returned or dropped as v_min_f32 does it either way (a NaN ray never reaches the walk).
__device__ __forceinline__ float raw_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Clip of o + t d against [lo, hi] on one axis, folded into [t0, t1].  Raw reciprocal: see eps above.
// Returns the raw reciprocal of d it used (0 for a parallel ray): the walk steps with the same values.
__device__ __forceinline__ float clip_axis(float o, float d, float lo, float hi, float& t0, float& t1) {
#ifdef RTIOW_CLIP_BRANCHES
    if (__builtin_fabsf(d) < 1e-30f) {
        if (!(o >= lo && o <= hi)) t1 = -__builtin_huge_valf();
        return 0.0f;
    }
    const float inv = __builtin_amdgcn_rcpf(d);
    const float ta = (lo - o) * inv, tb = (hi - o) * inv;
    t0 = __builtin_fmaxf(t0, __builtin_fminf(ta, tb));
    t1 = __builtin_fminf(t1, __builtin_fmaxf(ta, tb));
    return inv;
#else
    // The same, predicated: the parallel-ray case is three selects instead of a divergent region per axis (exec-mask
    // save / branch / restore, ~8 scalar instructions each, for a case no camera ray takes).
    const bool par = __builtin_fabsf(d) < 1e-30f;
    const float inv = __builtin_amdgcn_rcpf(d);
    const float ta = (lo - o) * inv, tb = (hi - o) * inv;
    const float n0 = __builtin_fmaxf(t0, __builtin_fminf(ta, tb));
    const float n1 = __builtin_fminf(t1, __builtin_fmaxf(ta, tb));
    const bool inside = o >= lo && o <= hi;
    t0 = par ? t0 : n0;
    t1 = par ? (inside ? t1 : -__builtin_huge_valf()) : n1;
    return par ? 0.0f : inv;
#endif
}

template <class T>
__device__ __forceinline__ void hit_world_grid(const RenderParams<T>& p, const T* lds_exact, const float* lds_screen,
                                               V3<T> O, V3<T> D, T a, T& closest, int& hit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const auto& g = grid_of(p);
    REGION_BEGIN(setup);
    // ---- which rays the registration margins cover
    const auto& sc = screen_of(p);
    const float fx = (float)(O.x - (T)sc.ctr_x), fy = (float)(O.y - (T)sc.ctr_y), fz = (float)(O.z - (T)sc.ctr_z);
    const float k2 = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));
    const float af = (float)a;
    const bool sane = af > 1e-30f && af < 1e30f && k2 < 1e30f;        // false for NaN as well
    const bool near = sane && k2 <= g.far2;
    const float ox = (float)O.x - g.x0, oy = (float)O.y, oz = (float)O.z - g.z0;
    const float dx = (float)D.x, dy = (float)D.y, dz = (float)D.z;
    float t0 = 0.0f, t1 = __builtin_huge_valf();
    float xlo = 0.0f, xhi = (float)g.nx * g.cell, zlo = 0.0f, zhi = (float)g.nz * g.cell, ylo = g.ylo, yhi = g.yhi;
    if (__builtin_amdgcn_ballot_w64(!near) != 0) {
        // a far ray: can any gridded sphere pass the reference's discriminant test at all?  Only if the
        // line comes within rho = sqrt(rmax^2 + E) of a centre, E = 18u((|O'| + Cmax)^2 + rmax^2)
        // (2^-20 instead of 18 * 2^-24 and the 1.001 cover the raw square roots).
        const float reach = fast_sqrt(k2) * 1.001f + g.cmax;
        const float E = 9.5367431640625e-07f * __builtin_fmaf(reach, reach, g.rmax2);
        const float rho = fast_sqrt(g.rmax2 + E) * 1.001f;
        if (!near) {
            xlo = g.core_lo[0] - g.x0 - rho; xhi = g.core_hi[0] - g.x0 + rho;
            ylo = g.core_lo[1] - rho;        yhi = g.core_hi[1] + rho;
            zlo = g.core_lo[2] - g.z0 - rho; zhi = g.core_hi[2] - g.z0 + rho;
        }
    }
    clip_axis(oy, dy, ylo, yhi, t0, t1);
    const float inv_dx = clip_axis(ox, dx, xlo, xhi, t0, t1);
    const float inv_dz = clip_axis(oz, dz, zlo, zhi, t0, t1);
    const bool crosses = !sane || t0 <= t1;
    REGION_END(setup, RG_GRID_SETUP);
    if (__builtin_amdgcn_ballot_w64(!near && crosses) != 0) {
        REGION_BEGIN(fallback);
        hit_world_screened<T>(p, lds_exact, lds_screen, O, D, a, closest, hit);   // exact for every lane of the wave
        REGION_END(fallback, RG_GRID_FALLBACK);
        return;
    }
    // ---- one reciprocal for every quotient of this segment (FastDiv above ieee_roots)
    FastDiv<T> fd = {(T)0, false};
    if (p.range_flags & 2) {
        fd.on = __builtin_amdgcn_ballot_w64(!(a >= (T)0x1p-40 && a <= (T)0x1p40)) == 0;
        fd.ra = refined_reciprocal(a);
    }
    REGION_BEGIN(direct);
    RT_PROBE_DIRECT(T, smem_raw, g, O, D, a, fd);
    // The direct list and the walk, instantiated for both values of fd.on: ONE scalar branch per segment picks the
    // copy, inside it every quotient's form is fixed at compile time (a branch at each of the eight finishing
    // sites cost 6 % more scalar instructions).
    auto direct_list_and_walk = [&](auto fast_tag) __attribute__((always_inline)) {
        const FastDiv<T> fdc = {fd.ra, decltype(fast_tag)::value};
        // ---- the direct list: packed trips, every ray
        {
            const T* dg = reinterpret_cast<const T*>(smem_raw + g.direct_offset);
            const int* ids = reinterpret_cast<const int*>(smem_raw + g.direct_ids_offset);
            const LoopRay<T> r = make_loop_ray(O.x, O.y, O.z, D.x, D.y, D.z, a);
            for (int s = 0; s < g.n_direct_padded; s += 4) direct_trip<T>(dg, ids, s, r, closest, hit, fdc);
        }
        REGION_END(direct, RG_GRID_DIRECT);
        // ---- the walk
        REGION_BEGIN(walk);
        bool walking = near && crosses;
        if (__builtin_amdgcn_ballot_w64(walking) == 0) { REGION_END(walk, RG_GRID_WALK); return; }
        const T* aos = sizeof(T) == 4 ? reinterpret_cast<const T*>(smem_raw + g.aos_offset) : lds_exact;
        const uint2* cells = reinterpret_cast<const uint2*>(smem_raw + g.cells_offset); (void)cells;
        const unsigned pad2 = (unsigned)p.n * 0x10001u;     // a cell record's upper word when slots 2 and 3 are pads
        const float px = __builtin_fmaf(t0, dx, ox), pz = __builtin_fmaf(t0, dz, oz);
        int cx = (int)__builtin_floorf(px * g.inv_cell), cz = (int)__builtin_floorf(pz * g.inv_cell);
        cx = cx < 0 ? 0 : (cx >= g.nx ? g.nx - 1 : cx);
        cz = cz < 0 ? 0 : (cz >= g.nz ? g.nz - 1 : cz);
        const bool step_x = __builtin_fabsf(dx) >= 1e-30f, step_z = __builtin_fabsf(dz) >= 1e-30f;
        const int sx = dx > 0.0f ? 1 : -1, sz = dz > 0.0f ? 1 : -1;
#ifdef RTIOW_PATH_STATS
        int step_no = 0;                                   // which step of its walk the wave is in (instrumented build: profiles/r04/path_stats_walk_steps*.json)
        if (walking) PATH_STAT(PS_WALK);
#endif
#ifndef RTIOW_DDA_INCREMENTAL
#define RTIOW_DDA_INCREMENTAL 1
#endif
#if RTIOW_DDA_INCREMENTAL
        // The classic incremental form (round 5): the parameters at which the ray leaves the current cell, per axis, are carried and ADVANCED by
        // cell * |1 / d| on the axis that steps, and the cell record's LDS address by +-8 bytes / +-8 nx bytes -- 18 vector instructions a step instead
        // of 30 (closed form: two int -> float conversions, two boundary products, the address from cz * nx + cx with a quarter-rate integer
        // multiply).  The advanced parameters differ from the closed form's by one rounding per step taken: after k steps at most k ulps of the
        // parameter, i.e. a position error of k 2^-24 of the distance walked -- the walk takes at most nx + nz <= 128 steps (plan_grid: nx nz <= 4096),
        // so at most 2^-17 of the largest coordinate in play, half of the eps = 2^-16 L every registration is widened by (the clip's own roundings
        // are a few ulps).  Which cell a ray is said to be in near a boundary is therefore still decided within eps of the truth, which is all the
        // exactness argument above asks of the walk.
        const float dtx = step_x ? g.cell * __builtin_fabsf(inv_dx) : 0.0f, dtz = step_z ? g.cell * __builtin_fabsf(inv_dz) : 0.0f;
        float tx = step_x ? ((float)(cx + (sx > 0 ? 1 : 0)) * g.cell - ox) * inv_dx : __builtin_huge_valf();
        float tz = step_z ? ((float)(cz + (sz > 0 ? 1 : 0)) * g.cell - oz) * inv_dz : __builtin_huge_valf();
        int cell_at = (cz * g.nx + cx) * 8;                                   // byte offset of the cell's record
        const int dax = sx * 8, daz = sz * g.nx * 8;
        const unsigned char* cells_b = smem_raw + g.cells_offset;
        while (__builtin_amdgcn_ballot_w64(walking) != 0) {
#ifdef RTIOW_PATH_STATS
            ++step_no;
#endif
            if (walking) {
                PATH_STAT(PS_GRID_STEP);
#ifdef RTIOW_PATH_STATS
                path_stat(step_no == 1 ? PS_STEP_1 : step_no == 2 ? PS_STEP_2 : step_no == 3 ? PS_STEP_3 : step_no == 4 ? PS_STEP_4 : step_no <= 8 ? PS_STEP_5_8 : PS_STEP_9_UP);
#endif
                const uint2 rec = *reinterpret_cast<const uint2*>(cells_b + cell_at);
                if (rec.x != 0xffffffffu) cell_tests<T>(aos, rec.x, rec.y, pad2, O, D, a, closest, hit, fdc);
                const float tnext = raw_min(tx, tz);
                const float tend = raw_min(t1, (float)closest);                    // (float) rounds to nearest: covered by eps
                if (tnext >= tend) walking = false;                               // leaves the slab / the grid, or a nearer hit is known
                else {
                    const bool x_first = tx <= tz;
                    cx += x_first ? sx : 0; cz += x_first ? 0 : sz;
                    cell_at += x_first ? dax : daz;
                    tx += x_first ? dtx : 0.0f; tz += x_first ? 0.0f : dtz;
                    if ((unsigned)cx >= (unsigned)g.nx || (unsigned)cz >= (unsigned)g.nz) walking = false;
                }
            }
        }
#else
        while (__builtin_amdgcn_ballot_w64(walking) != 0) {
#ifdef RTIOW_PATH_STATS
            ++step_no;
#endif
            if (walking) {
                PATH_STAT(PS_GRID_STEP);
#ifdef RTIOW_PATH_STATS
                path_stat(step_no == 1 ? PS_STEP_1 : step_no == 2 ? PS_STEP_2 : step_no == 3 ? PS_STEP_3 : step_no == 4 ? PS_STEP_4 : step_no <= 8 ? PS_STEP_5_8 : PS_STEP_9_UP);
#endif
                const uint2 rec = cells[cz * g.nx + cx];
                if (rec.x != 0xffffffffu) cell_tests<T>(aos, rec.x, rec.y, pad2, O, D, a, closest, hit, fdc);
                // the parameter at which the ray leaves this cell, per axis
                const float bx = (float)(cx + (sx > 0 ? 1 : 0)) * g.cell, bz = (float)(cz + (sz > 0 ? 1 : 0)) * g.cell;
                const float tx = step_x ? (bx - ox) * inv_dx : __builtin_huge_valf();
                const float tz = step_z ? (bz - oz) * inv_dz : __builtin_huge_valf();
                const float tnext = __builtin_fminf(tx, tz);
                const float tend = __builtin_fminf(t1, (float)closest);           // (float) rounds to nearest: covered by eps
                if (tnext >= tend) walking = false;                               // leaves the slab / the grid, or a nearer hit is known
                else {
                    if (tx <= tz) cx += sx; else cz += sz;
                    if ((unsigned)cx >= (unsigned)g.nx || (unsigned)cz >= (unsigned)g.nz) walking = false;
                }
            }
        }
#endif
        REGION_END(walk, RG_GRID_WALK);
    };
    if (fd.on) direct_list_and_walk(std::true_type{});
    else direct_list_and_walk(std::false_type{});
}

template <class T, int SRC>
__device__ __forceinline__ void hit_world(const RenderParams<T>& p, const T* lds_geom, V3<T> O, V3<T> D, T a, T& closest, int& hit) {
    hit_world_direct<T, SRC>(p, lds_geom, O, D, a, closest, hit);
}
template <>
__device__ __forceinline__ void hit_world<double, RTIOW_SCENE_LDS>(const RenderParams<double>& p, const double* lds_geom, V3<double> O, V3<double> D,
                                                                   double a, double& closest, int& hit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (p.use_grid) {
        hit_world_grid<double>(p, lds_geom, reinterpret_cast<const float*>(smem_raw + p.screen_offset), O, D, a, closest, hit);
    } else if (p.use_screen) {
        hit_world_screened<double>(p, lds_geom, reinterpret_cast<const float*>(smem_raw + p.screen_offset), O, D, a, closest, hit);
    } else {
        hit_world_direct<double, RTIOW_SCENE_LDS>(p, lds_geom, O, D, a, closest, hit);
    }
}
template <>
__device__ __forceinline__ void hit_world<float, RTIOW_SCENE_LDS>(const RenderParams<float>& p, const float* lds_geom, V3<float> O, V3<float> D,
                                                                  float a, float& closest, int& hit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (p.use_grid) {
        hit_world_grid<float>(p, lds_geom, reinterpret_cast<const float*>(smem_raw + p.screen_offset), O, D, a, closest, hit);
    } else if (p.use_screen) {
        hit_world_screened<float>(p, lds_geom, reinterpret_cast<const float*>(smem_raw + p.screen_offset), O, D, a, closest, hit);
    } else {
        hit_world_direct<float, RTIOW_SCENE_LDS>(p, lds_geom, O, D, a, closest, hit);
    }
}

}  // namespace
