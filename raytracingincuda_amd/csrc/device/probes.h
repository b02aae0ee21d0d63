// probes.h -- instrumentation that is compiled out by default: double-execution probes, execution profile
// Part of the single gfx950 translation unit rtiow_hip.hip (included there, in this order; internal linkage).
#pragma once
#include "params.h"

namespace {

// ---- double-execution probes (scripts/cost_probe.sh; compiled out by default).  With -DRTIOW_PROBE_<X> the
// component X runs a SECOND time on copies of its inputs and the results are thrown away behind an opaque
// asm, so the image is unchanged and the growth of SQ_INSTS_VALU is exactly what X costs.
#define RT_KEEP1(v) asm volatile("" :: "v"(v))
template <class T> __device__ __forceinline__ void rt_opaque(V3<T>& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z)); }
__device__ __forceinline__ void rt_opaque(Rng& r) { asm volatile("" : "+v"(r.v0), "+v"(r.v1), "+v"(r.v2), "+v"(r.v3), "+v"(r.v4), "+v"(r.d)); }
// The probe sites, one macro each (the functions they call are defined later in the translation unit; a macro
// expands where it is used).  Every macro is empty in the product build.
#ifdef RTIOW_PROBE_DIRECT      // hit_world_grid: the direct list a second time
#define RT_PROBE_DIRECT(T, smem, g, O, D, a, fd) do { \
        V3<T> o2 = (O), d2 = (D); rt_opaque(o2); rt_opaque(d2); T c2 = __builtin_huge_val(); int h2 = -1; \
        const T* dg_ = reinterpret_cast<const T*>((smem) + (g).direct_offset); \
        const int* ids_ = reinterpret_cast<const int*>((smem) + (g).direct_ids_offset); \
        const LoopRay<T> r_ = make_loop_ray(o2.x, o2.y, o2.z, d2.x, d2.y, d2.z, (a)); \
        for (int s_ = 0; s_ < (g).n_direct_padded; s_ += 4) direct_trip<T>(dg_, ids_, s_, r_, c2, h2, (fd)); \
        RT_KEEP1(c2); RT_KEEP1(h2); } while (0)
#else
#define RT_PROBE_DIRECT(T, smem, g, O, D, a, fd) ((void)0)
#endif
#ifdef RTIOW_PROBE_RUV         // the rejection rounds of random_unit_vector a second time on a copy of the generator
#define RT_PROBE_RUV(T, rs, rounds) do { Rng c_ = (rs); rt_opaque(c_); T x_, y_, z_, l_; const bool f_ = random_unit_vector_rounds<T>(c_, (rounds), x_, y_, z_, l_); \
        RT_KEEP1(x_); RT_KEEP1(y_); RT_KEEP1(z_); RT_KEEP1(l_); RT_KEEP1((int)f_); RT_KEEP1(c_.v4); } while (0)
#else
#define RT_PROBE_RUV(T, rs, rounds) ((void)0)
#endif
#ifdef RTIOW_PROBE_GEN         // persistent_body: the primary ray a second time
#define RT_PROBE_GEN(T, p, i, j, rs) do { Rng c_ = (rs); rt_opaque(c_); V3<T> o2, d2; T u2; gen_primary((p), (i), (j), c_, o2, d2, u2); \
        RT_KEEP1(o2.x); RT_KEEP1(o2.y); RT_KEEP1(o2.z); RT_KEEP1(d2.x); RT_KEEP1(d2.y); RT_KEEP1(d2.z); RT_KEEP1(u2); RT_KEEP1(c_.v4); } while (0)
#else
#define RT_PROBE_GEN(T, p, i, j, rs) ((void)0)
#endif
#ifdef RTIOW_PROBE_HIT         // persistent_body: hit_world a second time
#define RT_PROBE_HIT(T, SRC, p, lds_geom, O, D) do { V3<T> o2 = (O), d2 = (D); rt_opaque(o2); rt_opaque(d2); T c2 = __builtin_huge_val(); int h2 = -1; \
        hit_world<T, SRC>((p), (lds_geom), o2, d2, dot3(d2, d2), c2, h2); RT_KEEP1(c2); RT_KEEP1(h2); } while (0)
#else
#define RT_PROBE_HIT(T, SRC, p, lds_geom, O, D) ((void)0)
#endif
#ifdef RTIOW_PROBE_SHADE       // persistent_body: the shade step (with its random_unit_vector) a second time on a copy of the path state
#define RT_PROBE_SHADE(T, p, lds_shade, st, closest, hit) do { PathState<T> s2 = (st); rt_opaque(s2.O); rt_opaque(s2.D); rt_opaque(s2.rs); V3<T> c2; bool r2_; \
        const bool t2 = shade_step<T, false>((p), (lds_shade), s2, (closest), (hit), c2, 0, r2_); RT_KEEP1(c2.x); RT_KEEP1(c2.y); RT_KEEP1(c2.z); RT_KEEP1(s2.O.x); \
        RT_KEEP1(s2.D.x); RT_KEEP1(s2.D.y); RT_KEEP1(s2.D.z); RT_KEEP1(s2.rs.v4); RT_KEEP1(s2.atten.x); RT_KEEP1((int)t2); } while (0)
#else
#define RT_PROBE_SHADE(T, p, lds_shade, st, closest, hit) ((void)0)
#endif

// ---- optional execution profile (build with -DRTIOW_PATH_STATS, `python -m raytracingincuda_amd.build
// --stats`): per region, how many times a WAVE executed it and with how many active lanes.  The
// kernel is bound by the vector instructions it issues, and a divergent region costs its full
// instruction count whenever one lane needs it, so (wave executions x static instruction count)
// is the time budget (scripts/path_stats_probe.py).  Compiled out by default.
#ifdef RTIOW_PATH_STATS
enum { PS_ITERATION = 0, PS_RUV_CALL, PS_RUV_ROUND, PS_DISK_ROUND, PS_GEN_PRIMARY, PS_SHADE_HIT, PS_SKY, PS_DIELECTRIC, PS_METAL,
       PS_EXACT_BLOCK, PS_FINISH_CALL, PS_IEEE_BLOCK, PS_SECOND_DIV, PS_SCHLICK_DRAW, PS_REFILL, PS_FINISH_PIXEL, PS_GRID_STEP,
       PS_STEP_1, PS_STEP_2, PS_STEP_3, PS_STEP_4, PS_STEP_5_8, PS_STEP_9_UP, PS_WALK, PS_CELL_PAIR2, PS_COUNT };
__device__ unsigned long long g_path_stats[2 * PS_COUNT];
__device__ __forceinline__ void path_stat(int region) {
    const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (lane == __builtin_ctzll(act)) {
        atomicAdd(&g_path_stats[2 * region], 1ull);
        atomicAdd(&g_path_stats[2 * region + 1], (unsigned long long)__builtin_popcountll(act));
    }
}
#define PATH_STAT(r) path_stat(r)
// Region clocks of the same build: shader cycles a WAVE spends between two points, summed over all
// waves (s_memtime; the reads themselves cost ~10 % and other waves' issue slots are included, so
// only the proportions mean something).  scripts/path_stats_probe.py prints them.
enum { RG_REFILL = 0, RG_GEN_PRIMARY, RG_HIT_WORLD, RG_HIT_COOP, RG_SHADE, RG_ACCUMULATE, RG_GRID_SETUP, RG_GRID_DIRECT, RG_GRID_WALK, RG_GRID_FALLBACK,
       RG_RUV_ROUNDS, RG_LOOP_TOTAL, RG_COUNT };
__device__ unsigned long long g_region_cycles[RG_COUNT];
__device__ __forceinline__ void region_add(int region, unsigned long long t0) {
    const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0;
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) atomicAdd(&g_region_cycles[region], dt);
}
#define REGION_BEGIN(name) const unsigned long long rg_##name = __builtin_amdgcn_s_memtime()
#define REGION_END(name, region) region_add(region, rg_##name)
#else
#define PATH_STAT(r) ((void)0)
#define REGION_BEGIN(name) ((void)0)
#define REGION_END(name, region) ((void)0)
#endif

}  // namespace
