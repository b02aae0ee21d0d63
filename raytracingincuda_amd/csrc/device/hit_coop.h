// hit_coop.h -- cooperative hit_world of the drain tail: one ray per wave (DPP) and n rays per wave (LDS slots)
// Part of the single gfx950 translation unit rtiow_hip.hip (included there, in this order; internal linkage).
#pragma once
#include "hit_loop.h"

namespace {

// ---- the last stage of the drain: ONE ray left in a full wave.  The ray is broadcast with
// v_readlane (no LDS round trip), every lane takes one 4-sphere trip, and the 64 partial hits are
// reduced as one 64-bit key {t bits, index} -- t > 0 or +inf, so the IEEE bits order like the
// values and the key minimum is the lexicographic (t, index) minimum of the exact loop -- with DPP
// row operations (register-to-register) plus four readlanes, instead of 14 ds_bpermute round trips.
template <int CTRL> __device__ __forceinline__ unsigned dpp_mov(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ __forceinline__ void key_min_step(unsigned& hi, unsigned& lo) {
    const unsigned ohi = dpp_mov<CTRL>(hi), olo = dpp_mov<CTRL>(lo);
    const bool take = (ohi < hi) || (ohi == hi && olo < lo);
    hi = take ? ohi : hi;
    lo = take ? olo : lo;
}
__device__ __forceinline__ unsigned long long wave_min_key(unsigned hi, unsigned lo) {
    key_min_step<0xB1>(hi, lo);      // quad_perm [1,0,3,2]  : lane ^ 1
    key_min_step<0x4E>(hi, lo);      // quad_perm [2,3,0,1]  : lane ^ 2
    key_min_step<0x141>(hi, lo);     // row_half_mirror      : across the quads of a half row
    key_min_step<0x140>(hi, lo);     // row_mirror           : across the half rows -> every lane of a 16-lane row holds the row minimum
    unsigned long long best = ~0ull;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned long long k = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, row * 16) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)lo, row * 16);
        best = k < best ? k : best;
    }
    return best;
}

template <int SRC>
__device__ __forceinline__ void hit_world_solo(const RenderParams<float>& p, const float* lds_geom, int owner, bool is_owner,
                                               V3<float> O, V3<float> D, float a, float& closest, int& hit) {
    auto bcast = [owner](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), owner)); };
    const float ox = bcast(O.x), oy = bcast(O.y), oz = bcast(O.z);
    const float dx = bcast(D.x), dy = bcast(D.y), dz = bcast(D.z);
    const float ra = bcast(a);
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const float* gm = (SRC == RTIOW_SCENE_LDS) ? lds_geom : p.geom_a;
    const LoopRay<float> r = make_loop_ray(ox, oy, oz, dx, dy, dz, ra);
    float best = __builtin_huge_valf();
    int best_idx = -1;
#ifndef RTIOW_SOLO_ONE_FINISH
#define RTIOW_SOLO_ONE_FINISH 1
#endif
    for (int s = lane * 4; s < p.n_padded; s += 256) {
        if (!RTIOW_SOLO_ONE_FINISH) { sphere_trip<float>(gm, s, r, best, best_idx); continue; }
        // One finishing instance per round instead of the trip's four guarded ones: the ray's handful of candidates sit in different
        // lanes AND at different positions of their lanes' trips (the ground is sphere 0 = lane 0, position 0; a small sphere anywhere),
        // so the four sites ran one after the other, each for one lane -- on the critical path of a lone ray.  Here every lane takes its
        // own lowest pending candidate per round (index order within the lane, which is what the strict `root < closest` needs), and a
        // round is ONE pre-test + IEEE block for all of them: rounds = the most candidates any lane holds, almost always one.
        const Trip<float> t = trip_discriminants(gm, s, r);
        unsigned pend = (t.d0 >= 0.0f ? 1u : 0u) | (t.d1 >= 0.0f ? 2u : 0u) | (t.d2 >= 0.0f ? 4u : 0u) | (t.d3 >= 0.0f ? 8u : 0u);
        while (__builtin_amdgcn_ballot_w64(pend != 0) != 0) {
            if (pend != 0) {
                const int k = __builtin_ctz(pend);
                pend &= pend - 1;
                const float hk = k == 0 ? t.h0 : (k == 1 ? t.h1 : (k == 2 ? t.h2 : t.h3));
                const float dk = k == 0 ? t.d0 : (k == 1 ? t.d1 : (k == 2 ? t.d2 : t.d3));
                finish_sphere_test<float>(s + k, hk, dk, r.a, best, best_idx);
            }
        }
    }
    // Few lanes hold a hit at all (the ray meets a handful of spheres): walk those lanes with readlanes -- a short
    // scalar loop -- instead of the 64-lane DPP minimum, which is ~60 dependent instructions of pure latency here.
    const unsigned long long holders = __builtin_amdgcn_ballot_w64(best_idx >= 0);
    unsigned long long k = 0x7f800000ffffffffull;   // {+inf, -1}: no hit
    if (__builtin_popcountll(holders) <= 6) {
        unsigned long long m = holders;
        while (m != 0) {
            const int l = (int)__builtin_ctzll(m);
            m &= m - 1;
            const unsigned long long kl = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(best), l) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane(best_idx, l);
            k = kl < k ? kl : k;
        }
    } else k = wave_min_key(__float_as_uint(best), (unsigned)best_idx);
    if (is_owner) { closest = __uint_as_float((unsigned)(k >> 32)); hit = (int)(unsigned)k; }
}

template <int SRC>
__device__ __forceinline__ void coop_solo(const RenderParams<float>& p, const float* g, int owner, bool is_owner, V3<float> O, V3<float> D, float a, float& closest, int& hit) {
    hit_world_solo<SRC>(p, g, owner, is_owner, O, D, a, closest, hit);
}
template <int SRC>
__device__ __forceinline__ void coop_solo(const RenderParams<double>&, const double*, int, bool, V3<double>, V3<double>, double, double&, int&) {}

// ---- cooperative hit_world for the drain tail of the persistent kernel.
// When the work pool is empty and n <= 32 lanes of a wave still carry a path, the wave's
// idle lanes help: the n rays are published in LDS, each ray is served by a group of
// g = 2^floor(log2(lanes/n)) lanes that split the 4-sphere trips of hit_world_direct between
// them, and the partial nearest hits are reduced with xor-shuffles.  The nearest hit of
// the reference loop is the lexicographic minimum of (t, index) over the spheres -- a
// sphere's accepted root does not depend on closest_so_far except through `root < closest`
// (hittable.h:53-57) -- so any partition + min-reduction returns exactly what the
// sequential loop returns.  This cuts the latency of one segment from N sphere tests to
// N/g, which is what bounds the kernel once only the long glass paths are left.
template <class T> struct CoopSlot { T ox, oy, oz, a, dx, dy, dz, pad; };
// Slots a wave needs: the drain shares loops only while at most half of the wave's lanes carry a ray (persistent_body:
// 2 * popcount(alive) <= wave_lanes), so 32.  (64 until round 3: the 4 KB per workgroup were what kept the
// 487-sphere scene at four workgroups per CU -- 35.1 KB of LDS each -- instead of five.)
constexpr int COOP_SLOTS = 32;

// the value lane l holds, as a wave-uniform scalar
__device__ __forceinline__ float lane_value(float v, int l) { return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l)); }
__device__ __forceinline__ double lane_value(double v, int l) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// log2 of the lanes each of n rays gets when a wave of `lanes` lanes splits their sphere loops: floor(log2(lanes)) -
// ceil(log2(n)) -- exact for the 64-lane waves of the dynamic schedules, never too large otherwise (two count-
// leading-zeros instead of a loop: this runs every trip of the drain, where a lone ray's trip is all latency).
__device__ __forceinline__ int lanes_per_ray_log2(int n, int lanes) {
    const int up = n > 1 ? 32 - __builtin_clz((unsigned)(n - 1)) : 0;
    const int lg = (31 - __builtin_clz((unsigned)lanes)) - up;
    return lg > 0 ? lg : 0;
}

template <class T, int SRC>
__device__ __forceinline__ void hit_world_coop(const RenderParams<T>& p, const T* lds_geom, CoopSlot<T>* slots,
                                               bool alive, unsigned long long alive_mask, int n_alive, int wave_lanes,
                                               V3<T> O, V3<T> D, T a, T& closest, int& hit) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(alive_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)alive_mask, 0u));
    const int lg = lanes_per_ray_log2(n_alive, wave_lanes);   // g = 2^lg lanes per ray, n_alive * g <= lanes of this wave
    const int g = 1 << lg;
    if (alive) slots[rank] = {O.x, O.y, O.z, a, D.x, D.y, D.z, (T)0};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int group = lane >> lg, sub = lane & (g - 1);
    T best = __builtin_huge_val();
    int best_idx = -1;
    if (group < n_alive) {
        const CoopSlot<T> cs = slots[group];
        const LoopRay<T> r = make_loop_ray(cs.ox, cs.oy, cs.oz, cs.dx, cs.dy, cs.dz, cs.a);
        const T* gm = (SRC == RTIOW_SCENE_LDS) ? lds_geom : p.geom_a;
        for (int s = sub * 4; s < p.n_padded; s += g * 4) sphere_trip<T>(gm, s, r, best, best_idx);
    }
    if (n_alive <= 4) {
        // Few rays, wide groups: the xor-shuffle reduction below is log2(g) dependent LDS round trips (five for two
        // rays; fp64 has no single-ray path, so six for one).  A ray meets a handful of spheres, so few lanes of its
        // group hold a hit: walk those lanes with readlanes, ray by ray, and hand the result to the ray's owner --
        // all scalar, no LDS.
        const unsigned long long holders = __builtin_amdgcn_ballot_w64(best_idx >= 0);
        const unsigned long long group_lanes = g >= 64 ? ~0ull : ((1ull << g) - 1);
        for (int j = 0; j < n_alive; ++j) {
            unsigned long long m = holders & (group_lanes << (j << lg));
            T bt = __builtin_huge_val();
            int bi = -1;
            while (m != 0) {
                const int l = (int)__builtin_ctzll(m);
                m &= m - 1;
                const T tl = lane_value(best, l);
                const int il = __builtin_amdgcn_readlane(best_idx, l);
                const bool take = (tl < bt) || (tl == bt && (unsigned)il < (unsigned)bi);
                bt = take ? tl : bt;
                bi = take ? il : bi;
            }
            if (alive && rank == j) { closest = bt; hit = bi; }
        }
        return;
    }
    // lexicographic (t, index) minimum over the g lanes of the group
    for (int off = 1; off < g; off <<= 1) {
        const T ot = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(best_idx, off, 64);
        const bool take = (ot < best) || (ot == best && (unsigned)oi < (unsigned)best_idx);
        best = take ? ot : best;
        best_idx = take ? oi : best_idx;
    }
    // owner of slot k reads lane k*g
    const T rt = __shfl(best, rank << lg, 64);
    const int ri = __shfl(best_idx, rank << lg, 64);
    if (alive) { closest = rt; hit = ri; }
}

}  // namespace
