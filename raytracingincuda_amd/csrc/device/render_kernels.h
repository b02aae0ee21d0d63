// render_kernels.h -- the render kernels: static (reference geometry), persistent / prepass / solo, debug kernels
// Part of the single gfx950 translation unit rtiow_hip.hip (included there, in this order; internal linkage).
#pragma once
#include "shade.h"
#include "hit_coop.h"
#include "pixel_io.h"

namespace {

// ---- SCHED_STATIC: the reference's launch geometry, one lane = one pixel of a T x T block
// (camera.h:131-134), with the flattened sample/bounce loop.
template <class T, int SRC, bool COUNT>
__global__ void __launch_bounds__(1024)
render_kernel(const RenderParams<T> p) {
    const T* lds_geom = stage_scene<T, SRC>(p);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const T* lds_shade = reinterpret_cast<const T*>(smem_raw + p.shade_offset);
    const ColdParams<T>& c = p.cold;
    const int tid = threadIdx.x;
    int tx, ty;
    if (c.wave_tiles) {
        const int wave = tid >> 6, lane = tid & 63;
        const int tiles_x = c.bx >> 3;
        tx = (wave % tiles_x) * 8 + (lane & 7);
        ty = (wave / tiles_x) * 8 + (lane >> 3);
    } else {
        tx = tid % c.bx;                        // CUDA's threadIdx.x
        ty = tid / c.bx;                        // CUDA's threadIdx.y
    }
    const int i = blockIdx.x * c.bx + tx;
    const int jl = blockIdx.y * c.by + ty;      // local row
    if (i >= c.W || jl >= c.local_rows) return; // camera.h:133
    const int j = global_row(jl, c.strip_rows, c.nranks, c.rank);
    const size_t lp = (size_t)jl * c.W + i;

    const size_t npix = (size_t)c.W * c.local_rows;
    PathState<T> st;
    st.rs.v0 = c.rng_in[0 * npix + lp]; st.rs.v1 = c.rng_in[1 * npix + lp]; st.rs.v2 = c.rng_in[2 * npix + lp];   // camera.h:136
    st.rs.v3 = c.rng_in[3 * npix + lp]; st.rs.v4 = c.rng_in[4 * npix + lp]; st.rs.d = c.rng_in[5 * npix + lp];
    st.acc = {0, 0, 0};
    st.sample = c.s_begin; st.depth = 0;          // this launch renders samples [s_begin, s_end)
    unsigned int nseg = 0, cost = 0;
    const int S = p.s_end;
    bool fresh = true;                            // the lane needs a primary ray (camera.h:141-155)

    while (st.sample < S) {
        PATH_STAT(PS_ITERATION);
        if (fresh) { gen_primary(p, i, j, st.rs, st.O, st.D, st.sky_uy); st.atten = {1, 1, 1}; fresh = false; }
        V3<T> col;
        if (st.depth < p.B) { ++cost; if (COUNT) ++nseg; }
        if (segment_step<T, SRC>(p, lds_geom, lds_shade, st, col)) {
            st.acc = {st.acc.x + col.x, st.acc.y + col.y, st.acc.z + col.z};       // camera.h:160
            ++st.sample;
            st.depth = 0;
            fresh = true;
        }
    }
    if (COUNT) { atomicAdd(c.seg_counter, (unsigned long long)nseg); atomicMax(c.seg_counter + 2, (unsigned long long)cost); }
    finish_pixel<T>(c, lp, st, cost);
}

// ---- SCHED_PERSISTENT: lanes are not bound to pixels.  Each wave keeps a pool of 64 pixel
// slots (one 8x8 tile) taken from a global counter; a lane that finishes its pixel takes the
// next slot at once (ballot + mbcnt hand-out, no memory traffic), so no lane waits for the
// longest path of a tile-mate and the grid is balanced across CUs by construction.  Slots run
// tile-major from the BOTTOM of the image up (ground and spheres first, cheap sky last) to
// keep the drain tail short.  Per-pixel work and RNG streams are unchanged => same image.
constexpr int POOL = 64;
// Bounded rejection loop of random_unit_vector (shade_step<T, true>): rounds a wave runs per trip before the lanes still without a candidate
// resume in the next one.  Unbounded in the drain, where a lone chain's latency counts and no other lane waits for the rounds.  fp64 -- whose
// round is six XORWOW steps -- gains 5 % everywhere (round 3) and always bounds its loop.  fp32 gains where throughput binds and loses where one
// chain's latency does: round 3 measured -5.6 % vector instructions and no time on the 1080p frame, -2...4 % on small frames and shards; with
// round 4's shorter trips 1080p gains 1.8 %, 2560 x 1440 4.3 %, 3840 x 2160 2.2 %, 1280 x 720 loses 1-3 % (profiles/r04/ab_bounded_rounds_f32.jsonl,
// ruv_rounds_sweep.jsonl).  The bound has to be a compile-time constant (the three rounds unroll; as a launch parameter the loop and its two extra
// registers cost what the bound saves), so the fp32 main launch and prepass exist in both forms (template argument BOUND_F32) and launch_render
// takes the bounded one when the launch has at least four 64-pixel pools per resident wave.
#ifndef RTIOW_RUV_ROUNDS_PER_ITERATION
#define RTIOW_RUV_ROUNDS_PER_ITERATION 3
#endif
// Longest share of the brute-force sphere loop (trips of four spheres) for which the drain still splits it
// among idle lanes instead of walking the grid (persistent_body).
#ifndef RTIOW_COOP_MAX_TRIPS
#define RTIOW_COOP_MAX_TRIPS 6
#endif

// Slot -> pixel of the tile-ordered hand-out (no cost order): 8 x 8 tiles from the bottom of the image up, 64 slots
// per tile.  False for the padded slots of ragged tiles.
template <class COLD>
__device__ __forceinline__ bool tile_slot_pixel(const COLD& c, int slot, int& i, int& jl) {
    const int tiles_x = (c.W + 7) >> 3, tiles_y = (c.local_rows + 7) >> 3;
    const int t = slot >> 6, within = slot & 63;
    const int ty = tiles_y - 1 - t / tiles_x, tx = t % tiles_x;
    i = tx * 8 + (within & 7);
    jl = ty * 8 + (within >> 3);
    return i < c.W && jl < c.local_rows;
}

// {s_memtime, s_memrealtime} of wave 0 of workgroup 0 into ColdParams::clock_stamps[slot, slot + 1] (slot 0: the wave starts, 2: it ends).
template <class T>
__device__ __forceinline__ void clock_stamp(const RenderParams<T>& p, int slot) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long* s = cold_of(p).clock_stamps;
        if (s) { s[slot] = __builtin_amdgcn_s_memtime(); s[slot + 1] = __builtin_amdgcn_s_memrealtime(); }
    }
}

template <class T, int SRC, bool COUNT, bool SOLO = false, bool BOUND_F32 = false>
__device__ __forceinline__ void persistent_body(const RenderParams<T>& p) {
    const T* lds_geom = stage_scene<T, SRC>(p);
    // per-wave scratch for hit_world_coop, behind the staged tables
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const T* lds_shade = reinterpret_cast<const T*>(smem_raw + p.shade_offset);
    CoopSlot<T>* coop_slots = reinterpret_cast<CoopSlot<T>*>(smem_raw + p.coop_offset) + (threadIdx.x >> 6) * COOP_SLOTS;
    const int S = p.s_end;                       // this launch renders samples [cold.s_begin, p.s_end)
    clock_stamp(p, 0);                           // ColdParams::clock_stamps: one lane of the launch, here and behind the loop

    PathState<T> st;
    st.acc = {0, 0, 0};
    st.sample = 0; st.depth = 0;
    unsigned int cost = 0;                       // segments of the lane's current pixel in this launch
#ifndef RTIOW_RUV_BOUNDED_F64
#define RTIOW_RUV_BOUNDED_F64 1
#endif
    constexpr bool RETRY = sizeof(T) == 8 ? RTIOW_RUV_BOUNDED_F64 != 0 : BOUND_F32;   // see RTIOW_RUV_ROUNDS_PER_ITERATION
    bool retry = false;                          // RETRY: the lane's rejection loop goes on in this iteration (closest, hit kept)
    T closest = __builtin_huge_val();
    int hit = -1;
    bool alive = false, fresh = false;
    int i = 0, j = 0;
    size_t lp = 0;
    unsigned int nseg = 0;
    int pool_next = 0, pool_end = 0;             // wave-uniform
    bool exhausted = false;                      // wave-uniform
    // p.first_pools: wave w takes pool w first and the counter starts behind them.  Workgroups are
    // dispatched in blockIdx order and the SIMD arbiter favours older waves, so this puts the
    // heaviest block of the cost-sorted order on the waves that will run fastest.
    const int take = p.lane_cap;                 // slots per refill: 64, fewer in an underfilled launch
    int first_pool = -1, first_take = take;
    bool solo = false;                           // wave-uniform (SOLO kernels): this wave holds only its share of the heaviest pixels
    bool takes_pixels = (int)(threadIdx.x & 63u) < p.lane_cap;
    if (SOLO) {
        // ColdParams::solo_*: wave 0 of the first solo_waves workgroups takes solo_lanes of the top-ranked pixels and
        // nothing else until they are done; the other waves number their first pools without it.
        const auto& c = cold_of(p);
        const int wpb = (int)((blockDim.x + 63) >> 6), w = (int)(threadIdx.x >> 6), b = (int)blockIdx.x;
        const int ns = c.solo_waves, sl = c.solo_lanes;
        if (w == 0 && b < ns) {
            first_pool = b * sl; first_take = sl; solo = true;
            takes_pixels = (int)(threadIdx.x & 63u) < sl;
        } else {
            first_pool = ns * sl + (b * wpb + w - (b < ns ? b + 1 : ns)) * take;
        }
    } else if (cold_of(p).first_pools) {
        first_pool = ((int)blockIdx.x * (int)((blockDim.x + 63) >> 6) + (int)(threadIdx.x >> 6)) * take;
    }
    unsigned long long t_start = 0, t_exh = 0;
    unsigned int it_normal = 0, it_coop = 0, n_pixels = 0;
    if (COUNT) t_start = __builtin_amdgcn_s_memrealtime();
    const int lanes_left = (int)blockDim.x - (int)(threadIdx.x & ~63u);
    const int wave_lanes = lanes_left < 64 ? lanes_left : 64;   // partial last wave of a T x T block

    auto refill = [&]() __attribute__((always_inline)) {
        if (!exhausted && !(SOLO && solo && first_pool < 0) && __builtin_amdgcn_ballot_w64(!alive && takes_pixels) != 0) {
            bool want = !alive && takes_pixels;
            PATH_STAT(PS_REFILL);
            const auto& c = cold_of(p);          // image / shard geometry and buffers: scalar loads here, not live in the path loop
            const int total_slots = c.total_slots;
            for (;;) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(want);
                if (m == 0) break;
                if (pool_next >= pool_end) {     // refill the wave's pool: one atomic per 64 pixels
                    int base = 0, this_take = take;
                    if (first_pool >= 0) {       // the first pool follows dispatch order (= wave age), see launch_render
                        base = first_pool; this_take = first_take;
                        first_pool = -1;
                    } else {
                        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) base = (int)atomicAdd(c.work_counter, (unsigned)take);
                        base = __builtin_amdgcn_readfirstlane(base);
                    }
                    if (base >= total_slots) { exhausted = true; if (COUNT) t_exh = __builtin_amdgcn_s_memrealtime(); break; }
                    pool_next = base; pool_end = base + this_take;
                }
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                const int avail = pool_end - pool_next;
                const int wanted = __builtin_popcountll(m);
                const bool take = want && rank < avail;
                const int slot = pool_next + rank;
                pool_next += wanted < avail ? wanted : avail;
                if (take) {
                    int jl;
                    bool valid;
                    if (c.order) {                               // cost-sorted hand-out (main launch of the sorted schedule)
                        const int px = c.order[slot];            // (row << 16 | column): no division by the image width here
                        valid = px >= 0;
                        jl = valid ? px >> 16 : 0;
                        i = valid ? px & 0xffff : 0;
                    } else valid = tile_slot_pixel(c, slot, i, jl);   // 8x8 tiles, bottom-up
                    if (valid) {
                        want = false;
                        j = c.nranks == 1 ? jl : global_row(jl, c.strip_rows, c.nranks, c.rank);
                        lp = (size_t)jl * c.W + i;
                        if (c.mid_in) unpark_state<T>(c.mid_in, lp, st);
                        else {
                            const size_t npix = (size_t)c.W * c.local_rows;
                            st.rs.v0 = c.rng_in[0 * npix + lp]; st.rs.v1 = c.rng_in[1 * npix + lp]; st.rs.v2 = c.rng_in[2 * npix + lp];
                            st.rs.v3 = c.rng_in[3 * npix + lp]; st.rs.v4 = c.rng_in[4 * npix + lp]; st.rs.d = c.rng_in[5 * npix + lp];
                            st.acc = {0, 0, 0};
                        }
                        st.sample = c.s_begin; st.depth = 0;
                        cost = 0;
#ifdef RTIOW_PIXEL_TIMES      // study builds only (scripts/pixel_finish_study.py): the fp64 counting kernel has no registers to spare
                        if (COUNT && c.pixel_times) c.pixel_times[4 * lp] = (unsigned)__builtin_amdgcn_s_memrealtime();   // taken (no register carried to the finish)
#endif
                        if (c.stage_by_slot) lp = (size_t)slot;      // where this pixel will be stored (ColdParams::stage_by_slot); the state is loaded
                        if (COUNT) ++n_pixels;
                        if (c.s_begin < S) { alive = true; fresh = true; }
                        else { finish_pixel<T>(c, lp, st, cost); want = true; }   // nothing to render in this launch
                    }
                }
            }
        }
    };

#ifndef RTIOW_MERGED_ROUNDS
#define RTIOW_MERGED_ROUNDS 1
#endif
    // ---- The rotated trip (fp64, full launches, until the work counter runs dry): hit_world -> shade_front (sky, hit record, the whole dielectric)
    // -> accumulate + refill -> jitter of the lanes that start a sample -> ONE rejection loop for the lens samples of those lanes and the unit
    // vectors of the lanes that scatter diffusely (sampling.h merged_rounds) -> primary_finish / shade_back.  The loop below keeps the reference's
    // order (primary ray first) and runs the two rejection loops one after the other: 2.8 + 2.8 wave-rounds per trip for 24 and 35 lanes, in fp64
    // 48 and 72 vector instructions each (two generator steps per number) -- 38 % of the trip.  Side by side they take max, not sum, and share four of
    // six generator steps: vector instructions -5.8 %, fp64 headline 15.1 -> 14.5 ms (profiles/r05/ab_merged_rounds_f64.jsonl).  Every pixel draws the
    // same numbers in the same order.  When the wave finds the counter exhausted it finishes the trip and goes on in the loop below (cooperative
    // drain) with the same lane state.  fp32 (in-place generator blocks, 38 / 27 instructions a round): built, -1.8 % instructions, +-0.5 % time, not used.
#ifndef RTIOW_MERGED_ROUNDS_SOLO_KERNEL
#define RTIOW_MERGED_ROUNDS_SOLO_KERNEL 1     // the solo kernel's ordinary waves take the rotated trip too (its solo waves need the cooperative hit_world of the loop below):
#endif                                        // fp64 1280 x 720 -2.7 %, half-frame shard -3.4 %, 960 x 540 -2.3 % (profiles/r05/ab_rotated_trip_in_solo_kernel_f64.jsonl)
    if (RTIOW_MERGED_ROUNDS && (!SOLO || (RTIOW_MERGED_ROUNDS_SOLO_KERNEL && !solo)) && sizeof(T) == 8 && p.lane_cap == 64 && (blockDim.x & 63u) == 0) {
        const bool defocus = !((T)cam_of(p).defocus_angle <= (T)0);
        // camera.h:160-171 for a lane whose path has ended (a macro: as a lambda it kept `alive` and `fresh` in scratch memory)
#define RT_END_SAMPLE() do { \
            ++st.sample; \
            st.depth = 0; \
            if (st.sample < S) fresh = true; \
            else { \
                PATH_STAT(PS_FINISH_PIXEL); \
                const auto& c_ = cold_of(p); \
                if (COUNT) atomicMax(c_.seg_counter + 2, (unsigned long long)cost); \
                finish_pixel<T>(c_, lp, st, cost); alive = false; \
            } } while (0)
        for (;;) {
            REGION_BEGIN(total);
            if (alive) PATH_STAT(PS_ITERATION);
            // ---- hit_world (camera.h:84-88) for every lane that holds a ray and is not in the middle of its rejection loop
            const bool has_ray = alive && !fresh;
            const bool need_hit = has_ray && !(RETRY && retry) && st.depth < p.B;
            if (!(RETRY && retry)) { closest = __builtin_huge_val(); hit = -1; }
            pin_ray(st.O, st.D);
            if (COUNT) ++it_normal;
            REGION_BEGIN(hw);
            if (need_hit) {
                const T a = dot3(st.D, st.D);                 // hittable.h:43, ray-invariant
                hit_world<T, SRC>(p, lds_geom, st.O, st.D, a, closest, hit);
            }
            REGION_END(hw, RG_HIT_WORLD);
            // ---- sky / hit record / dielectric; lambertian and metal stop in front of random_unit_vector
            REGION_BEGIN(shade);
            bool need_ruv = false;
            ShadeCarry<T> sc;
            sc.nrm = {0, 0, 0}; sc.fuzz = 0; sc.mtype = 0;       // (defined on every path: an undefined value would be live around the whole loop)
            if (has_ray) {
                V3<T> col = {0, 0, 0};
                bool terminated = true;                                              // camera.h:127 at the depth limit
                if (need_hit || (RETRY && retry)) {
                    if (need_hit) { ++cost; if (COUNT) ++nseg; }
                    const int r = shade_front<T>(p, lds_shade, st, closest, hit, col, sc);
                    terminated = r == SF_TERMINATED;
                    need_ruv = r == SF_NEED_RUV;
                }
                if (terminated) {
                    st.acc = {st.acc.x + col.x, st.acc.y + col.y, st.acc.z + col.z};   // camera.h:160
                    RT_END_SAMPLE();
                }
            }
            REGION_END(shade, RG_SHADE);
            REGION_BEGIN(refill);
            refill();
            REGION_END(refill, RG_REFILL);
            // ---- the sampling step
            REGION_BEGIN(gen);
            T jox = 0, joy = 0;
            const bool starts = alive && fresh;
            if (starts) { PATH_STAT(PS_GEN_PRIMARY); primary_jitter<T>(st.rs, jox, joy); }
            int looking = (starts && defocus) ? MR_DISK : (need_ruv ? MR_RUV : MR_CLOSED);
            if (need_ruv) PATH_STAT(PS_RUV_CALL);
            T cx = 0, cy = 0, cz = 0, cl = 0;
            merged_rounds(st.rs, RETRY ? RTIOW_RUV_ROUNDS_PER_ITERATION : 0x7fffffff, looking, cx, cy, cz, cl);
            if (looking == MR_DISK) { disk_candidate<T>(st.rs, cx, cy); looking = MR_CLOSED; }
            const bool r_open = looking == MR_RUV;
            retry = RETRY && r_open;
            // ---- what follows the rejection loops
            if (starts) {
                primary_finish<T>(p, i, j, jox, joy, defocus, cx, cy, st.O, st.D, st.sky_uy);
                st.atten = {1, 1, 1};
                fresh = false;
            }
            REGION_END(gen, RG_GEN_PRIMARY);
            REGION_BEGIN(acc);
            if (need_ruv && !r_open) {
                if (!shade_back<T>(p, lds_shade, st, sc, closest, hit, cx, cy, cz, cl)) RT_END_SAMPLE();   // the metal absorbed the ray: black (camera.h:117)
            }
            REGION_END(acc, RG_ACCUMULATE);
            REGION_END(total, RG_LOOP_TOTAL);
            if (exhausted) break;
        }
#undef RT_END_SAMPLE
    }
    for (;;) {
        REGION_BEGIN(total);
        REGION_BEGIN(refill);
        if (SOLO && solo && first_pool < 0 && __builtin_amdgcn_ballot_w64(alive) == 0) {   // the solo pixels are done: an ordinary wave from here on
            solo = false;
            takes_pixels = (int)(threadIdx.x & 63u) < p.lane_cap;
        }
        refill();
        REGION_END(refill, RG_REFILL);
        const unsigned long long alive_mask = __builtin_amdgcn_ballot_w64(alive);
        if (alive_mask == 0) break;
        if (alive) PATH_STAT(PS_ITERATION);
        // one site generates every primary ray: first sample of a new pixel or the next sample
        REGION_BEGIN(gen);
        if (alive && fresh) RT_PROBE_GEN(T, p, i, j, st.rs);
        if (alive && fresh) { gen_primary(p, i, j, st.rs, st.O, st.D, st.sky_uy); st.atten = {1, 1, 1}; fresh = false; }
        REGION_END(gen, RG_GEN_PRIMARY);
        pin_ray(st.O, st.D);                                // hit_loop.h: the ray's components as opaque registers, once per trip and in place
        bool terminated = false;
        V3<T> col = {0, 0, 0};
        // hit_world for every lane that still traces (camera.h:84-88), then ONE shade site
        const bool need_hit = alive && !(RETRY && retry) && st.depth < p.B;
        if (!(RETRY && retry)) { closest = __builtin_huge_val(); hit = -1; }
        bool share_loops = (exhausted || (SOLO && solo) || 2 * p.lane_cap <= wave_lanes) && 2 * __builtin_popcountll(alive_mask) <= wave_lanes;
        const unsigned long long hit_mask = __builtin_amdgcn_ballot_w64(need_hit);
        if (share_loops && p.use_grid && hit_mask != 0) {
            // With a grid, sharing the brute-force loop only pays while a ray's share of it is short: g lanes per
            // ray leave it n_trips / g trips of ~38 instructions, the grid path costs ~300 whatever the lane count.
            const int n_need = __builtin_popcountll(hit_mask);
            const int lg = lanes_per_ray_log2(n_need, wave_lanes);
            share_loops = ((p.n_padded >> 2) + (1 << lg) - 1) >> lg <= RTIOW_COOP_MAX_TRIPS;
        }
        if (share_loops) {
            // drain tail: idle lanes share the survivors' sphere loops (hit_world_coop)
            if (COUNT) ++it_coop;
            REGION_BEGIN(coop);
            if (hit_mask != 0) {
                const T a = dot3(st.D, st.D);
                if (sizeof(T) == 4 && wave_lanes == 64 && (hit_mask & (hit_mask - 1)) == 0)
                    coop_solo<SRC>(p, lds_geom, (int)__builtin_ctzll(hit_mask), need_hit, st.O, st.D, a, closest, hit);
                else
                    hit_world_coop<T, SRC>(p, lds_geom, coop_slots, need_hit, hit_mask, __builtin_popcountll(hit_mask), wave_lanes, st.O, st.D, a, closest, hit);
            }
            REGION_END(coop, RG_HIT_COOP);
        } else {
            if (COUNT) ++it_normal;
            REGION_BEGIN(hw);
            if (need_hit) {
                const T a = dot3(st.D, st.D);                 // hittable.h:43, ray-invariant
                RT_PROBE_HIT(T, SRC, p, lds_geom, st.O, st.D);
                hit_world<T, SRC>(p, lds_geom, st.O, st.D, a, closest, hit);
            }
            REGION_END(hw, RG_HIT_WORLD);
        }
        REGION_BEGIN(shade);
        if (alive && need_hit) RT_PROBE_SHADE(T, p, lds_shade, st, closest, hit);
        if (alive) {
            if (need_hit) { ++cost; if (COUNT) ++nseg; }
            if (need_hit || (RETRY && retry))
                terminated = shade_step<T, RETRY>(p, lds_shade, st, closest, hit, col, share_loops ? 0x7fffffff : RTIOW_RUV_ROUNDS_PER_ITERATION, retry);
            else terminated = true;                                                  // camera.h:127 at the depth limit
        }
        REGION_END(shade, RG_SHADE);
        REGION_BEGIN(acc);
        if (alive && terminated) {
            st.acc = {st.acc.x + col.x, st.acc.y + col.y, st.acc.z + col.z};       // camera.h:160
            ++st.sample;
            st.depth = 0;
            if (st.sample < S) fresh = true;
            else {
                PATH_STAT(PS_FINISH_PIXEL);
                const auto& c = cold_of(p);
                if (COUNT) atomicMax(c.seg_counter + 2, (unsigned long long)cost);   // a pixel's samples are ONE sequential chain: the frame cannot be shorter than the longest
#ifdef RTIOW_PIXEL_TIMES
                if (COUNT && c.pixel_times) {
                    size_t px = lp;                                  // the local pixel: lp itself, or -- staged stores -- what the order holds for the slot
                    if (c.stage_by_slot) { const int e = c.order[lp]; px = (size_t)(e >> 16) * c.W + (size_t)(e & 0xffff); }
                    uint32_t* o = c.pixel_times + 4 * px;
                    o[1] = (unsigned)__builtin_amdgcn_s_memrealtime(); o[2] = cost;
                    o[3] = blockIdx.x * ((blockDim.x + 63) >> 6) + (threadIdx.x >> 6);
                }
#endif
                finish_pixel<T>(c, lp, st, cost); alive = false;
            }
        }
        REGION_END(acc, RG_ACCUMULATE);
        REGION_END(total, RG_LOOP_TOTAL);
    }
    clock_stamp(p, 2);
    if (COUNT) {
        const auto& c = cold_of(p);
        atomicAdd(c.seg_counter, (unsigned long long)nseg);
        if (c.timeline) {
            unsigned int px = n_pixels;
            for (int off = 32; off > 0; off >>= 1) px += __shfl_xor(px, off, 64);
            if ((threadIdx.x & 63) == 0) {
                unsigned long long* o = c.timeline + 8ull * ((unsigned long long)blockIdx.x * ((blockDim.x + 63) >> 6) + (threadIdx.x >> 6));
                o[0] = t_start; o[1] = t_exh; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = it_normal; o[4] = it_coop; o[5] = px; o[6] = 0; o[7] = 0;
            }
        }
    }
}

// The same body under two kernel names, so that profiles tell the launches of RTIOW_SCHED_SORTED
// apart: the prepass (samples [0, SA) in tile order, ~1.4 ms of the headline frame) and the main
// launch (everything else; also the only launch of RTIOW_SCHED_PERSISTENT).
#ifdef RTIOW_MAIN_WAVES_PER_EU      // experiment builds only: ask the allocator to fit that many waves per SIMD (80 VGPRs for 6)
#define RT_MAIN_OCCUPANCY __attribute__((amdgpu_waves_per_eu(RTIOW_MAIN_WAVES_PER_EU)))
#else
#define RT_MAIN_OCCUPANCY
#endif
template <class T, int SRC, bool COUNT, bool BOUND_F32 = false>
__global__ void __launch_bounds__(1024) RT_MAIN_OCCUPANCY render_persistent_kernel(const RenderParams<T> p) { persistent_body<T, SRC, COUNT, false, BOUND_F32>(p); }
template <class T, int SRC, bool COUNT, bool BOUND_F32 = false>
__global__ void __launch_bounds__(1024) render_prepass_kernel(const RenderParams<T> p) { persistent_body<T, SRC, COUNT, false, BOUND_F32>(p); }
// The main launch of a partly filled GPU (small frame, shard of a multi-GPU frame): the same body with the solo
// waves of ColdParams::solo_* compiled in (a kernel of its own, so that the full-frame launch does not carry the
// wave-uniform bookkeeping: +1 % measured).
template <class T, int SRC>
__global__ void __launch_bounds__(1024) render_solo_kernel(const RenderParams<T> p) { persistent_body<T, SRC, false, true>(p); }

#ifdef RTIOW_DEBUG_API       // kernels behind the test hooks of include/rtiow_debug.h
// Elementwise arithmetic probes (tests compare these with the host bit for bit).
template <class T>
__global__ void debug_ops_kernel(int op, size_t n, const T* a, const T* b, const T* c, T* out) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    switch (op) {
        case 0: out[k] = a[k] / b[k]; break;
        case 1: out[k] = Real<T>::sqrt(a[k]); break;
        case 2: out[k] = Real<T>::fma(a[k], b[k], c[k]); break;
        case 3: { uint32_t x; memcpy(&x, &a[k], 4); out[k] = Real<T>::from_u32(x); break; }
        case 4: out[k] = a[k] * b[k] + c[k]; break;
        default: out[k] = 0;
    }
}

// hit_world alone on caller-supplied rays, one per lane (rtiow_debug_hit_world): the tests feed it rays a
// render never produces and compare the scene sources ray by ray.
template <class T>
__global__ void __launch_bounds__(256) hit_probe_kernel(const RenderParams<T> p, const T* __restrict__ rays, int n, T* __restrict__ out_t, int* __restrict__ out_idx) {
    const T* lds_geom = stage_scene<T, RTIOW_SCENE_LDS>(p);
    for (int base = (int)blockIdx.x * (int)blockDim.x; base < n; base += (int)gridDim.x * (int)blockDim.x) {
        const int k = base + (int)threadIdx.x;
        if (k < n) {
            const V3<T> O = {rays[6 * (size_t)k], rays[6 * (size_t)k + 1], rays[6 * (size_t)k + 2]};
            const V3<T> D = {rays[6 * (size_t)k + 3], rays[6 * (size_t)k + 4], rays[6 * (size_t)k + 5]};
            T closest = __builtin_huge_val();
            int hit = -1;
            hit_world<T, RTIOW_SCENE_LDS>(p, lds_geom, O, D, dot3(D, D), closest, hit);
            out_t[k] = closest; out_idx[k] = hit;
        }
    }
}

#endif  // RTIOW_DEBUG_API

}  // namespace
