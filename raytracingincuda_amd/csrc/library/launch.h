// launch.h -- launch_render: LDS layout, schedule (static / persistent / sorted with prepass + cost sort + solo waves), kernel launches
// Host side of librtiow_hip.so; part of the single translation unit rtiow_hip.hip (internal linkage).
#pragma once
#include "scene_tables.h"
#include "../device/render_kernels.h"
#include "../device/cost_sort.h"

namespace {

// A schedule knob: its built-in value, or -- in the tuning build only (-DRTIOW_TUNING, scripts/tune_sweep.py,
// scripts/solo_sweep.py) -- the value of an environment variable.  The product build never reads the environment.
inline int tuned(const char* name, int builtin) {
#ifdef RTIOW_TUNING
    if (const char* e = std::getenv(name)) return std::atoi(e);
#endif
    (void)name;
    return builtin;
}

template <class T> using RenderFn = void (*)(const RenderParams<T>);

template <class T, int SRC, bool COUNT>
RenderFn<T> pick_sched(bool persistent) {
    return persistent ? render_persistent_kernel<T, SRC, COUNT> : render_kernel<T, SRC, COUNT>;
}
template <class T>
RenderFn<T> pick_prepass_kernel(bool lds, bool count) {
    if (lds) return count ? (RenderFn<T>)render_prepass_kernel<T, RTIOW_SCENE_LDS, true> : (RenderFn<T>)render_prepass_kernel<T, RTIOW_SCENE_LDS, false>;
    return count ? (RenderFn<T>)render_prepass_kernel<T, RTIOW_SCENE_SCALAR, true> : (RenderFn<T>)render_prepass_kernel<T, RTIOW_SCENE_SCALAR, false>;
}
// The fp32 persistent kernels with the bounded rejection loop (render_kernels.h, BOUND_F32); fp64 bounds its loop in every kernel.
template <class T> RenderFn<T> pick_bounded_kernel(bool prepass, bool lds, bool count) { (void)prepass; (void)lds; (void)count; return nullptr; }
template <> RenderFn<float> pick_bounded_kernel<float>(bool prepass, bool lds, bool count) {
#define RT_PICK(K) (lds ? (count ? (RenderFn<float>)K<float, RTIOW_SCENE_LDS, true, true> : (RenderFn<float>)K<float, RTIOW_SCENE_LDS, false, true>) \
                        : (count ? (RenderFn<float>)K<float, RTIOW_SCENE_SCALAR, true, true> : (RenderFn<float>)K<float, RTIOW_SCENE_SCALAR, false, true>))
    return prepass ? RT_PICK(render_prepass_kernel) : RT_PICK(render_persistent_kernel);
#undef RT_PICK
}

template <class T>
RenderFn<T> pick_kernel(bool persistent, bool lds, bool count) {
    if (lds) return count ? pick_sched<T, RTIOW_SCENE_LDS, true>(persistent) : pick_sched<T, RTIOW_SCENE_LDS, false>(persistent);
    return count ? pick_sched<T, RTIOW_SCENE_SCALAR, true>(persistent) : pick_sched<T, RTIOW_SCENE_SCALAR, false>(persistent);
}

// (Re)allocates a device buffer kept in the handle when it is too small.
template <class P>
int ensure_buffer(rtiow_handle_s* h, P** ptr, size_t* have, size_t need) {
    if (*ptr && *have >= need) return 0;
    if (*ptr) { HIP_TRY(h, hipFree(*ptr)); *ptr = nullptr; *have = 0; }
    HIP_TRY(h, hipMalloc((void**)ptr, need));
    *have = need;
    return 0;
}

template <class T, class CAM>
int launch_render(rtiow_handle_s* h, const CAM& cam, int bx, int by, int wave_tiles, unsigned long long* seg_counter = nullptr,
                  bool prepare_only = false) {
    RenderParams<T> p = make_params<T>(h, cam);
    p.cold.bx = bx; p.cold.by = by; p.cold.wave_tiles = wave_tiles;
    p.cold.seg_counter = seg_counter ? seg_counter + 1 : nullptr;     // [0] prepass launch, [1] main (or only) launch
    p.lane_cap = 64;
    const bool persistent = h->schedule != RTIOW_SCHED_STATIC;
    const int threads = bx * by;
    // A scene whose tables do not fit the CU's LDS next to the drain scratch (several thousand
    // spheres) is read through the scalar cache instead of failing: same image, exact loop.
    size_t coop_scratch = persistent ? (size_t)((threads + 63) / 64) * COOP_SLOTS * sizeof(CoopSlot<T>) : 0;
    bool lds_source = h->scene_source != RTIOW_SCENE_SCALAR;
    int effective_source = h->scene_source;
    const bool screened = h->scene_source == RTIOW_SCENE_LDS || h->scene_source == RTIOW_SCENE_GRID;
    if (lds_source && (sizeof(T) + (screened ? sizeof(float) : 0)) * 4 * (size_t)h->n_padded + coop_scratch > 160 * 1024) {
        lds_source = false;
        effective_source = RTIOW_SCENE_SCALAR;
    }
    if (lds_source && screened && h->screen_dirty) {
        const auto t0 = std::chrono::steady_clock::now();
        int rc = build_screen_table<T>(h);
        if (rc) return rc;
        if ((rc = build_grid_tables<T>(h))) return rc;
        h->stats.scene_prepare_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    fill_screen_params<T>(p, h);
    if (!lds_source) p.use_screen = 0;
    size_t lds = lds_source ? sizeof(T) * 4 * (size_t)h->n_padded : 0;
    p.screen_offset = (int)lds;
    if (p.use_screen) lds += sizeof(float) * 4 * (size_t)h->n_padded;
    p.cold.timeline = nullptr;                                    // set below, once the grid is known
    p.cold.clock_stamps = (!seg_counter && h->schedule != RTIOW_SCHED_STATIC && h->clock_stamps_dev) ? h->clock_stamps_dev + 4 : nullptr;   // the main (or only) launch; timed renders only
    p.cold.pixel_times = seg_counter ? h->pixel_times : nullptr;
    // shade records ride along in LDS while a workgroup's share stays within 1/5 of the CU's LDS
    const size_t coop_bytes = coop_scratch;
    const size_t shade_bytes = (sizeof(T) * 12 * (size_t)h->n + 15) / 16 * 16;
    p.shade_offset = (int)lds;
    // ... and a wave's share stays under ~6.5 KB, so that LDS never caps occupancy below 6 waves/SIMD
    const size_t waves_in_block = (size_t)((threads + 63) / 64);
    p.shade_in_lds = (lds + shade_bytes + coop_bytes <= 32 * 1024 && (lds + shade_bytes + coop_bytes) / waves_in_block <= 6656) ? 1 : 0;
    if (p.shade_in_lds) lds += shade_bytes;
    p.coop_offset = (int)lds;                                // a multiple of 16
    lds += coop_bytes;
    // the grid blob (cells | fp32 AoS table | direct table | direct ids) goes last
    p.grid = GridParams{};
    p.use_grid = 0;
    if (!seg_counter) { h->stats.grid_nx = h->stats.grid_nz = h->stats.grid_registered = h->stats.grid_direct = 0; h->stats.grid_cell = 0; }
    if (lds_source && h->scene_source == RTIOW_SCENE_GRID && p.use_screen && h->grid.use_grid && lds + (size_t)h->grid.blob_bytes <= 160 * 1024) {
        p.grid = h->grid;
        p.grid.cells_offset = (int)lds;
        p.grid.aos_offset = p.grid.cells_offset + h->grid_cells_bytes;
        p.grid.direct_offset = p.grid.aos_offset + h->grid_aos_bytes;
        p.grid.direct_ids_offset = p.grid.direct_offset + h->grid_direct_bytes;
        lds += (size_t)h->grid.blob_bytes;
        p.use_grid = 1;
        if (!seg_counter) { h->stats.grid_nx = h->grid.nx; h->stats.grid_nz = h->grid.nz; h->stats.grid_registered = h->grid_registered; h->stats.grid_direct = h->grid_direct; h->stats.grid_cell = h->grid.cell; }
    } else if (effective_source == RTIOW_SCENE_GRID) effective_source = RTIOW_SCENE_LDS;   // no grid for this scene: the screened loop
    if (lds > 160 * 1024) return fail_arg(h, RTIOW_E_BADARG, "scene too large for LDS staging; use RTIOW_SCENE_SCALAR");
#ifdef RTIOW_DEBUG_API
    if (h->probe_n > 0) {                                    // rtiow_debug_hit_world: the tables are laid out, run hit_world on the caller's rays
        if (!lds_source) return fail_arg(h, RTIOW_E_STATE, "rtiow_debug_hit_world needs an LDS scene source");
        if (lds > 64 * 1024) HIP_TRY(h, hipFuncSetAttribute((const void*)hit_probe_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int blocks = std::min(2048, (h->probe_n + 255) / 256);
        hipLaunchKernelGGL(hit_probe_kernel<T>, dim3(blocks), dim3(256), lds, h->stream, p, (const T*)h->probe_rays, h->probe_n, (T*)h->probe_t, h->probe_idx);
        HIP_TRY(h, hipGetLastError());
        if (!seg_counter) h->stats.scene_source = effective_source;
        return 0;
    }
#endif
    RenderFn<T> k = pick_kernel<T>(persistent, lds_source, seg_counter != nullptr);
    if (lds > 64 * 1024) HIP_TRY(h, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipFuncAttributes fa{};
    HIP_TRY(h, hipFuncGetAttributes(&fa, (const void*)k));
    dim3 grid, block(threads);
    int phases = 1;
    if (persistent) {
        if (!h->work_counter) HIP_TRY(h, hipMalloc((void**)&h->work_counter, 2 * sizeof(unsigned int)));
        HIP_TRY(h, hipMemsetAsync(h->work_counter, 0, 2 * sizeof(unsigned int), h->stream));
        int per_cu = 0;
        HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k, threads, lds));
        if (per_cu < 1) per_cu = 1;
        const int waves_per_block = (threads + 63) / 64;
        if (h->waves_per_simd > 0) {                       // knob: fewer resident waves, more pixels per lane
            const int cap = (h->waves_per_simd * 4 + waves_per_block - 1) / waves_per_block;
            if (cap < per_cu) per_cu = cap;
        }
        const long long tile_slots = (long long)((p.cold.W + 7) / 8) * ((h->local_rows + 7) / 8) * POOL;
        // Underfilled launch (fewer 64-pixel pools than resident waves: small frames): let only the first
        // `lane_cap` lanes of every wave take pixels.  More waves are busy, each permanently in the
        // cooperative mode, where its idle lanes split the sphere loops of the live ones: a trip gets
        // shorter, and with so little work the frame is as long as its longest chain of trips.
        // Measured (profiles/archive/r01_lane_cap_sweep.txt): scene 1 320x192x10 2.27 -> 1.06 ms, 640x384x100
        // 19.4 -> 17.0 ms; frames with at least one pool per wave are unchanged (cap 64).
        int lane_cap = 64;
        {
            const long long pools = tile_slots / POOL, waves = (long long)h->num_cus * per_cu * waves_per_block;
            while (lane_cap > 16 && pools * (64 / lane_cap) < waves) lane_cap >>= 1;  // the largest share that keeps every wave busy; not below 16 (with the grid walk 8-lane waves lose: scene 1 320x192x100 6.85 vs 5.96 ms, profiles/archive/r02_lane_cap_sweep.jsonl)
            lane_cap = tuned("RTIOW_TUNE_LANE_CAP", lane_cap);
        }
        p.lane_cap = lane_cap;
        // fp32: the bounded rejection loop where throughput binds -- at least four pools per resident wave (1080p: 6.3; 1280 x 720, shards and small
        // frames end with one chain's latency and keep the blocking loop) -- if that kernel keeps the occupancy this launch was sized for
        bool bounded_f32 = false;
        if (sizeof(T) == 4 && h->schedule != RTIOW_SCHED_STATIC) {
            const long long pools = tile_slots / POOL, waves = (long long)h->num_cus * per_cu * waves_per_block;
#ifdef RTIOW_TUNING
            const bool want = tuned("RTIOW_TUNE_RUV_BOUNDED", pools >= 4 * waves ? 1 : 0) != 0;
#else
            const bool want = pools >= 4 * waves;
#endif
            if (want) {
                RenderFn<T> kb = pick_bounded_kernel<T>(false, lds_source, seg_counter != nullptr);
                if (lds > 64 * 1024) HIP_TRY(h, hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                int per_cu_b = 0;
                HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, (const void*)kb, threads, lds));
                if (per_cu_b >= per_cu) { k = kb; bounded_f32 = true; HIP_TRY(h, hipFuncGetAttributes(&fa, (const void*)k)); }
            }
        }
        long long blocks = (long long)h->num_cus * per_cu;
        const long long per_block = (long long)waves_per_block * lane_cap;
        const long long useful = (tile_slots + per_block - 1) / per_block;
        if (blocks > useful) blocks = useful;               // never more waves than lane_cap-pixel shares of the pools
        grid = dim3((unsigned)blocks);

        const int npix = p.cold.W * h->local_rows;
        const int S = p.cold.S;
        // prepass length: enough samples to rank the pixels, a small share of the frame
        int SA = S >= 64 ? 3 : (S >= 24 ? 2 : 0);
        SA = tuned("RTIOW_TUNE_SA", SA);       // measured on the headline config: 1 -> 25.5 ms, 2 -> 22.5, 3 -> 22.1, 4 -> 22.4, 8 -> 23.1
        p.cold.work_counter = h->work_counter;
        p.cold.s_begin = 0; p.s_end = S; p.cold.rng_in = h->rng; p.cold.mid_in = nullptr; p.cold.mid_out = nullptr;
        p.cold.cost_out = nullptr; p.cold.order = nullptr; p.cold.total_slots = (int)tile_slots; p.cold.first_pools = 0;
        p.cold.solo_waves = 0; p.cold.solo_lanes = 1; p.cold.stage_by_slot = 0;
        if (h->schedule == RTIOW_SCHED_SORTED && SA > 0 && npix >= 4096 && p.cold.W < 65536 && h->local_rows < 32768) {   // (the order's entries are row << 16 | column)
            phases = 2;
            const int total_pools = (npix + POOL - 1) / POOL;
            int rc;
            if ((rc = ensure_buffer(h, &h->mid, &h->mid_bytes, (size_t)npix * sizeof(MidState<T>)))) return rc;
            if ((rc = ensure_buffer(h, &h->cost, &h->cost_bytes, (size_t)npix * sizeof(uint32_t)))) return rc;
            if ((rc = ensure_buffer(h, &h->cost_rank, &h->cost_rank_bytes, (size_t)npix * sizeof(uint32_t)))) return rc;
            // Solo waves (ColdParams::solo_*, render_solo_kernel).  A shard or small frame ends with its longest sample
            // chains (one pixel = one sequential chain), and a chain advances at the pace of its wave: 2452 segments at
            // ~3 us per trip among 63 other pixels.  Two heavy pixels alone in a wave share every sphere loop with the
            // idle lanes and skip the divergent work of wave-mates.  Which pixels: the top of the cost ranking.  How
            // many waves: more than ~5 % of the resident waves cost more throughput than the chains gain; measured per
            // fill level (profiles/archive/r02_handout_study/): 1/8 frame 6.96 -> 5.65 ms with 128 waves (5.78 with 256),
            // 1/4 frame 7.85 -> 6.61 with 256 (7.03 with 128), 1/2 frame 8.43 -> 8.21, 1280x720 8.82 -> 8.34; the full
            // frame (6.3 pools per wave) loses 1-2 % and keeps the plain kernel.  Outlier chains need a bounce limit
            // that lets rare long paths exist: at 10 bounces the solo waves cost 4-11 % on both scenes, at 25 scene 3
            // gains 9 % and scene 1 -- the reference's own benchmark grid -- loses 2-5 %, from 50 on both gain
            // (sweep6_bounce_limit.txt): the rule asks for more than 32.
            const double fill_level = (double)total_pools / (double)(blocks * waves_per_block);
            int solo_waves = (seg_counter || p.B < 32) ? 0 : (fill_level < 1.2 ? 128 : (fill_level < 4.0 ? 256 : 0)), solo_lanes = 2;
            solo_waves = tuned("RTIOW_TUNE_SOLO_WAVES", solo_waves);
            solo_lanes = tuned("RTIOW_TUNE_SOLO_LANES", solo_lanes);
            if (solo_lanes < 1) solo_lanes = 1;
            RenderFn<T> k_solo = lds_source ? (RenderFn<T>)render_solo_kernel<T, RTIOW_SCENE_LDS> : (RenderFn<T>)render_solo_kernel<T, RTIOW_SCENE_SCALAR>;
            if (solo_waves > 0) {
                if (lds > 64 * 1024) HIP_TRY(h, hipFuncSetAttribute((const void*)k_solo, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                int per_cu_solo = 0;
                HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_solo, (const void*)k_solo, threads, lds));
                if ((long long)per_cu_solo * h->num_cus < blocks) solo_waves = 0;   // its workgroups must all be resident, as the first pools assume
            }
            if (solo_lanes > lane_cap) solo_lanes = lane_cap;
            if (solo_waves > (int)blocks) solo_waves = (int)blocks;
            if ((long long)solo_waves * solo_lanes > npix / 2) solo_waves = npix / 2 / solo_lanes;
            const int solo_slots = solo_waves * solo_lanes;
            if ((rc = ensure_buffer(h, &h->order, &h->order_bytes, ((size_t)total_pools * POOL + (size_t)solo_slots) * sizeof(int)))) return rc;
            if ((rc = ensure_buffer(h, &h->sort_scratch, &h->sort_scratch_bytes, (size_t)3 * COST_BINS * sizeof(unsigned)))) return rc;
            // finished pixels go to their slot in a staging buffer and place_pixels_kernel writes the image (ColdParams::stage_by_slot)
#ifdef RTIOW_DIRECT_STORES
            const bool staged_stores = false;           // A/B build: every lane stores its pixel at its place in the image when it finishes
#else
            const bool staged_stores = true;
#endif
            const size_t total_slots = (size_t)total_pools * POOL + (size_t)solo_slots;
            if (staged_stores) {
                if ((rc = ensure_buffer(h, &h->slot_of, &h->slot_of_bytes, (size_t)npix * sizeof(int)))) return rc;
                if ((rc = ensure_buffer(h, &h->staged, &h->staged_bytes, total_slots * 3 * sizeof(T)))) return rc;
            }
            if (prepare_only) return 0;                  // every table and buffer of this configuration now exists
            // ---- prepass: samples [0, SA) in tile order through the same persistent body (the static
            // kernel keeps only ~40 % of its lanes busy over a few samples: 2.6 ms vs 1.4 ms measured
            // for 4 samples); RNG state, colour sum and segment count are parked per pixel.
            RenderParams<T> pa = p;
            pa.s_end = SA; pa.cold.mid_out = h->mid; pa.cold.cost_out = h->cost;
            pa.cold.seg_counter = seg_counter;
            if (pa.cold.clock_stamps) pa.cold.clock_stamps = h->clock_stamps_dev;          // the prepass's four words
            RenderFn<T> kp = bounded_f32 ? pick_bounded_kernel<T>(true, lds_source, seg_counter != nullptr) : pick_prepass_kernel<T>(lds_source, seg_counter != nullptr);
            if (lds > 64 * 1024) HIP_TRY(h, hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kp, grid, block, lds, h->stream, pa);
            if (h->time_phases) HIP_TRY(h, hipEventRecord(h->ev_a, h->stream));
            h->stats.prepass_samples = SA;
            HIP_TRY(h, hipGetLastError());
            // ---- rank the pixels by measured cost, heavy first, dealt into balanced pools.
            // Blocks of the order are one "age class" of resident waves wide (see first_pools).
            unsigned* hist = h->sort_scratch; unsigned* start = hist + COST_BINS; unsigned* fill = start + COST_BINS;
            HIP_TRY(h, hipMemsetAsync(hist, 0, COST_BINS * sizeof(unsigned), h->stream));
            HIP_TRY(h, hipMemsetAsync(h->order, 0xff, ((size_t)total_pools * POOL + (size_t)solo_slots) * sizeof(int), h->stream));
            const int sort_blocks = (npix + 255) / 256;
            const uint32_t* rank_by = h->cost;
            int smooth_hw = 6;                          // 13 x 13 window: profiles/archive/r02_cost_smoothing_sweep.jsonl
            smooth_hw = tuned("RTIOW_TUNE_SMOOTH", smooth_hw);
            if (smooth_hw > 0) {
                if (smooth_hw > 24) smooth_hw = 24;      // 2 x (tile + halo) words of LDS: 37 KB at 24
                const int smooth_blocks = ((p.cold.W + SMOOTH_TW - 1) / SMOOTH_TW) * ((h->local_rows + SMOOTH_TH - 1) / SMOOTH_TH);
                const size_t smooth_lds_bytes = ((size_t)(SMOOTH_TW + 2 * smooth_hw) + SMOOTH_TW) * (size_t)(SMOOTH_TH + 2 * smooth_hw) * sizeof(uint32_t);
                // one rank: its strips are adjacent in the image, the window may cross them (it did not before: 13 x <= 8 rows)
                int window_strip = h->nranks == 1 ? (h->local_rows > 0 ? h->local_rows : 1) : h->strip_rows;
                if (const int ws = tuned("RTIOW_TUNE_SMOOTH_STRIP", -1); ws >= 0) window_strip = ws > 0 ? ws : (h->local_rows > 0 ? h->local_rows : 1);
                hipLaunchKernelGGL(cost_smooth_kernel, dim3(smooth_blocks), dim3(256), smooth_lds_bytes, h->stream, h->cost, h->cost_rank, p.cold.W, h->local_rows, window_strip, smooth_hw, hist);
                rank_by = h->cost_rank;
            } else {
                hipLaunchKernelGGL(cost_hist_kernel, dim3(sort_blocks < 1024 ? sort_blocks : 1024), dim3(256), 0, h->stream, rank_by, npix, hist);
            }
            hipLaunchKernelGGL(cost_scan_kernel, dim3(1), dim3(COST_BINS), 0, h->stream, hist, start, fill);
            const int resident_waves = (int)blocks * waves_per_block;
            const int age_classes = (int)((blocks + h->num_cus - 1) / h->num_cus);
            int pools_per_block = (resident_waves + age_classes - 1) / age_classes;
            if (pools_per_block > total_pools) pools_per_block = total_pools;
            // Deal granularity: `deal_group` consecutive ranks (= neighbouring pixels of equal cost) stay
            // in one pool, the groups go round-robin over the block's pools.  Coherent groups mean fewer
            // distinct spheres pass the screen per wave (8.9 exact blocks per wave-iteration with single
            // ranks vs 3.6 in tile order); mixed costs in a pool let a heavy pixel finish in the fast
            // cooperative mode, which is what small shards need.  Measured (profiles/archive/r01_deal_group_sweep.txt):
            // full frame 24.1 -> 22.5 ms with 16-32, half frame 14.7 -> 14.0 with 8, quarter and eighth
            // frames are fastest with 1.
            const double pools_per_wave = (double)total_pools / (double)resident_waves;
            // With the grid walk (a lane's cost follows ITS ray) coherence pays more: whole pools of 64 neighbouring
            // ranks, 15.3 -> 14.7 ms on the full frame (profiles/archive/r02_tune_sweep.jsonl) and, once the ranks come from
            // the smoothed cost, on every frame with at least 2.5 pools per wave (1280x720: 9.3 ms with groups of 1,
            // 11.4 with 8, 8.7 with 64; profiles/archive/r02_cost_smoothing_sweep.jsonl); smaller shards keep single ranks.
            int deal_group = pools_per_wave >= 2.5 ? 64 : 1;
            deal_group = tuned("RTIOW_TUNE_DEAL", deal_group);
            const int scatter_blocks = ((p.cold.W + 63) / 64) * ((h->local_rows + 63) / 64);   // one per 64 x 64 super-tile
            hipLaunchKernelGGL(cost_scatter_kernel, dim3(scatter_blocks), dim3(1024), 0, h->stream, rank_by, p.cold.W, h->local_rows, start, fill, h->order,
                               pools_per_block, total_pools, deal_group, solo_slots, staged_stores ? h->slot_of : nullptr);
            HIP_TRY(h, hipGetLastError());
            // ---- main launch: samples [SA, S) in that order
            p.cold.s_begin = SA; p.cold.mid_in = h->mid; p.cold.order = h->order;
            p.cold.total_slots = solo_slots + total_pools * POOL;
            p.cold.work_counter = h->work_counter + 1;
            p.cold.first_pools = 1;
            p.cold.solo_waves = solo_waves; p.cold.solo_lanes = solo_lanes;
            if (staged_stores) { p.cold.stage_by_slot = 1; p.cold.fb = (T*)h->staged; }
            if (solo_waves > 0) {
                k = k_solo;
                HIP_TRY(h, hipFuncGetAttributes(&fa, (const void*)k));
            }
            const unsigned counter_start = (unsigned)solo_slots + (unsigned)(resident_waves - solo_waves) * (unsigned)lane_cap;
            HIP_TRY(h, hipMemsetD32Async((hipDeviceptr_t)(h->work_counter + 1), (int)counter_start, 1, h->stream));
        }
    } else {
        grid = dim3((p.cold.W + bx - 1) / bx, (h->local_rows + by - 1) / by);
        p.cold.s_begin = 0; p.s_end = p.cold.S; p.cold.rng_in = h->rng; p.cold.mid_in = nullptr; p.cold.mid_out = nullptr;
        p.cold.cost_out = nullptr; p.cold.order = nullptr; p.cold.total_slots = 0; p.cold.first_pools = 0; p.cold.work_counter = nullptr;
        p.cold.solo_waves = 0; p.cold.solo_lanes = 1; p.cold.stage_by_slot = 0;
    }
    if (prepare_only) return 0;
    if (seg_counter) {
        h->last_count_blocks = (int)(grid.x * grid.y);
        h->last_count_waves_per_block = (threads + 63) / 64;
        // the kernel writes 8 words per wave: hand the buffer over only if it holds every wave of this launch
        if (h->timeline && (size_t)h->last_count_blocks * h->last_count_waves_per_block <= h->timeline_cap_waves) p.cold.timeline = h->timeline;
    }
    if (h->time_phases && phases == 2) HIP_TRY(h, hipEventRecord(h->ev_b, h->stream));
    hipLaunchKernelGGL(k, grid, block, lds, h->stream, p);
    HIP_TRY(h, hipGetLastError());
    if (p.cold.stage_by_slot) {                             // slot order -> image, in whole lines
        if (h->time_phases) HIP_TRY(h, hipEventRecord(h->ev_c, h->stream));
        const int npix = p.cold.W * h->local_rows;
        hipLaunchKernelGGL(place_pixels_kernel<T>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, h->stream, (const T*)h->staged, h->slot_of, (T*)h->fb, npix);
        HIP_TRY(h, hipGetLastError());
    }
    if (!seg_counter) {
        h->stats.vgprs = fa.numRegs;
        h->stats.sgprs = 0;
        h->stats.lds_bytes = (int)(lds + fa.sharedSizeBytes);
        h->stats.block_x = bx; h->stats.block_y = by;
        h->stats.scene_source = effective_source;
        h->stats.schedule = h->schedule;
        h->stats.grid_blocks = (int)(grid.x * grid.y);
        h->stats.phases = phases;
        if (phases == 1) h->stats.prepass_samples = 0;
        h->stats.solo_waves = phases == 2 ? p.cold.solo_waves : 0;
        h->stats.solo_lanes = phases == 2 && p.cold.solo_waves > 0 ? p.cold.solo_lanes : 0;
        h->stats.staged_stores = p.cold.stage_by_slot;
    }
    return 0;
}

// T (the reference's --threads) shapes the workgroup of RTIOW_SCHED_STATIC, whose lanes ARE the
// pixels of a T x T block.  The dynamic schedules hand pixels to lanes themselves, so a workgroup
// there is just four waves whatever T says (measured with T as the workgroup size: 69 / 22.3 / 22.3 /
// 33 / 26 ms for T = 4 / 8 / 16 / 24 / 32 -- partly filled waves and uneven SIMD packing).
void block_shape(int T, bool static_schedule, int& bx, int& by, int& wave_tiles) {
    if (!static_schedule) T = 0;
    if (T == 0) { bx = 16; by = 16; wave_tiles = 1; }       // library tiling: 4 waves, each an 8x8 tile
    else if (T == 8) { bx = 8; by = 8; wave_tiles = 1; }    // == the reference's 8x8 block (one wave)
    else { bx = T; by = T; wave_tiles = 0; }                 // the reference's T x T row-major block
}

}  // namespace
