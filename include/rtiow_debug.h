/* rtiow_debug.h -- test hooks of the HIP render library.  NOT part of the product: librtiow_hip.so exports none of these; the test build
 * lib/librtiow_hip_debug.so (the same sources compiled with -DRTIOW_DEBUG_API, raytracingincuda_amd/build.py) does, and only tests/, the
 * study scripts and the profiling scripts load it (raytracingincuda_amd.api: Renderer(..., debug=True), load_hip_library(debug=True)). */
#ifndef RTIOW_DEBUG_H
#define RTIOW_DEBUG_H
#include "rtiow.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Device RNG states after rtiow_init_rng, as
 * local_pixels x 6 uint32 {v0..v4,d}; and elementwise device arithmetic probes that the
 * parity tests compare bit-for-bit with the host (op: 0 a/b, 1 sqrt(a), 2 fma(a,b,c),
 * 3 uniform(u32 a -> T), 4 a*b+c unfused). */
int rtiow_debug_read_rng(rtiow_handle h, uint32_t* host_states, size_t count_words);
/* After a render with RTIOW_SCHED_SORTED that sorted (rtiow_stats.phases == 2): per local pixel the prepass cost
 * (path segments of its prepass samples) and the key the sort ranked it by (the mean of that cost over the
 * pixel's neighbourhood inside its row strip, in quarter segments).  count = local pixels. */
int rtiow_debug_read_costs(rtiow_handle h, uint32_t* own, uint32_t* smoothed, size_t count);
/* Per-wave timeline of one (untimed, counting) persistent render: 8 words per wave
 * {t_start, t_pool_exhausted, t_end (100 MHz ticks), iterations alone, iterations cooperative,
 * pixels taken, 0, 0}. */
int rtiow_debug_timeline(rtiow_handle h, int threads_per_block_row, uint64_t* out_words, size_t cap_words, int* waves);
/* One (untimed, counting) persistent render: per local pixel 4 words {when a lane took it, when it finished (100 MHz ticks, low 32 bits),
 * segments it ran in the launch, wave that ran it} of the LAST launch that rendered it (the main launch of the sorted schedule). */
int rtiow_debug_pixel_times(rtiow_handle h, int threads_per_block_row, uint32_t* out_words, size_t cap_words);
int rtiow_debug_ops(rtiow_handle h, int op, size_t n, const void* a, const void* b, const void* c, void* out);
/* hit_world (hittable.h:80-98) alone, with the handle's scene and scene source, on n caller-supplied rays
 * {ox,oy,oz,dx,dy,dz} in the handle's precision: nearest root (+inf: none) and sphere index (-1: none) per ray.
 * Lets the tests compare the scene sources on rays no render produces. */
int rtiow_debug_hit_world(rtiow_handle h, int n, const void* rays, void* t_out, int32_t* index_out);
/* The 32 XORWOW subsequence-jump matrices A^(2^(67+b)) (160 x 5 words each) as the library builds
 * them: from its committed constant A^(2^67), or from_scratch != 0 from the one-step matrix A.
 * Host arithmetic only (no GPU needed).  Returns the number of matrices. */
int rtiow_debug_jump_matrices(uint32_t* out_words, size_t cap_words, int from_scratch);
/* The uniform-grid plan of RTIOW_SCENE_GRID for n spheres {cx,cy,cz,r} (doubles) around the recentring point
 * centre3, as the library builds it.  Host arithmetic only (no GPU needed).  dims4 = {nx, nz, registered
 * spheres, direct-list length}; params8 = {x0, z0, cell, slab ylo, slab yhi, Rfar, eps, Cmax}; cells (optional)
 * = nx*nz*4 sphere indices (0xffff x4: empty cell, n: pad); direct (optional) = the direct list; halfwidth
 * (optional, n entries) = registration half-width of every gridded sphere.  Returns 1 when the library would
 * use the grid for this scene, 0 when it keeps the screened loop, negative on bad arguments. */
int rtiow_debug_grid_plan(int n, const double* center_radius, const double* centre3, int32_t* dims4, double* params8,
                          uint16_t* cells, size_t cells_cap, int32_t* direct, size_t direct_cap, double* halfwidth);

/* Test hook, host only (no GPU, no RCCL): the exchange's schedule -- the very function rtiow_group_gather runs --
 * against a recording table instead of HIP / RCCL, for n ranks on `devices` with `rows` local rows each.
 * mode = RTIOW_GATHER_RCCL, RTIOW_GATHER_PEER or RTIOW_GATHER_HOST, optionally | 0x100: with the fallback chain of RTIOW_GATHER_AUTO
 * (RCCL -> peer copies -> host-staged copies; the last record is then {12, -, the transport that carried the image}).  One record of
 * 8 int64 per call (layout: csrc/rtiow_group.hip); fail_at >= 0 makes that call fail.  Returns the number of records;
 * *schedule_rc = what the schedule returned. */
int rtiow_debug_gather_schedule(int n, const int* devices, const int* rows, int W, int precision, int mode, int fail_at,
                                int64_t* records, size_t cap_records, int* schedule_rc);


/* Instrumented builds only (-DRTIOW_PATH_STATS, scripts/path_stats_probe.py): wave-level execution counts and region clocks. */
int rtiow_debug_region_cycles(unsigned long long* out, int cap_words, int reset);
int rtiow_debug_path_stats(unsigned long long* out, int cap_words, int reset);

#ifdef __cplusplus
}
#endif
#endif /* RTIOW_DEBUG_H */
