// params.h -- kernel arguments: RenderParams (hot) / ColdParams / GridParams and their kernarg accessors
// Part of the single gfx950 translation unit rtiow_hip.hip (included there, in this order; internal linkage).
#pragma once
#include "xorwow.h"

namespace {

// =====================================================================================
// render
// =====================================================================================
template <class T> struct V3 { T x, y, z; };

// Uniform grid over the small spheres of the scene (hit_world_grid).  x/z run over the cells, the
// y extent of the gridded spheres is one slab.  All coordinates are fp32 and relative to nothing:
// x0/z0 are subtracted by the kernel.  Offsets are bytes from the start of dynamic LDS.
struct GridParams {
    int use_grid;
    int nx, nz;
    float x0, z0, cell, inv_cell;     // cell (ix, iz) covers [x0 + ix*cell, x0 + (ix+1)*cell) x [z0 + iz*cell, ...)
    float ylo, yhi;                   // slab of the registered (inflated) spheres
    float far2;                       // |O - ctr|^2 above this: the per-sphere registration margin no longer covers the reference's rounding noise
    float core_lo[3], core_hi[3];     // box of the gridded spheres' CENTRES (far rays are clipped against it, inflated per ray)
    float rmax2, cmax;                // largest gridded radius squared; largest |C - ctr| over the gridded spheres
    int n_direct_padded;              // spheres every ray tests exactly (too big for a cell, or the overflow of a full cell), padded to x4
    int cells_offset, aos_offset, direct_offset, direct_ids_offset;
    const unsigned char* __restrict__ blob;   // cells | aos | direct table | direct ids, as laid out in LDS from cells_offset on
    int blob_bytes;
};

// Launch parameters, split by how often the kernel needs them.  The HOT part (camera, table
// offsets, loop bounds) stays in SGPRs for the whole kernel.  The COLD part (buffer pointers, image
// and shard geometry, sort hand-over) is needed only when a lane takes a new pixel or finishes one:
// it is read with scalar loads from the kernarg segment at those sites (cold_of), so it does not
// occupy ~35 SGPRs during the path loop (the all-by-value form spilled 45 SGPRs to VGPR lanes, with
// 75 v_readlane/v_writelane moves inside the loop).
template <class T> struct ColdParams {
    int W, H, S;
    T pixel_samples_scale;
    const uint32_t* __restrict__ rng; // [6][npix_local] SoA
    T* __restrict__ fb;               // [local_rows][W][3]
    int local_rows, rank, nranks, strip_rows;
    int bx, by;                       // tile (block) shape in pixels (static schedule)
    int wave_tiles;                   // 1: lanes of a wave form 8x8 tiles inside the block
    unsigned long long* seg_counter;  // COUNT variant only: [0] total hit_world calls (path segments) of this launch, [2] the longest per-pixel chain
    unsigned int* work_counter;       // SCHED_PERSISTENT: next unassigned pixel slot (zeroed per launch)
    // SCHED_SORTED (two phases of the persistent kernel): first sample of this launch, the
    // per-pixel state carried between the phases, and the cost-sorted hand-out order.
    int s_begin;                      // samples [s_begin, s_end) of every pixel
    const uint32_t* __restrict__ rng_in;   // [6][npix] SoA state at sample s_begin
    // SCHED_SORTED hand-over between the prepass and the main launch: ONE record per pixel
    // (MidState<T>: RNG state after sample s_end-1 + colour sum), so that the main launch, which
    // visits the pixels in cost order, fetches one or two cache lines per pixel instead of nine
    // (SoA cost 630 MB of fetches per frame for 83 MB of state).
    const unsigned char* __restrict__ mid_in;   // main launch: state at sample s_begin (nullptr: rng_in, zero sum)
    unsigned char* __restrict__ mid_out;        // prepass: park the state (nullptr: final launch, the pixel is stored)
    uint32_t* __restrict__ cost_out;  // prepass only: segments the pixel ran in this launch
    const int* __restrict__ order;    // slot -> local pixel as (local row << 16 | column), or -1; nullptr: 8x8 tiles bottom-up (frames wider than 65535 or taller than 32767 keep the tile order)
    // Main launch of the sorted schedule: a finished pixel is stored at its SLOT in a staging buffer (`fb` then points
    // there), not at its place in the image.  Lanes store 12-byte pixels whenever they finish; in image order the
    // partial lines of neighbouring pixels come from different waves on different XCDs (1.8 x write amplification,
    // measured); a wave's slots are 768 contiguous bytes that only that wave writes, and place_pixels_kernel then
    // writes the image in whole lines.
    int stage_by_slot;
    int total_slots;
    int first_pools;                  // 1: wave w starts with pool w (the work counter then starts at the wave count)
    // Solo waves: the first solo_waves*solo_lanes slots of the order (the heaviest pixels) go solo_lanes each to
    // wave 0 of the first solo_waves workgroups, which take nothing else until those pixels are done.
    int solo_waves, solo_lanes;
    unsigned long long* timeline;     // COUNT variant, optional: per wave {t_start, t_exhausted, t_end, iters_normal, iters_coop, pixels, 0, 0}
    uint32_t* pixel_times;            // COUNT variant, optional (rtiow_debug_pixel_times): per local pixel {taken, finished (100 MHz ticks, low 32 bits), segments in this launch, wave}
    // Effective shader clock of the launch (rtiow_stats.main_clock_mhz / prepass_clock_mhz): wave 0 of workgroup 0 -- dispatched first, and resident
    // until the hand-out runs dry -- stores {s_memtime, s_memrealtime} when it starts and when it ends; clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
    // (MI355X_MICROARCH.md, DVFS).  Pinned host memory, written by one lane twice per launch; nullptr: no stamps.
    unsigned long long* clock_stamps;
};

// The camera (camera.h:10-30 as camera::initialize leaves it): 19 scalars that only gen_primary reads, once per
// primary ray.  Like the cold part they are re-read from the kernarg segment where they are used (cam_of) instead
// of occupying 19-38 SGPRs for the whole path loop.
template <class T> struct CameraParams {
    V3<T> center, pixel00, du, dv;
    T defocus_angle;
    V3<T> ddu, ddv;
};

// The fp32 screening table and the recentring point of screen and grid: read where hit_world starts (screen_of).
template <class T> struct ScreenParams {
    // fp32 screening table (hit_world_screened): recentred centres and q' = |C'|^2 - r^2 - margin,
    // pair-interleaved like geom_a; staged in LDS behind geom_a (screen_offset bytes)
    const float* __restrict__ geom_s;
    T ctr_x, ctr_y, ctr_z, omax2;     // recentring point; omax2 = 2 Cmax of the per-ray margin term
    // everything the shade step needs about the sphere that was hit, 12 T per sphere:
    // {cx,cy,cz,1/r | albedo r,g,b,fuzz (dielectric: Schlick r0^2 front, back, -, -) | eta, 1/eta, material type, 0};
    // the global copy is read only when the records do not ride along in LDS (shade_in_lds == 0)
    const T* __restrict__ shade_tbl;
};

template <class T> struct RenderParams {
    int B, s_end;                     // bounce limit; this launch renders samples [cold.s_begin, s_end)
    int lane_cap;                     // lanes of a wave that take pixels (64; fewer when the launch is underfilled, see launch_render)
    int range_flags;                  // host-checked operand ranges.  bit 0 (primary_rays_in_range): |D|^2 of every primary ray lies well
                                      // inside [2^-80, 2^80]; bit 1 (scene_in_range): every coordinate of spheres and lens is below 2^18
    CameraParams<T> cam;              // read through cam_of
    int n, n_padded;                  // spheres, and the table length padded to a multiple of 4
    const T* __restrict__ geom_a;     // [n_padded][4] cx,cy,cz,r*r (sphere loop; padding never hits)
    int use_screen, screen_offset;
    ScreenParams<T> screen;           // read through screen_of
    int shade_in_lds;                 // 1: the table is staged behind the loop table in LDS (shade_offset bytes)
    int shade_offset;
    int coop_offset;                  // SCHED_PERSISTENT: byte offset of the per-wave coop scratch in LDS
    int use_grid;                     // RTIOW_SCENE_GRID: hit_world_grid (its description below is read through grid_of)
    GridParams grid;
    ColdParams<T> cold;
};

// The cold half of the kernel's own argument, re-read from the kernarg segment.  The empty asm
// makes the base pointer opaque at every call site, so the scalar loads stay inside the (rare)
// block that needs them instead of being hoisted to the kernel entry and kept live.
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) ColdParams<T>& cold_of(const RenderParams<T>&) {
    typedef const __attribute__((address_space(4))) char* kptr;
    kptr k = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *(const __attribute__((address_space(4))) ColdParams<T>*)(k + offsetof(RenderParams<T>, cold));
}
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) CameraParams<T>& cam_of(const RenderParams<T>&) {
    typedef const __attribute__((address_space(4))) char* kptr;
    kptr k = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *(const __attribute__((address_space(4))) CameraParams<T>*)(k + offsetof(RenderParams<T>, cam));
}
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) ScreenParams<T>& screen_of(const RenderParams<T>&) {
    typedef const __attribute__((address_space(4))) char* kptr;
    kptr k = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *(const __attribute__((address_space(4))) ScreenParams<T>*)(k + offsetof(RenderParams<T>, screen));
}
// Same for the grid description: ~25 scalars that only hit_world_grid needs, loaded at its entry.
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) GridParams& grid_of(const RenderParams<T>&) {
    typedef const __attribute__((address_space(4))) char* kptr;
    kptr k = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *(const __attribute__((address_space(4))) GridParams*)(k + offsetof(RenderParams<T>, grid));
}

#define RT_FMA(a, b, c) Real<T>::fma((a), (b), (c))

}  // namespace
