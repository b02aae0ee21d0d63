/* rtiow.h -- C-ABI of librtiow_hip.so: the MI355X (gfx950) `render` hot path of the
 * RayTracingInOneWeekend tracer, as a drop-in for the launch sequence in the reference's
 *   src/GlobalFloatCUDAInOneWeekend/main.cu  (fp32)  and
 *   src/GlobalDoubleCUDAInOneWeekend/main.cu (fp64).
 *
 * The reference has no FFI: it launches its kernels inline.  Each entry point below replaces
 * one phase of that inline sequence (cited as main.cu:LINE, relative to
 * /root/reference/src/GlobalFloatCUDAInOneWeekend/ unless noted) so that a host `main` keeps
 * the reference's ordering and timing semantics.  INTEGRATION.md shows the host-side binding.
 *
 * Conventions: plain C, no C++ types, no exceptions.  Every function returns an int:
 * 0 on success, otherwise the hipError_t value of the failing runtime call (or a negative
 * RTIOW_E_* code for argument errors); rtiow_last_error_string() gives the text that the
 * reference's CUDA_SAFE_CALL (main.cu:14-21) would have printed.  The caller owns host
 * memory; the library owns device memory behind the opaque handle.  A handle is not
 * thread-safe.  One handle == one GPU; multi-GPU jobs use either one handle per rank (process;
 * raytracingincuda_amd/distributed.py gathers with torch.distributed) or one rtiow_group (below)
 * in a single process.
 */
#ifndef RTIOW_H
#define RTIOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTIOW_ABI_VERSION 6

#define RTIOW_E_BADARG   (-1)
#define RTIOW_E_STATE    (-2)   /* call order violated (e.g. render before set_scene) */
#define RTIOW_E_NOMEM    (-3)

/* MaterialType (material.h:11-15). */
#define RTIOW_LAMBERTIAN 0
#define RTIOW_METAL      1
#define RTIOW_DIELECTRIC 2

/* Scene-table source selected for the sphere loop (the AMD analogue of the reference's
 * global / constant / texture variants, README.md:7-12). */
#define RTIOW_SCENE_LDS       0 /* sphere list staged into LDS per workgroup, with the packed-
                                 * fp32 8-operation conservative screen in front of the exact test
                                 * (both precisions)                                             */
#define RTIOW_SCENE_SCALAR    1 /* wave-uniform scalar loads through the scalar cache, exact loop */
#define RTIOW_SCENE_LDS_EXACT 2 /* LDS, the reference's 12-operation test on every sphere        */
#define RTIOW_SCENE_GRID      3 /* default: LDS tables plus a uniform grid over the small spheres -- a
                                 * lane walks the cells its ray crosses and tests only their spheres,
                                 * exactly; big spheres are tested by every ray.  Same image bit for
                                 * bit.  Scenes the grid does not suit run as RTIOW_SCENE_LDS.     */

/* Pixel scheduling (same image either way):
 * STATIC     = the reference's launch geometry: grid of T x T blocks, one lane per pixel
 *              (main.cu:137-139, camera.h:131-134);
 * PERSISTENT = resident waves pull 64-pixel pools from a global counter and hand a new pixel
 *              to every lane the moment it finishes one (lanes are not pixels here, so
 *              --threads has no effect: workgroups are always four waves). */
#define RTIOW_SCHED_STATIC     0
#define RTIOW_SCHED_PERSISTENT 1
/* SORTED = PERSISTENT in two launches: a prepass renders the first 3 samples of every pixel and
 * records their path-segment counts; the pixels are then ranked heavy-first into balanced pools
 * of neighbouring pixels and the main launch renders the remaining samples in that order (RNG
 * state and colour sum carried exactly, so the image is unchanged).  Removes the drain tail of
 * late heavy pixels.  Default.  The hand-out order packs a pixel as (local row << 16 | column): a frame wider than 65535
 * columns or a shard taller than 32767 rows (or smaller than 4096 pixels) is rendered like RTIOW_SCHED_PERSISTENT, in one
 * launch in tile order (same image; rtiow_stats.phases then reads 1). */
#define RTIOW_SCHED_SORTED     2

typedef struct rtiow_handle_s* rtiow_handle;

/* The fields of `struct camera` that `render` reads (camera.h:10-30), produced by
 * camera::initialize (camera.h:33-68).  Scalars are in the handle's precision: pass
 * rtiow_camera_f32 to a 32-bit handle and rtiow_camera_f64 to a 64-bit one. */
typedef struct {
    int32_t img_width, img_height, samples_per_pixel, max_depth;
    float   pixel_samples_scale;
    float   center[3], pixel00_loc[3], pixel_delta_u[3], pixel_delta_v[3];
    float   defocus_angle;
    float   defocus_disk_u[3], defocus_disk_v[3];
} rtiow_camera_f32;

typedef struct {
    int32_t img_width, img_height, samples_per_pixel, max_depth;
    double  pixel_samples_scale;
    double  center[3], pixel00_loc[3], pixel_delta_u[3], pixel_delta_v[3];
    double  defocus_angle;
    double  defocus_disk_u[3], defocus_disk_v[3];
} rtiow_camera_f64;

typedef struct {
    double   rng_init_ms;        /* last rtiow_init_rng kernel time (HIP events)           */
    double   render_ms;          /* last rtiow_render kernel time (HIP events)             */
    uint64_t primary_rays;       /* local_rows * width * samples of the last render        */
    int32_t  local_rows;         /* rows of the image this handle renders                  */
    int32_t  num_spheres;        /* spheres uploaded (invalid slots already dropped)       */
    int32_t  block_x, block_y;   /* thread-block shape used by the last render             */
    int32_t  vgprs, sgprs;       /* register use of the render kernel variant (0: unknown) */
    int32_t  lds_bytes;          /* dynamic+static LDS per workgroup of the last render    */
    int32_t  scene_source;       /* RTIOW_SCENE_*                                          */
    int32_t  schedule;           /* RTIOW_SCHED_*                                          */
    int32_t  grid_blocks;        /* workgroups launched by the last render                 */
    int32_t  phases;             /* 2 when RTIOW_SCHED_SORTED split the render, else 1     */
    int32_t  prepass_samples;    /* samples per pixel rendered by the prepass launch (0: none) */
    double   prepass_ms;         /* HIP-event time of the prepass launch of the last timed render */
    double   main_ms;            /* HIP-event time of the main launch (render_persistent_kernel /
                                  * render_kernel): the dominant kernel of the roofline      */
    uint64_t segments_prepass;   /* last rtiow_count_segments: hit_world calls per launch   */
    uint64_t segments_main;
    uint64_t max_chain_prepass;  /* last rtiow_count_segments: the longest per-pixel chain of      */
    uint64_t max_chain_main;     /* segments in each launch (a pixel's samples are sequential: one
                                  * RNG stream, so no schedule finishes before its longest chain)  */
    /* RTIOW_SCENE_GRID, last render (all 0 when the scene runs without a grid): */
    int32_t  grid_nx, grid_nz;   /* cells along x and z                                           */
    int32_t  grid_registered;    /* spheres binned into cells (each in up to 2 x 2 of them)       */
    int32_t  grid_direct;        /* spheres every ray tests exactly (too big for a cell / overflow) */
    double   grid_cell;          /* cell width                                                    */
    /* RTIOW_SCHED_SORTED on a partly filled GPU (small frame, shard): waves that held only the top-ranked
     * pixels, and how many each (0: the plain kernel ran) */
    int32_t  solo_waves, solo_lanes;
    /* Host time of the per-scene pre-processing the render kernels rely on -- screening table + uniform-grid plan
     * and blob (build_screen_table, build_grid_tables) -- spent once per rtiow_set_scene, at the first render after
     * it, BEFORE the start event of that render: like the reference's scene set-up (main.cu:148-321) it is outside
     * render_only and inside end_to_end.  0 when the last scene needed no tables (scalar / exact sources). */
    double   scene_prepare_ms;
    /* RTIOW_SCHED_SORTED: 1 when the main launch stored finished pixels in slot order into a staging buffer and
     * place_pixels_kernel wrote the image in whole lines (coalesced framebuffer writes); place_ms = HIP-event time of
     * that kernel in the last timed render (inside render_ms, outside main_ms). */
    int32_t  staged_stores;
    int32_t  num_cus;            /* compute units of the handle's device (hipDeviceProp_t::multiProcessorCount)              */
    double   place_ms;
    int32_t  clock_mhz;          /* its nominal shader clock (hipDeviceProp_t::clockRate): what an issue-slot figure is rated against */
    int32_t  reserved0;
    /* ABI 6.  EFFECTIVE shader clock of the last timed render's launches (persistent schedules; 0: none taken): one wave of each launch --
     * the first dispatched, resident until the hand-out runs dry -- stamps s_memtime and s_memrealtime when it starts and ends, the clock is
     * d(s_memtime) / d(s_memrealtime) x 100 MHz over that wave's life.  A cold process whose render takes 16 ms instead of 11.6 shows here
     * whether the chip was still ramping its clock (profiles/r05/cold_process_study.md). */
    double   main_clock_mhz, prepass_clock_mhz;
    double   main_wave0_ms;      /* that wave's life in the main launch (s_memrealtime), for reading main_clock_mhz against main_ms */
} rtiow_stats;

/* ---- lifetime -------------------------------------------------------------------------
 * rtiow_create replaces cudaSetDevice(0) + event creation (main.cu:81-92).
 * precision is 32 (GlobalFloat) or 64 (GlobalDouble). */
int rtiow_abi_version(void);
/* SHA-256 (hex) of the sources and compiler flags this library was built from (raytracingincuda_amd/build.py
 * passes it in; "unknown" for a hand build).  Profiles record it, and bench.py reports counter-derived figures
 * only from records taken with the very build that is loaded. */
const char* rtiow_build_id(void);
int rtiow_create(int device, int precision, rtiow_handle* out);
int rtiow_destroy(rtiow_handle h);                                   /* main.cu:384-388 */
const char* rtiow_last_error_string(rtiow_handle h);                 /* main.cu:14-21   */

/* Run on an existing hipStream_t (e.g. torch's current stream) instead of the handle's own. */
int rtiow_set_stream(rtiow_handle h, void* hip_stream);

/* ---- scene: replaces cudaMalloc/cudaMemcpy of materials+spheres+world and the two
 * pointer fix-up kernels (main.cu:301-321).  Arrays are host arrays in the handle's
 * precision T:  center_radius[4n] = {cx,cy,cz,r}, albedo_fuzz[4n] = {r,g,b,fuzz},
 * refraction_index[n]; type[n] = RTIOW_*; valid[n] (may be NULL = all valid): slots the
 * reference leaves default-constructed (skipped grid cells, main.cu:168) are dropped. */
int rtiow_set_scene(rtiow_handle h, int n, const void* center_radius, const void* albedo_fuzz,
                    const void* refraction_index, const int32_t* type, const int32_t* valid);

/* ---- camera: replaces passing `cam` by value to render (main.cu:335). */
int rtiow_set_camera(rtiow_handle h, const void* camera /* rtiow_camera_f32 | _f64 */);

/* ---- multi-GPU row sharding (new; the reference is single-GPU, main.cu:81).
 * The image is cut into strips of strip_rows rows dealt round-robin: this handle renders
 * the strips s with s % nranks == rank.  Default (0,1,8) = the whole image.  RNG streams
 * are keyed by the GLOBAL pixel index, so the assembled image does not depend on nranks. */
int rtiow_set_shard(rtiow_handle h, int rank, int nranks, int strip_rows);
int rtiow_local_rows(rtiow_handle h, int* rows);
/* Global row index of each local row (rows_out has rtiow_local_rows entries). */
int rtiow_local_row_map(rtiow_handle h, int32_t* rows_out);

/* ---- RNG: replaces cudaMalloc(rand_states) + init_rng<<<>>> (main.cu:326-330,
 * rtweekend.h:43-50): XORWOW, curand_init(seed, global_pixel_index, 0). */
int rtiow_init_rng(rtiow_handle h, uint64_t seed);

/* ---- render: replaces render<<<dimGrid,dimBlock>>> + sync (main.cu:334-341).
 * threads_per_block_row is the reference's --threads (block = T x T pixels, main.cu:137-139;
 * the grid is ceil-divided, unlike main.cu:137-138): it shapes the launch of RTIOW_SCHED_STATIC and
 * is accepted and ignored by the dynamic schedules.  0 selects the library's own tiling.
 * kernel_ms (may be NULL) receives the HIP-event time around the kernel only; passing NULL
 * makes the call asynchronous on the handle's stream. */
int rtiow_render(rtiow_handle h, int threads_per_block_row, float* kernel_ms);

/* The two halves of a timed rtiow_render, for callers that drive several GPUs from one thread
 * (rtiow_group_*, below): rtiow_render_async enqueues start event, launches and stop event on the
 * handle's stream and returns; rtiow_render_wait blocks on the stop event and returns the
 * HIP-event time of the kernels alone (the same figure rtiow_render reports). */
int rtiow_render_async(rtiow_handle h, int threads_per_block_row);
int rtiow_render_wait(rtiow_handle h, float* kernel_ms);

/* Same render with a path-segment counter (hit_world calls, hittable.h:80) added: untimed,
 * used by bench.py for the algorithmic-flop figure and by tests against the oracle's count.
 * The image it leaves in the framebuffer is identical to rtiow_render's. */
int rtiow_count_segments(rtiow_handle h, int threads_per_block_row, uint64_t* segments);

/* Framebuffer: `vec3 pixel_buffer[]` (main.cu:133-134), local_rows x width x 3 T, row-major.
 * By default device memory owned by the library; rtiow_bind_framebuffer lets the caller
 * supply device memory (e.g. a torch tensor that torch.distributed will gather). */
int rtiow_bind_framebuffer(rtiow_handle h, void* device_ptr, size_t bytes);
int rtiow_framebuffer_device_ptr(rtiow_handle h, void** device_ptr, size_t* bytes);
/* D2H copy of the local rows (replaces the managed-memory read at main.cu:373). */
int rtiow_read_framebuffer(rtiow_handle h, void* host_rgb, size_t bytes);
/* The writer's quantisation done on the device (main.cu:367, 374-376: int(256 * clamp(c, 0.000, 0.999)) per channel, in T, truncated) and
 * the local rows read back as one byte per channel: local_rows x width x 3 bytes, a quarter (fp32) or an eighth (fp64) of
 * rtiow_read_framebuffer's bytes.  *nan_channels = channels holding a NaN (their level is int(NaN), undefined in the reference; its x86 build
 * prints -2147483648): when it is not 0 the bytes of those channels mean nothing -- read the T framebuffer and use the T writer
 * (rtiow_host_write_ppm), which prints what the reference prints.  The host writers for levels: rtiow_host_write_ppm_levels. */
int rtiow_read_levels(rtiow_handle h, unsigned char* host_levels, size_t bytes, uint64_t* nan_channels);

/* ---- knobs / introspection */
int rtiow_set_scene_source(rtiow_handle h, int scene_source /* RTIOW_SCENE_* */);
/* waves_per_simd: 0 = as many resident waves as fit; 1..8 caps them (PERSISTENT only). */
int rtiow_set_schedule(rtiow_handle h, int schedule /* RTIOW_SCHED_* */, int waves_per_simd);
int rtiow_get_stats(rtiow_handle h, rtiow_stats* out);
int rtiow_synchronize(rtiow_handle h);
int rtiow_stream(rtiow_handle h, void** hip_stream);   /* the hipStream_t the handle launches on */
int rtiow_device(rtiow_handle h, int* device);

/* Test hooks (device RNG states, per-pixel costs, per-wave timeline, hit_world on caller-supplied rays, ...) are NOT part of this
 * library's ABI: include/rtiow_debug.h declares them and only the test build (librtiow_hip_debug.so, -DRTIOW_DEBUG_API) exports them. */

/* ======================================================================================
 * Multi-GPU inside one process (new work: the reference is single-GPU, main.cu:81).
 *
 * A group is ngpus handles -- one per device, each with its own stream -- that render the
 * interleaved row strips of ONE image (rtiow_set_shard(rank, ngpus, strip_rows)) and exchange
 * them exactly once, after the render: every device sends its strips to device 0 over RCCL
 * (ncclCommInitAll + one ncclGroupStart/End of ncclSend/ncclRecv pairs over xGMI; librccl.so is
 * dlopen'ed on first use) or, when RCCL is unavailable, with hipMemcpyPeerAsync; device 0
 * de-interleaves the strips into the full image.  The assembled image equals the single-GPU
 * image bit for bit (RNG streams are keyed by the global pixel index).
 *
 * The calls mirror the single-handle ones phase by phase, so a host main() keeps the reference's
 * ordering (main.cu:81-400): create, set_camera, set_scene, init_rng, render, read_framebuffer.
 * rtiow_group_render's kernel_ms is the reference's render_only figure for the slowest device;
 * the exchange is NOT inside it (it belongs to the read-back, like the managed-memory read at
 * main.cu:373) and is reported separately in rtiow_group_stats.
 * ====================================================================================== */
#define RTIOW_GATHER_AUTO 0   /* RCCL if it loads and initialises, else peer copies */
#define RTIOW_GATHER_RCCL 1   /* RCCL or fail                                        */
#define RTIOW_GATHER_PEER 2   /* hipMemcpyPeerAsync per device                       */
#define RTIOW_GATHER_HOST 3   /* every strip through a host bounce buffer (blocking copies): the last resort of RTIOW_GATHER_AUTO when a
                               * transport fails AT GATHER TIME (first ncclGroupEnd / send / recv, first peer copy): the group then falls back
                               * RCCL -> peer copies -> host inside the same call, drains the devices in between, and says so in
                               * rtiow_group_transport_note / rtiow_group_stats.gather_mode.  May also be requested outright. */
#define RTIOW_GROUP_MAX_STATS 16

typedef struct rtiow_group_s* rtiow_group;

typedef struct {
    int32_t  ngpus, strip_rows;
    int32_t  gather_mode;                       /* transport of the last gather: RTIOW_GATHER_RCCL | _PEER (0: none yet) */
    int32_t  rccl_version;                      /* ncclGetVersion() when RCCL is in use, else 0 */
    double   kernel_ms[RTIOW_GROUP_MAX_STATS];  /* per device: HIP-event time of its own kernels, last render */
    double   kernel_ms_max;                     /* = what rtiow_group_render returned */
    double   gather_ms;                         /* HIP events on device 0 around exchange + de-interleave, opened when the last render finished */
    uint64_t gather_bytes;                      /* bytes that arrived on device 0 */
    double   create_ms;                         /* wall time of rtiow_group_create: contexts, streams and -- RCCL -- ncclCommInitAll
                                                 * (seconds: topology discovery); like the reference's cudaSetDevice / event creation
                                                 * (main.cu:81-92) it lies BEFORE the end-to-end timer of the executables */
    int32_t  peer_links;                        /* peer mode: ranks whose device got direct access to device 0 enabled */
    int32_t  reserved;
} rtiow_group_stats;

/* devices == NULL: devices 0..ngpus-1.  A device may be listed more than once (ranks then share
 * it and the exchange uses copies: RCCL needs distinct devices) -- used to test the N-rank logic
 * on a one-GPU box. */
/* The transport is chosen and the RCCL communicator created here (seconds: keep it out of timed
 * regions; rtiow_group_stats.create_ms); RTIOW_GATHER_RCCL fails here (RTIOW_E_STATE) when RCCL cannot serve the
 * group.  RCCL prints a version banner on stdout when a process creates its first communicator: a caller whose
 * stdout is data (the executables' CSV fragment) points fd 1 elsewhere around this call -- the library does not. */
int rtiow_group_create(int ngpus, const int* devices, int precision, int strip_rows, int gather, rtiow_group* out);
const char* rtiow_group_create_error(void);   /* text for the calling thread's last failed rtiow_group_create */
int rtiow_group_destroy(rtiow_group g);
const char* rtiow_group_last_error_string(rtiow_group g);
int rtiow_group_size(rtiow_group g);
int rtiow_group_member(rtiow_group g, int rank, rtiow_handle* out);   /* borrowed: knobs, per-device stats */
int rtiow_group_set_scene(rtiow_group g, int n, const void* center_radius, const void* albedo_fuzz,
                          const void* refraction_index, const int32_t* type, const int32_t* valid);
int rtiow_group_set_camera(rtiow_group g, const void* camera);
int rtiow_group_set_scene_source(rtiow_group g, int scene_source);
int rtiow_group_set_schedule(rtiow_group g, int schedule, int waves_per_simd);
int rtiow_group_init_rng(rtiow_group g, uint64_t seed);
int rtiow_group_render(rtiow_group g, int threads_per_block_row, float* kernel_ms);
/* The exchange alone (device 0 then holds the full image, rtiow_group_framebuffer_device_ptr). */
int rtiow_group_gather(rtiow_group g);
int rtiow_group_framebuffer_device_ptr(rtiow_group g, void** device_ptr, size_t* bytes);
/* gather + D2H of the full width*height*3 T image. */
int rtiow_group_read_framebuffer(rtiow_group g, void* host_rgb, size_t bytes);
int rtiow_group_get_stats(rtiow_group g, rtiow_group_stats* out);
/* Why RTIOW_GATHER_AUTO fell back to peer copies ("" if it did not). */
const char* rtiow_group_transport_note(rtiow_group g);
#ifdef __cplusplus
}
#endif
#endif /* RTIOW_H */
