/* rtiow_host.h -- C interface of librtiow_host.so: the host-side (CPU, no GPU needed) half
 * of the drop-in: scene tables, camera set-up, output naming and the P3 writer of
 *   /root/reference/src/GlobalFloatCUDAInOneWeekend/main.cu  (fp32)  and
 *   /root/reference/src/GlobalDoubleCUDAInOneWeekend/main.cu (fp64).
 * The executables global-{float,double}-hip-raytrace are built from these functions plus
 * the device C-ABI in rtiow.h; the Python mirror (raytracingincuda_amd/api.py) binds both.
 * All functions return 0 on success or a negative RTIOW_E_* code.
 */
#ifndef RTIOW_HOST_H
#define RTIOW_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "rtiow.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Number of table slots the reference allocates for a scene id (main.cu:151,198,243):
 * 1 -> 488, 2 -> 40, anything else -> 125. */
int rtiow_host_scene_slots(int scene_id);

/* World creation (main.cu:148-296) with the host compiler's (g++) argument evaluation
 * order made explicit and glibc's unseeded rand() stream restated, so the tables are the
 * ones the reference binary builds.  Arrays in precision T (32|64):
 * center_radius[4*slots], albedo_fuzz[4*slots], refraction_index[slots], type[slots],
 * valid[slots] (0 for grid cells the reference skips at main.cu:168). Returns slot count. */
int rtiow_host_build_scene(int scene_id, int precision, void* center_radius, void* albedo_fuzz,
                           void* refraction_index, int32_t* type, int32_t* valid);

/* camera configuration + camera::initialize (main.cu:100-124, camera.h:33-68).
 * out points to rtiow_camera_f32 or rtiow_camera_f64. */
int rtiow_host_camera(int precision, int width, int height, int samples, int bounces, void* out);

/* Output file name (main.cu:349-358): "<variant>_scene{id}_{W}x{H}_{S}samples_{B}bounces_
 * {T}threadsPerBlockRow.ppm", variant = "global_float" | "global_double". */
int rtiow_host_ppm_filename(int precision, int scene_id, int width, int height, int samples,
                            int bounces, int threads, char* out, size_t cap);

/* P3 writer (main.cu:368-379): "P3\nW H\n255\n" then "r g b\n" per pixel with
 * int(256*clamp(c, 0.000, 0.999)).  rgb is width*height*3 T, row-major.
 * rtiow_host_format_ppm writes into memory (returns the length needed via *len). */
int rtiow_host_write_ppm(const char* path, int precision, int width, int height, const void* rgb);
int rtiow_host_format_ppm(int precision, int width, int height, const void* rgb, char* out, size_t cap, size_t* len);
/* Binary variant (not in the reference): "P6\nW H\n255\n" + one byte per channel, the same
 * int(256*clamp(c, 0.000, 0.999)) levels; ~12x smaller and faster than the text file. */
int rtiow_host_write_ppm_binary(const char* path, int precision, int width, int height, const void* rgb);
/* The same two files from LEVELS (one byte per channel, 0..255: rtiow_read_levels, or rtiow_host_levels below): text P3 (binary = 0; formatted
 * by up to 16 threads, every thread writing its own range of the file) or P6 (binary = 1).  Byte-identical to the writers above for frames
 * without NaN channels. */
int rtiow_host_write_ppm_levels(const char* path, int width, int height, const unsigned char* levels, int binary);
/* main.cu:367, 374-376 per channel on the host: levels[k] = int(256 * clamp(rgb[k], 0.000, 0.999)); returns the number of NaN channels (their byte is 0). */
long long rtiow_host_levels(int precision, int width, int height, const void* rgb, unsigned char* levels);

/* Row sharding used by rtiow_set_shard (rtiow.h): strips of strip_rows rows dealt round-robin.
 * Writes the global row index of every local row of `rank` (rows_out may be NULL) and returns
 * the local row count (negative on bad arguments). */
int rtiow_host_shard_rows(int height, int rank, int nranks, int strip_rows, int32_t* rows_out);

/* Scatter `local_rows` rows rendered by shard (rank, nranks, strip_rows) into the full
 * image (both width*rows*3 T).  Used after the gather in multi-GPU runs. */
int rtiow_host_place_rows(int precision, int width, int height, int rank, int nranks, int strip_rows,
                          const void* local_rgb, void* full_rgb);

#ifdef __cplusplus
}
#endif
#endif /* RTIOW_HOST_H */
