// pixel_io.h -- scene staging into LDS, pixel store, hand-over records of the sorted schedule
// Part of the single gfx950 translation unit rtiow_hip.hip (included there, in this order; internal linkage).
#pragma once
#include "vecmath.h"

namespace {

// camera.h:167-171, color.h:10-13.  The RNG state is deliberately not written back.
template <class T, class COLD>
__device__ __forceinline__ void store_pixel(const COLD& c, size_t lp, V3<T> acc) {
    acc = scale3((T)c.pixel_samples_scale, acc);
    T* o = c.fb + lp * 3;
    o[0] = acc.x > (T)0 ? Real<T>::sqrt(acc.x) : (T)0;
    o[1] = acc.y > (T)0 ? Real<T>::sqrt(acc.y) : (T)0;
    o[2] = acc.z > (T)0 ? Real<T>::sqrt(acc.z) : (T)0;
}

// The writer's quantisation on the device (main.cu:367, 374-376; interval.h:25-29): level = int(256 * clamp(c, 0.000, 0.999)) per channel,
// the same comparisons, the same multiplication in T and the same truncation as csrc/host/rtiow_host.cpp to_level -- so the drop-in
// executable reads back one byte per channel instead of a T (rtiow_read_levels).  A NaN channel (undefined in the reference:
// int(NaN)) is counted; the caller then takes the T framebuffer and the host writer, which prints what the reference's x86 build prints.
template <class T>
__global__ void __launch_bounds__(256) quantise_kernel(const T* __restrict__ fb, unsigned char* __restrict__ levels, size_t n, unsigned long long* __restrict__ nan_channels) {
    const size_t k0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (k0 >= n) return;
    unsigned word = 0, nans = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        unsigned level = 0;
        if (k0 + u < n) {
            const T c = fb[k0 + u];
            const T lo = (T)0.000, hi = (T)0.999;
            if (!(c == c)) ++nans;
            const T cl = c < lo ? lo : (c > hi ? hi : c);
            level = (unsigned)(int)((T)256 * cl) & 255u;
        }
        word |= level << (8 * u);
    }
    if (k0 + 4 <= n) *reinterpret_cast<unsigned*>(levels + k0) = word;          // k0 is a multiple of 4, the buffer 256-byte aligned
    else for (int u = 0; k0 + u < n; ++u) levels[k0 + u] = (unsigned char)(word >> (8 * u));
    if (nans) atomicAdd(nan_channels, (unsigned long long)nans);
}

template <class T, int SRC>
__device__ __forceinline__ T* stage_scene(const RenderParams<T>& p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds_geom = reinterpret_cast<T*>(smem_raw);
    if (SRC == RTIOW_SCENE_LDS || p.shade_in_lds) {
        // Stage the loop table {cx,cy,cz,r^2} (and the shade records when they fit): coalesced
        // global reads, one pass.
        if (SRC == RTIOW_SCENE_LDS) {
            for (int k = threadIdx.x; k < p.n_padded * 4; k += blockDim.x) lds_geom[k] = p.geom_a[k];
            if (p.use_screen) {
                float* lds_screen = reinterpret_cast<float*>(smem_raw + p.screen_offset);   // fp32 for both precisions
                const float* geom_s = p.screen.geom_s;   // kernel entry: nothing is live yet
                for (int k = threadIdx.x; k < p.n_padded * 4; k += blockDim.x) lds_screen[k] = geom_s[k];
            }
        }
        if (p.shade_in_lds) {
            T* lds_shade = reinterpret_cast<T*>(smem_raw + p.shade_offset);
            const T* tbl = p.screen.shade_tbl;
            for (int k = threadIdx.x; k < p.n * 12; k += blockDim.x) lds_shade[k] = tbl[k];
        }
        if (SRC == RTIOW_SCENE_LDS && p.use_grid) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(smem_raw + p.grid.cells_offset);
            const uint32_t* src = reinterpret_cast<const uint32_t*>(p.grid.blob);
            for (int k = threadIdx.x; k < p.grid.blob_bytes / 4; k += blockDim.x) dst[k] = src[k];
        }
        __syncthreads();
    }
    return lds_geom;
}

__device__ __forceinline__ int global_row(int jl, int strip_rows, int nranks, int rank) {
    return ((jl / strip_rows) * nranks + rank) * strip_rows + (jl % strip_rows);
}

// End of a pixel in one launch: the final phase writes the pixel
// (camera.h:167-171); the prepass of the sorted schedule parks the exact state instead.
// Per-pixel hand-over record, read and written as 16-byte vectors: fp32 48 bytes, fp64 64 bytes.
template <class T> struct MidState;
template <> struct alignas(16) MidState<float>  { uint32_t v[5], d; float acc[3]; uint32_t pad[3]; };
template <> struct alignas(16) MidState<double> { uint32_t v[5], d; uint32_t pad[2]; double acc[3]; uint32_t pad2[2]; };
static_assert(sizeof(MidState<float>) == 48 && sizeof(MidState<double>) == 64, "hand-over record layout");

template <class T>
__device__ __forceinline__ void park_state(unsigned char* base, size_t lp, const PathState<T>& st) {
    MidState<T> m;
    m.v[0] = st.rs.v0; m.v[1] = st.rs.v1; m.v[2] = st.rs.v2; m.v[3] = st.rs.v3; m.v[4] = st.rs.v4; m.d = st.rs.d;
    m.acc[0] = st.acc.x; m.acc[1] = st.acc.y; m.acc[2] = st.acc.z;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4* src = reinterpret_cast<const u4*>(&m);
    u4* dst = reinterpret_cast<u4*>(base + lp * sizeof(MidState<T>));
#pragma unroll
    for (int k = 0; k < (int)(sizeof(MidState<T>) / 16); ++k) dst[k] = src[k];
}
template <class T>
__device__ __forceinline__ void unpark_state(const unsigned char* base, size_t lp, PathState<T>& st) {
    MidState<T> m;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4* src = reinterpret_cast<const u4*>(base + lp * sizeof(MidState<T>));
    u4* dst = reinterpret_cast<u4*>(&m);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(MidState<T>) / 16); ++k) dst[k] = src[k];
    st.rs.v0 = m.v[0]; st.rs.v1 = m.v[1]; st.rs.v2 = m.v[2]; st.rs.v3 = m.v[3]; st.rs.v4 = m.v[4]; st.rs.d = m.d;
    st.acc = {m.acc[0], m.acc[1], m.acc[2]};
}

template <class T, class COLD>
__device__ __forceinline__ void finish_pixel(const COLD& c, size_t lp, const PathState<T>& st, unsigned int cost) {
    if (c.mid_out) {
        park_state<T>(c.mid_out, lp, st);
        c.cost_out[lp] = cost;
    } else {
        store_pixel<T>(c, lp, st.acc);
    }
}

// Clock warm-up (study knob RTIOW_CLOCK_WARMUP_US, profiles/r05/cold_process_study.md): every SIMD busy with dependent FMAs for `ticks` of the
// 100 MHz counter.  A process's first render finds the chip at ~2.24 GHz and its second at ~2.34 (DVFS ramps under load); the reference's own
// set-up ends with a long kernel (curand_init with a subsequence per pixel, main.cu:326-330), ours with a 0.6 ms one.
__global__ void __launch_bounds__(256) clock_warmup_kernel(unsigned long long ticks, float* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float a = 1.0f + 1e-7f * (float)threadIdx.x, b = 0.5f;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
        for (int k = 0; k < 64; ++k) { a = __builtin_fmaf(a, 0.999f, 1e-3f); b = __builtin_fmaf(b, 0.998f, 2e-3f); }
    }
    if (a + b == 12345.678f) sink[0] = a;
}

}  // namespace
