// cost_sort.h -- cost smoothing and counting sort of the sorted schedule
// Part of the single gfx950 translation unit rtiow_hip.hip (included there, in this order; internal linkage).
#pragma once
#include "render_kernels.h"

namespace {

// ---- SCHED_SORTED: counting sort of the pixels by the cost measured in the prepass, heavy first,
// dealt into balanced pools.  Sorted rank r -> slot: ranks are cut into blocks of
// `pools_per_block` pools (the resident waves of one dispatch-age class); inside a block groups of
// `group` consecutive ranks go to consecutive pools, so every pool of a block gets the same mix of
// costs and the blocks run from the heaviest pixels to the lightest.  Ranks inside a cost bin follow
// the image (64 x 64 super-tiles, 8 x 8 tiles), so a group is a handful of neighbouring pixels.
constexpr int COST_BINS = 1024;
__device__ __forceinline__ int cost_bin(unsigned c) { return c < (unsigned)COST_BINS ? (int)c : COST_BINS - 1; }

// What the sort ranks a pixel by: the prepass cost averaged over its (2 hw + 1)^2 neighbourhood (inside its
// own row strip), in quarter segments.  A pixel's own 3 samples predict the cost of its remaining 97 poorly
// (correlation 0.51 on the oracle's segment maps: half a percent of the heaviest pixels were handed out after
// more than half of the frame's work); heavy pixels cluster -- the rims of the glass spheres, the crevices
// between spheres -- and the 75 samples of a 5 x 5 neighbourhood predict it well (0.91; the same pixels then
// start within the first 16 %).  Measured: headline 14.6 -> 13.5 ms, 1280x720 10.0 -> 8.6, half-frame shard
// 10.6 -> 8.5, scene 1 26.0 -> 23.8 (hw = 6).  hw = half-width of the window.  `strip_rows` = the rows that are
// neighbours in the image: a rank's strip in a sharded frame (windows that cross into the rank's next strip, N x
// strip rows away, rank the pixels worse: 1/4 frame 6.4 -> 7.2 ms), the whole frame on one rank (until the end of
// round 2 the window stopped at the default 8-row strips there too: 1280x720 8.3 -> 7.9 ms, headline 13.4 -> 13.3,
// profiles/archive/r02_handout_study/sweep8_smoothing_window.txt; half-widths 5-10 are equal, sweep9).
// One workgroup smooths a 64 x 16 tile from LDS: the tile with its halo, then the horizontal window sums of every
// row it needs, then the vertical sums (26 LDS reads per pixel instead of 169 cached global loads: 61 -> 20 us on
// the full frame).  Integer sums: the same values in any order.
// The histogram of the keys (what cost_hist_kernel counts for an unsmoothed key) rides along: one LDS histogram
// per tile, one global atomic per non-empty bin.
constexpr int SMOOTH_TW = 64, SMOOTH_TH = 16;
__global__ void __launch_bounds__(256) cost_smooth_kernel(const uint32_t* __restrict__ cost, uint32_t* __restrict__ out, int W, int rows, int strip_rows, int hw,
                                                          unsigned* __restrict__ hist) {
    extern __shared__ uint32_t smooth_lds[];
    __shared__ unsigned tile_hist[COST_BINS];
    for (int b = threadIdx.x; b < COST_BINS; b += blockDim.x) tile_hist[b] = 0;
    const int halo_w = SMOOTH_TW + 2 * hw, halo_h = SMOOTH_TH + 2 * hw;
    uint32_t* tile = smooth_lds;                       // [halo_h][halo_w], zero outside the image
    uint32_t* hsum = smooth_lds + halo_w * halo_h;     // [halo_h][SMOOTH_TW]
    const int tiles_x = (W + SMOOTH_TW - 1) / SMOOTH_TW;
    const int tx = (int)blockIdx.x % tiles_x, ty = (int)blockIdx.x / tiles_x;
    const int x_base = tx * SMOOTH_TW - hw, y_base = ty * SMOOTH_TH - hw;
    for (int k = threadIdx.x; k < halo_w * halo_h; k += blockDim.x) {
        const int ly = k / halo_w, lx = k - ly * halo_w;
        const int x = x_base + lx, y = y_base + ly;
        tile[k] = (x >= 0 && x < W && y >= 0 && y < rows) ? cost[y * W + x] : 0u;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < SMOOTH_TW * halo_h; k += blockDim.x) {
        const int ly = k / SMOOTH_TW, lx = k - ly * SMOOTH_TW;
        unsigned sum = 0;
        for (int d = 0; d <= 2 * hw; ++d) sum += tile[ly * halo_w + lx + d];     // columns outside the image hold 0
        hsum[k] = sum;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < SMOOTH_TW * SMOOTH_TH; k += blockDim.x) {
        const int ly = k / SMOOTH_TW, lx = k - ly * SMOOTH_TW;
        const int i = tx * SMOOTH_TW + lx, jl = ty * SMOOTH_TH + ly;
        if (i >= W || jl >= rows) continue;
        const int s0 = (jl / strip_rows) * strip_rows;                               // rows of other strips are not neighbours in the image
        const int j0 = jl - hw > s0 ? jl - hw : s0;
        int j1 = jl + hw < s0 + strip_rows - 1 ? jl + hw : s0 + strip_rows - 1;
        if (j1 > rows - 1) j1 = rows - 1;
        const int i0 = i - hw > 0 ? i - hw : 0, i1 = i + hw < W - 1 ? i + hw : W - 1;
        unsigned sum = 0;
        for (int j = j0; j <= j1; ++j) sum += hsum[(j - y_base) * SMOOTH_TW + lx];
        // mean over the window actually covered, in quarter segments: the bins keep their resolution at the image
        // border and in two-row strips
        const unsigned cells = (unsigned)((j1 - j0 + 1) * (i1 - i0 + 1));
        const unsigned key = (4u * sum + cells / 2) / cells;
        out[jl * W + i] = key;
        atomicAdd(&tile_hist[cost_bin(key)], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < COST_BINS; b += blockDim.x) if (tile_hist[b]) atomicAdd(&hist[b], tile_hist[b]);
}

__global__ void __launch_bounds__(256) cost_hist_kernel(const uint32_t* __restrict__ cost, int npix, unsigned* __restrict__ hist) {
    __shared__ unsigned local[COST_BINS];
    for (int b = threadIdx.x; b < COST_BINS; b += blockDim.x) local[b] = 0;
    __syncthreads();
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < npix; k += gridDim.x * blockDim.x) {
        atomicAdd(&local[cost_bin(cost[k])], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < COST_BINS; b += blockDim.x) if (local[b]) atomicAdd(&hist[b], local[b]);
}

// start[b] = number of pixels with a HIGHER bin (heavy first); also zeroes the fill counters.
__global__ void __launch_bounds__(COST_BINS) cost_scan_kernel(const unsigned* __restrict__ hist, unsigned* __restrict__ start, unsigned* __restrict__ fill) {
    __shared__ unsigned tmp[COST_BINS];
    const int b = threadIdx.x;
    tmp[b] = hist[COST_BINS - 1 - b];            // reversed: index 0 = heaviest bin
    __syncthreads();
    for (int off = 1; off < COST_BINS; off <<= 1) {
        const unsigned v = b >= off ? tmp[b - off] : 0;
        __syncthreads();
        tmp[b] += v;
        __syncthreads();
    }
    start[COST_BINS - 1 - b] = tmp[b] - hist[COST_BINS - 1 - b];   // exclusive
    fill[b] = 0;
}

// Each 1024-thread block ranks 4096 pixels: a block-private histogram in LDS, ONE global atomic per
// non-empty bin to reserve the block's range of ranks, then LDS atomics for the rank inside it
// (2 M contended global atomics on ~20 hot bins took 17.8 ms; this takes microseconds).
constexpr int SCATTER_PER_THREAD = 4;
__global__ void __launch_bounds__(1024) cost_scatter_kernel(const uint32_t* __restrict__ cost, int W, int rows, const unsigned* __restrict__ start,
                                                            unsigned* __restrict__ fill, int* __restrict__ order, int pools_per_block, int total_pools, int group,
                                                            int solo_slots, int* __restrict__ slot_of) {
    __shared__ unsigned local[COST_BINS];        // block histogram, then the running rank inside the reserved range
    __shared__ unsigned base[COST_BINS];
    for (int b = threadIdx.x; b < COST_BINS; b += blockDim.x) local[b] = 0;
    __syncthreads();
    // A block ranks one 64 x 64-pixel super-tile, each wave an 8 x 8 tile of it, so that pixels which
    // are neighbours in the image and equal in cost get adjacent ranks (see `group` below).
    const int st_x = (W + 63) >> 6;
    const int sx = (int)blockIdx.x % st_x, sy = (int)blockIdx.x / st_x;
    int bins[SCATTER_PER_THREAD], pix[SCATTER_PER_THREAD], packed_px[SCATTER_PER_THREAD];   // packed: what the render kernel's hand-out reads, (row << 16 | column)
#pragma unroll
    for (int u = 0; u < SCATTER_PER_THREAD; ++u) {
        const int idx = u * (int)blockDim.x + (int)threadIdx.x, tile = idx >> 6, within = idx & 63;
        const int px = sx * 64 + (tile & 7) * 8 + (within & 7), py = sy * 64 + (tile >> 3) * 8 + (within >> 3);
        const int k = (px < W && py < rows) ? py * W + px : -1;
        pix[u] = k;
        packed_px[u] = (py << 16) | px;
        bins[u] = -1;
        if (k >= 0) {
            bins[u] = cost_bin(cost[k]);
            atomicAdd(&local[bins[u]], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < COST_BINS; b += blockDim.x) {
        const unsigned n = local[b];
        base[b] = n ? start[b] + atomicAdd(&fill[b], n) : 0;
        local[b] = 0;
    }
    __syncthreads();
    const int per_block = pools_per_block * POOL;
#pragma unroll
    for (int u = 0; u < SCATTER_PER_THREAD; ++u) {
        if (bins[u] < 0) continue;
        const int k = pix[u];
        int r = (int)(base[bins[u]] + atomicAdd(&local[bins[u]], 1u));   // sorted rank (order inside a bin is immaterial)
        const int packed = packed_px[u];
        if (r < solo_slots) { order[r] = packed; if (slot_of) slot_of[k] = r; continue; }   // the heaviest pixels: slot = rank, handed to the solo waves
        r -= solo_slots;
        const int blk = r / per_block, q = r - blk * per_block;
        const int pools_here = (blk + 1) * pools_per_block <= total_pools ? pools_per_block : total_pools - blk * pools_per_block;
        const int g = q / group, j = q - g * group;                 // groups of `group` consecutive ranks stay together
        const int pool = blk * pools_per_block + g % pools_here;
        const int lane_slot = (g / pools_here) * group + j;
        order[solo_slots + pool * POOL + lane_slot] = packed;
        if (slot_of) slot_of[k] = solo_slots + pool * POOL + lane_slot;       // the inverse, for place_pixels_kernel
    }
}

// Staging buffer in slot order -> image: one thread per pixel in IMAGE order reads its slot (coalesced), gathers the
// 3 T of its pixel from the staging buffer (25 MB at 1080p: L2 / Infinity-Cache resident, written microseconds ago)
// and writes the image in whole lines.  ~20 us at 1920 x 1080.
template <class T>
__global__ void __launch_bounds__(256) place_pixels_kernel(const T* __restrict__ staged, const int* __restrict__ slot_of, T* __restrict__ fb, int npix) {
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= npix) return;
    const T* src = staged + 3 * (size_t)slot_of[k];
    T* dst = fb + 3 * (size_t)k;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

}  // namespace
