// handle.h -- rtiow_handle_s and the small helpers every host file uses (error text, scratch, shard rows, operand-range checks, make_params)
// Host side of librtiow_hip.so; part of the single translation unit rtiow_hip.hip (internal linkage).
#pragma once
#include "../device/params.h"

struct rtiow_handle_s {
    int device = 0;
    int precision = 32;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_a = nullptr, ev_b = nullptr, ev_c = nullptr;   // ev_a: prepass done, ev_b: main launch starts, ev_c: main launch done (place_pixels_kernel follows)
    bool time_phases = false;
    bool render_pending = false;                  // rtiow_render_async recorded its stop event, rtiow_render_wait has not read it yet
    std::string err;

    // scene
    int n = 0, n_padded = 0;
    void *geom_a = nullptr, *shade_tbl = nullptr;
    void* geom_s = nullptr;                       // screening table (built lazily at the first render of a scene)
    std::vector<double> host_cr;                  // compact {cx,cy,cz,r} kept for building it
    bool screen_dirty = true;
    double ctr[3] = {0, 0, 0}, omax2 = 0;
    // uniform grid over the small spheres (RTIOW_SCENE_GRID; built with the screening table)
    void* grid_blob = nullptr;
    GridParams grid{};                            // offsets are relative to the blob until launch_render places it in LDS
    int grid_cells_bytes = 0, grid_aos_bytes = 0, grid_direct_bytes = 0, grid_ids_bytes = 0;
    int grid_direct = 0, grid_registered = 0;
    // camera
    bool have_camera = false;
    rtiow_camera_f32 cam32{};
    rtiow_camera_f64 cam64{};
    // shard
    int rank = 0, nranks = 1, strip_rows = 8;
    int local_rows = 0;
    // rng
    uint32_t* rng = nullptr;
    size_t rng_pixels = 0;
    bool rng_ready = false;
    uint32_t* rng_low_table = nullptr;            // J^lo * s0 for lo < 2^XW_LOW_BITS (xw_low_table_kernel), per rtiow_init_rng
    uint32_t* jump = nullptr;
    int jump_count = 0;                           // matrices of `jump` that are filled: enough for the bits of the largest pixel index so far
    // framebuffer
    void* fb = nullptr;
    size_t fb_bytes = 0;
    bool fb_external = false;
    // knobs / stats
    int scene_source = RTIOW_SCENE_GRID;
    int schedule = RTIOW_SCHED_SORTED;
    unsigned char* mid = nullptr; size_t mid_bytes = 0;          // SCHED_SORTED: MidState records parked between the launches
    uint32_t* cost = nullptr; size_t cost_bytes = 0;
    unsigned char* levels = nullptr; size_t levels_bytes = 0;     // rtiow_read_levels: one byte per channel + an 8-byte NaN counter behind them
    uint32_t* cost_rank = nullptr; size_t cost_rank_bytes = 0;    // the smoothed cost the sort ranks by
    int* order = nullptr; size_t order_bytes = 0;
    int* slot_of = nullptr; size_t slot_of_bytes = 0;            // SCHED_SORTED: pixel -> slot (the inverse of `order`)
    unsigned char* staged = nullptr; size_t staged_bytes = 0;             // SCHED_SORTED: finished pixels in slot order (place_pixels_kernel writes the image)
    unsigned* sort_scratch = nullptr; size_t sort_scratch_bytes = 0;
    int waves_per_simd = 0;
    int num_cus = 256;
    int last_count_blocks = 0, last_count_waves_per_block = 0;
    size_t timeline_cap_waves = 0;            // waves the debug timeline buffer holds
    unsigned int* work_counter = nullptr;
    int warmup_us = 0;                               // RTIOW_CLOCK_WARMUP_US: busy kernel in front of the handle's FIRST timed render (before its start event), see clock_warmup_kernel
    bool warmed = false;
    unsigned long long* clock_stamps = nullptr;      // pinned + mapped host memory, 8 words: {memtime, realtime} x {start, end} of the prepass [0..3] and the main launch [4..7]
    unsigned long long* clock_stamps_dev = nullptr;  // its device address
    unsigned long long* timeline = nullptr;   // debug: set only during rtiow_debug_timeline
    uint32_t* pixel_times = nullptr;          // debug: set only during rtiow_debug_pixel_times
    int probe_n = 0; const void* probe_rays = nullptr; void* probe_t = nullptr; int* probe_idx = nullptr;   // debug: set only during rtiow_debug_hit_world
    rtiow_stats stats{};
};

namespace {

size_t elem_size(const rtiow_handle_s* h) { return h->precision == 64 ? 8 : 4; }

int fail(rtiow_handle_s* h, hipError_t e, const char* file, int line) {
    char buf[512];
    // same text the reference's CUDA_SAFE_CALL prints (main.cu:16-17)
    std::snprintf(buf, sizeof buf, "HIP_SAFE_CALL: %s %s %d", hipGetErrorString(e), file, line);
    if (h) h->err = buf;
    return (int)e;
}
int fail_arg(rtiow_handle_s* h, int code, const char* msg) { if (h) h->err = msg; return code; }

#define HIP_TRY(h, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail((h), e_, __FILE__, __LINE__); } while (0)

// Device memory of one call: released on every return path (HIP_TRY returns early).
struct DeviceScratch {
    void* ptr = nullptr;
    hipError_t alloc(size_t bytes) { return hipMalloc(&ptr, bytes); }
    ~DeviceScratch() { if (ptr) (void)hipFree(ptr); }
    DeviceScratch() = default;
    DeviceScratch(const DeviceScratch&) = delete;
    DeviceScratch& operator=(const DeviceScratch&) = delete;
};

int compute_local_rows(int H, int rank, int nranks, int strip_rows) {
    int rows = 0;
    const int nstrips = (H + strip_rows - 1) / strip_rows;
    for (int s = rank; s < nstrips; s += nranks) {
        const int r0 = s * strip_rows;
        rows += (r0 + strip_rows <= H) ? strip_rows : (H - r0);
    }
    return rows;
}

int img_w(const rtiow_handle_s* h) { return h->precision == 64 ? h->cam64.img_width : h->cam32.img_width; }
int img_h(const rtiow_handle_s* h) { return h->precision == 64 ? h->cam64.img_height : h->cam32.img_height; }

int ensure_framebuffer(rtiow_handle_s* h) {
    const size_t need = (size_t)h->local_rows * img_w(h) * 3 * elem_size(h);
    if (h->fb_external) {
        if (h->fb_bytes < need) return fail_arg(h, RTIOW_E_BADARG, "bound framebuffer too small");
        return 0;
    }
    if (h->fb && h->fb_bytes >= need) return 0;
    if (h->fb) { HIP_TRY(h, hipFree(h->fb)); h->fb = nullptr; h->fb_bytes = 0; }
    if (need == 0) return 0;
    HIP_TRY(h, hipMalloc(&h->fb, need));
    h->fb_bytes = need;
    return 0;
}

// Can gen_primary take 1/sqrt(|D|^2) without range handling (inv_sqrt_accepted)?  D = pixel sample - lens point:
// the samples lie in the pixel plane (pixel00 + fi du + fj dv, fi in [-0.5, W - 0.5]), the lens points on the
// defocus disk around the centre (|px|, |py| <= 1).  |D| is at most the sum of the extents and at least the
// distance of the lens from the pixel plane; both with room for the fp32 rounding of coordinates up to M.
template <class CAM>
int primary_rays_in_range(const CAM& c) {
    auto v = [](const auto* a) { return std::array<double, 3>{(double)a[0], (double)a[1], (double)a[2]}; };
    auto dot = [](const std::array<double, 3>& a, const std::array<double, 3>& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    auto len = [&](const std::array<double, 3>& a) { return std::sqrt(dot(a, a)); };
    const auto ctr = v(c.center), p00 = v(c.pixel00_loc), du = v(c.pixel_delta_u), dv = v(c.pixel_delta_v);
    std::array<double, 3> ddu = v(c.defocus_disk_u), ddv = v(c.defocus_disk_v);
    if (c.defocus_angle <= 0) ddu = ddv = {0, 0, 0};
    const std::array<double, 3> rel = {p00[0] - ctr[0], p00[1] - ctr[1], p00[2] - ctr[2]};
    std::array<double, 3> n = {du[1] * dv[2] - du[2] * dv[1], du[2] * dv[0] - du[0] * dv[2], du[0] * dv[1] - du[1] * dv[0]};
    const double nl = len(n);
    if (!(nl > 0) || !std::isfinite(nl)) return 0;
    n = {n[0] / nl, n[1] / nl, n[2] / nl};
    const double W = c.img_width + 1.0, H = c.img_height + 1.0;
    const double dmax = len(rel) + W * len(du) + H * len(dv) + len(ddu) + len(ddv);
    const double dmin = std::fabs(dot(rel, n)) - std::fabs(dot(ddu, n)) - std::fabs(dot(ddv, n));
    const double M = len(ctr) + len(p00) + W * len(du) + H * len(dv) + len(ddu) + len(ddv);   // largest coordinate in play
    const double slack = M * 0x1p-18;                                                          // >> the fp32 rounding of ps, org and D
    return std::isfinite(dmax) && dmax + slack < 0x1p30 && dmin - slack > 0x1p-30;
}

// FastDiv (above ieee_roots): every sphere (centre +- radius) and the lens within 2^18 of the origin.
template <class CAM>
int scene_in_range(const rtiow_handle_s* h, const CAM& c) {
    double reach = 0;
    for (size_t i = 0; i + 3 < h->host_cr.size(); i += 4)
        for (int k = 0; k < 3; ++k) reach = std::fmax(reach, std::fabs(h->host_cr[i + k]) + std::fabs(h->host_cr[i + 3]));
    for (int k = 0; k < 3; ++k)
        reach = std::fmax(reach, std::fabs((double)c.center[k]) + std::fabs((double)c.defocus_disk_u[k]) + std::fabs((double)c.defocus_disk_v[k]));
    return !h->host_cr.empty() && std::isfinite(reach) && reach < 0x1p18;
}

template <class T, class CAM>
RenderParams<T> make_params(const rtiow_handle_s* h, const CAM& c) {
    RenderParams<T> p;
    p.range_flags = primary_rays_in_range(c) | (scene_in_range(h, c) << 1);
    p.cold.W = c.img_width; p.cold.H = c.img_height; p.cold.S = c.samples_per_pixel; p.B = c.max_depth;
    p.cold.pixel_samples_scale = c.pixel_samples_scale;
    p.cam.center = {c.center[0], c.center[1], c.center[2]};
    p.cam.pixel00 = {c.pixel00_loc[0], c.pixel00_loc[1], c.pixel00_loc[2]};
    p.cam.du = {c.pixel_delta_u[0], c.pixel_delta_u[1], c.pixel_delta_u[2]};
    p.cam.dv = {c.pixel_delta_v[0], c.pixel_delta_v[1], c.pixel_delta_v[2]};
    p.cam.defocus_angle = c.defocus_angle;
    p.cam.ddu = {c.defocus_disk_u[0], c.defocus_disk_u[1], c.defocus_disk_u[2]};
    p.cam.ddv = {c.defocus_disk_v[0], c.defocus_disk_v[1], c.defocus_disk_v[2]};
    p.n = h->n; p.n_padded = h->n_padded;
    p.geom_a = (const T*)h->geom_a; p.screen.shade_tbl = (const T*)h->shade_tbl;
    p.cold.rng = h->rng; p.cold.fb = (T*)h->fb;
    p.cold.local_rows = h->local_rows; p.cold.rank = h->rank; p.cold.nranks = h->nranks; p.cold.strip_rows = h->strip_rows;
    return p;
}

}  // namespace
