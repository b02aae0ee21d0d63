// rtiow_group.hip -- single-process multi-GPU driver of the render path.
//
// The reference is single-GPU (/root/reference/src/GlobalFloatCUDAInOneWeekend/main.cu:81
// cudaSetDevice(0)); sharding the frame over the GPUs of one node is new work that north_star
// asks for: "the framebuffer shards by pixel-row tiles across the 8 GPUs of one node with a final
// RCCL gather over xGMI".  A group is N ordinary handles (include/rtiow.h) -- one per device, each
// with its own stream and events -- plus what only the group needs:
//
//   * shard:   device g renders the interleaved row strips s = g (mod N) (rtiow_set_shard); RNG
//              streams are keyed by the GLOBAL pixel index, so the assembled image is the
//              single-GPU image bit for bit whatever N and the strip height are;
//   * render:  every device's launches are enqueued from one host thread (they are asynchronous),
//              then every stop event is awaited; kernel_ms = max over devices of the HIP-event
//              time of the device's own kernels (the reference's render_only figure, main.cu:334-341);
//   * gather:  ONE exchange after the render: every device sends its strips to device 0, where
//              they land rank-major in a staging buffer and a small kernel de-interleaves them
//              into the full [H][W][3] image.  Transport = RCCL over xGMI: ncclCommInitAll
//              (/opt/rocm/include/rccl/rccl.h:236) + one ncclGroupStart/End of ncclSend/ncclRecv
//              pairs (:700, :722; rank 0 sends to itself, which is what ncclGather does for the
//              root) -- or, when RCCL cannot be loaded or initialised (or the caller asks),
//              hipMemcpyPeerAsync per device.  Payload 1920x1080x12 B = 24.9 MB in total.
//
// librccl.so is dlopen'ed on first use: librtiow_hip.so itself has no link-time dependency on
// RCCL, so single-GPU users never load it and a process that already carries torch's copy of
// RCCL does not get a second one forced upon it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only; every call goes through dlsym'ed pointers

#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rtiow.h"

namespace {

// rank-major staging buffer -> full image.  One thread per pixel component triple.
//   staged: for rank r, rows_of(r) x W x 3 elements starting at offset[r] (elements)
//   a global row j belongs to strip s = j / strip_rows, rank s % N, local row
//   (s / N) * strip_rows + j % strip_rows  (the inverse of rtiow_local_row_map).
template <class T>
__global__ void __launch_bounds__(256)
place_strips_kernel(const T* __restrict__ staged, const unsigned long long* __restrict__ offset, T* __restrict__ full,
                    int W, int H, int nranks, int strip_rows) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // pixel index in the full image
    if (k >= (size_t)W * H) return;
    const int j = (int)(k / W), i = (int)(k - (size_t)j * W);
    const int s = j / strip_rows, r = s % nranks;
    const int jl = (s / nranks) * strip_rows + (j - s * strip_rows);
    const T* src = staged + offset[r] + ((size_t)jl * W + i) * 3;
    T* dst = full + k * 3;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    bool load(std::string& why) {
        if (lib) return true;
        const char* names[] = {getenv("RTIOW_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { why = std::string("dlopen(librccl.so) failed: ") + (dlerror() ? dlerror() : "?"); return false; }
#define RT_SYM(field, name) field = (decltype(field))dlsym(lib, name); if (!field) { why = std::string("librccl.so lacks ") + name; dlclose(lib); lib = nullptr; return false; }
        RT_SYM(CommInitAll, "ncclCommInitAll") RT_SYM(CommDestroy, "ncclCommDestroy") RT_SYM(Send, "ncclSend") RT_SYM(Recv, "ncclRecv")
        RT_SYM(GroupStart, "ncclGroupStart") RT_SYM(GroupEnd, "ncclGroupEnd") RT_SYM(GetErrorString, "ncclGetErrorString")
        RT_SYM(GetVersion, "ncclGetVersion")
#undef RT_SYM
        return true;
    }
};

}  // namespace

struct rtiow_group_s {
    int n = 0, precision = 32, strip_rows = 8;
    std::vector<int> dev;
    std::vector<rtiow_handle> h;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> done;               // per device: its render (and, peer mode, its copy) is complete
    hipEvent_t g0 = nullptr, g1 = nullptr;      // device 0: around the exchange + de-interleave
    int W = 0, H = 0;
    bool have_camera = false, rendered = false;
    // transport
    int gather_requested = RTIOW_GATHER_AUTO, gather_mode = 0;   // resolved at the first gather: RTIOW_GATHER_RCCL | RTIOW_GATHER_PEER
    RcclApi rccl;
    std::vector<ncclComm_t> comms;
    int rccl_version = 0;
    std::string transport_note;                 // why auto mode fell back, if it did
    // device 0 buffers
    void* staged = nullptr; size_t staged_bytes = 0;
    void* full = nullptr; size_t full_bytes = 0;
    unsigned long long* offsets = nullptr;      // device copy of the per-rank element offsets
    std::vector<unsigned long long> host_offsets;
    std::vector<int> rows;                      // local rows per rank
    rtiow_group_stats stats{};
    std::string err;
};

namespace {

size_t gelem(const rtiow_group_s* g) { return g->precision == 64 ? 8 : 4; }

int gfail(rtiow_group_s* g, int code, const std::string& msg) { if (g) g->err = msg; return code; }
int gfail_hip(rtiow_group_s* g, hipError_t e, const char* file, int line) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "HIP_SAFE_CALL: %s %s %d", hipGetErrorString(e), file, line);
    if (g) g->err = buf;
    return (int)e;
}
// a failing member handle: carry its message (the text the reference's CUDA_SAFE_CALL would print)
int gfail_member(rtiow_group_s* g, int k, int rc) {
    g->err = std::string(rtiow_last_error_string(g->h[(size_t)k])) + " [device " + std::to_string(g->dev[(size_t)k]) + "]";
    return rc;
}
#define G_HIP(g, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return gfail_hip((g), e_, __FILE__, __LINE__); } while (0)
#define G_EACH(g, call) do { for (int k_ = 0; k_ < (g)->n; ++k_) { rtiow_handle hk = (g)->h[(size_t)k_]; int rc_ = (call); if (rc_) return gfail_member((g), k_, rc_); } } while (0)

// RCCL prints a version banner on STDOUT when its first communicator is created ("RCCL version :
// ... Librccl path : ..."), and stdout is the reference's CSV fragment (main.cu:342-343, 397-398):
// while RCCL initialises, file descriptor 1 points at stderr.
struct StdoutToStderr {
    int saved = -1;
    StdoutToStderr() { std::fflush(stdout); saved = dup(1); if (saved >= 0) dup2(2, 1); }
    ~StdoutToStderr() { std::fflush(stdout); if (saved >= 0) { dup2(saved, 1); close(saved); } }
};

bool distinct_devices(const rtiow_group_s* g) {
    std::vector<int> d = g->dev;
    std::sort(d.begin(), d.end());
    return std::adjacent_find(d.begin(), d.end()) == d.end();
}

// Decide the transport once.  AUTO prefers RCCL and records why it did not get it.
int resolve_transport(rtiow_group_s* g) {
    if (g->gather_mode) return 0;
    if (g->gather_requested == RTIOW_GATHER_PEER) { g->gather_mode = RTIOW_GATHER_PEER; return 0; }
    std::string why;
    bool ok = true;
    if (!distinct_devices(g)) { ok = false; why = "the group maps several ranks to one device (ncclCommInitAll needs distinct devices)"; }
    if (ok && !g->rccl.load(why)) ok = false;
    if (ok) {
        g->comms.assign((size_t)g->n, nullptr);
        ncclResult_t r;
        { StdoutToStderr quiet; r = g->rccl.CommInitAll(g->comms.data(), g->n, g->dev.data()); }
        if (r != ncclSuccess) {
            ok = false;
            why = std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(r);
            g->comms.clear();
            (void)hipGetLastError();
        } else {
            (void)g->rccl.GetVersion(&g->rccl_version);
        }
    }
    if (ok) { g->gather_mode = RTIOW_GATHER_RCCL; return 0; }
    if (g->gather_requested == RTIOW_GATHER_RCCL) return gfail(g, RTIOW_E_STATE, "RCCL gather requested but unavailable: " + why);
    g->transport_note = why;
    g->gather_mode = RTIOW_GATHER_PEER;
    for (int k = 1; k < g->n; ++k) {            // direct xGMI copies where the devices can reach each other (else the runtime stages them)
        int can = 0;
        if (g->dev[(size_t)k] != g->dev[0] && hipDeviceCanAccessPeer(&can, g->dev[(size_t)k], g->dev[0]) == hipSuccess && can) {
            (void)hipSetDevice(g->dev[(size_t)k]);
            (void)hipDeviceEnablePeerAccess(g->dev[0], 0);
            (void)hipGetLastError();
        }
    }
    return 0;
}

int ensure_group_buffers(rtiow_group_s* g) {
    const size_t es = gelem(g);
    g->rows.assign((size_t)g->n, 0);
    g->host_offsets.assign((size_t)g->n, 0);
    unsigned long long off = 0;
    for (int k = 0; k < g->n; ++k) {
        int r = 0;
        int rc = rtiow_local_rows(g->h[(size_t)k], &r);
        if (rc) return gfail_member(g, k, rc);
        g->rows[(size_t)k] = r;
        g->host_offsets[(size_t)k] = off;
        off += (unsigned long long)r * g->W * 3;
    }
    if (off != (unsigned long long)g->W * g->H * 3) return gfail(g, RTIOW_E_STATE, "shards do not cover the image");
    const size_t need = (size_t)g->W * g->H * 3 * es;
    G_HIP(g, hipSetDevice(g->dev[0]));
    if (g->staged_bytes < need) {
        if (g->staged) { G_HIP(g, hipFree(g->staged)); g->staged = nullptr; g->staged_bytes = 0; }
        G_HIP(g, hipMalloc(&g->staged, need));
        g->staged_bytes = need;
    }
    if (g->full_bytes < need) {
        if (g->full) { G_HIP(g, hipFree(g->full)); g->full = nullptr; g->full_bytes = 0; }
        G_HIP(g, hipMalloc(&g->full, need));
        g->full_bytes = need;
    }
    if (!g->offsets) G_HIP(g, hipMalloc((void**)&g->offsets, sizeof(unsigned long long) * (size_t)g->n));
    G_HIP(g, hipMemcpy(g->offsets, g->host_offsets.data(), sizeof(unsigned long long) * (size_t)g->n, hipMemcpyHostToDevice));
    return 0;
}

}  // namespace

namespace { std::string g_create_error; }   // why the last rtiow_group_create of this process failed

extern "C" {

const char* rtiow_group_create_error(void) { return g_create_error.c_str(); }

int rtiow_group_create(int ngpus, const int* devices, int precision, int strip_rows, int gather, rtiow_group* out) {
    if (!out) return RTIOW_E_BADARG;
    *out = nullptr;
    if (ngpus < 1 || ngpus > 64 || (precision != 32 && precision != 64) || strip_rows < 1 ||
        (gather != RTIOW_GATHER_AUTO && gather != RTIOW_GATHER_RCCL && gather != RTIOW_GATHER_PEER)) return RTIOW_E_BADARG;
    rtiow_group_s* g = new (std::nothrow) rtiow_group_s();
    if (!g) return RTIOW_E_NOMEM;
    g->n = ngpus; g->precision = precision; g->strip_rows = strip_rows; g->gather_requested = gather;
    for (int k = 0; k < ngpus; ++k) g->dev.push_back(devices ? devices[k] : k);
    int rc = 0;
    for (int k = 0; k < ngpus && rc == 0; ++k) {
        rtiow_handle h = nullptr;
        rc = rtiow_create(g->dev[(size_t)k], precision, &h);
        if (rc) break;
        g->h.push_back(h);
        void* st = nullptr;
        hipEvent_t ev = nullptr;
        if ((rc = rtiow_stream(h, &st)) != 0) break;
        g->stream.push_back((hipStream_t)st);
        if ((rc = rtiow_set_shard(h, k, ngpus, strip_rows)) != 0) break;
        if ((rc = (int)hipSetDevice(g->dev[(size_t)k])) != 0) break;
        if ((rc = (int)hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != 0) break;
        g->done.push_back(ev);
    }
    if (rc == 0) {
        rc = (int)hipSetDevice(g->dev[0]);
        if (rc == 0) rc = (int)hipEventCreate(&g->g0);
        if (rc == 0) rc = (int)hipEventCreate(&g->g1);
    }
    // The communicator is created HERE, with the devices' contexts (the reference's cudaSetDevice /
    // event creation, main.cu:81-92, before its end-to-end timer starts): ncclCommInitAll takes
    // seconds (topology discovery), the exchange itself microseconds.
    if (rc == 0) rc = resolve_transport(g);
    if (rc) {
        g_create_error = g->err.empty() ? "rtiow_group_create: device or stream creation failed (error " + std::to_string(rc) + ")" : g->err;
        rtiow_group_destroy(g);
        return rc;
    }
    g_create_error.clear();
    g->stats.ngpus = ngpus; g->stats.strip_rows = strip_rows;
    g->stats.gather_mode = g->gather_mode; g->stats.rccl_version = g->rccl_version;
    *out = g;
    return 0;
}

int rtiow_group_destroy(rtiow_group g) {
    if (!g) return RTIOW_E_BADARG;
    for (size_t k = 0; k < g->h.size(); ++k) { (void)hipSetDevice(g->dev[k]); (void)hipStreamSynchronize(g->stream.size() > k ? g->stream[k] : nullptr); }
    for (ncclComm_t c : g->comms) if (c && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(c);
    if (!g->dev.empty()) {
        (void)hipSetDevice(g->dev[0]);
        if (g->staged) (void)hipFree(g->staged);
        if (g->full) (void)hipFree(g->full);
        if (g->offsets) (void)hipFree(g->offsets);
        if (g->g0) (void)hipEventDestroy(g->g0);
        if (g->g1) (void)hipEventDestroy(g->g1);
    }
    for (size_t k = 0; k < g->done.size(); ++k) { (void)hipSetDevice(g->dev[k]); (void)hipEventDestroy(g->done[k]); }
    for (rtiow_handle h : g->h) (void)rtiow_destroy(h);
    // the RCCL library stays loaded for the life of the process (unloading it under live HIP state is not worth the risk)
    delete g;
    return 0;
}

const char* rtiow_group_last_error_string(rtiow_group g) { return g ? g->err.c_str() : "null group"; }

int rtiow_group_size(rtiow_group g) { return g ? g->n : RTIOW_E_BADARG; }

int rtiow_group_member(rtiow_group g, int rank, rtiow_handle* out) {
    if (!g || !out || rank < 0 || rank >= g->n) return RTIOW_E_BADARG;
    *out = g->h[(size_t)rank];
    return 0;
}

int rtiow_group_set_scene(rtiow_group g, int n, const void* center_radius, const void* albedo_fuzz,
                          const void* refraction_index, const int32_t* type, const int32_t* valid) {
    if (!g) return RTIOW_E_BADARG;
    G_EACH(g, rtiow_set_scene(hk, n, center_radius, albedo_fuzz, refraction_index, type, valid));   // every GPU holds the full scene (<= 43 KB)
    return 0;
}

int rtiow_group_set_camera(rtiow_group g, const void* camera) {
    if (!g || !camera) return RTIOW_E_BADARG;
    G_EACH(g, rtiow_set_camera(hk, camera));
    const int32_t* wh = (const int32_t*)camera;                  // both camera structs start with img_width, img_height
    g->W = wh[0]; g->H = wh[1];
    g->have_camera = true; g->rendered = false;
    return 0;
}

int rtiow_group_set_scene_source(rtiow_group g, int scene_source) {
    if (!g) return RTIOW_E_BADARG;
    G_EACH(g, rtiow_set_scene_source(hk, scene_source));
    return 0;
}

int rtiow_group_set_schedule(rtiow_group g, int schedule, int waves_per_simd) {
    if (!g) return RTIOW_E_BADARG;
    G_EACH(g, rtiow_set_schedule(hk, schedule, waves_per_simd));
    return 0;
}

int rtiow_group_init_rng(rtiow_group g, uint64_t seed) {
    if (!g) return RTIOW_E_BADARG;
    G_EACH(g, rtiow_init_rng(hk, seed));
    return 0;
}

int rtiow_group_render(rtiow_group g, int threads_per_block_row, float* kernel_ms) {
    if (!g) return RTIOW_E_BADARG;
    if (!g->have_camera) return gfail(g, RTIOW_E_STATE, "rtiow_group_render before rtiow_group_set_camera");
    // enqueue everything first (asynchronous launches, one stream per device) ...
    G_EACH(g, rtiow_render_async(hk, threads_per_block_row));
    for (int k = 0; k < g->n; ++k) {
        G_HIP(g, hipSetDevice(g->dev[(size_t)k]));
        G_HIP(g, hipEventRecord(g->done[(size_t)k], g->stream[(size_t)k]));
    }
    // ... then wait for every device: render_only = the slowest device's kernels (main.cu:334-341)
    float worst = 0;
    for (int k = 0; k < g->n; ++k) {
        float ms = 0;
        const int rc = rtiow_render_wait(g->h[(size_t)k], &ms);
        if (rc) return gfail_member(g, k, rc);
        if (k < RTIOW_GROUP_MAX_STATS) g->stats.kernel_ms[k] = ms;
        worst = std::max(worst, ms);
    }
    g->stats.kernel_ms_max = worst;
    g->rendered = true;
    if (kernel_ms) *kernel_ms = worst;
    return 0;
}

// The exchange: every rank's strips -> device 0 (rank-major staging) -> de-interleaved full image.
int rtiow_group_gather(rtiow_group g) {
    if (!g) return RTIOW_E_BADARG;
    if (!g->rendered) return gfail(g, RTIOW_E_STATE, "rtiow_group_gather before rtiow_group_render");
    int rc = resolve_transport(g);
    if (rc) return rc;
    if ((rc = ensure_group_buffers(g)) != 0) return rc;
    const size_t es = gelem(g);
    std::vector<void*> fb((size_t)g->n, nullptr);
    for (int k = 0; k < g->n; ++k) {
        size_t bytes = 0;
        rc = rtiow_framebuffer_device_ptr(g->h[(size_t)k], &fb[(size_t)k], &bytes);
        if (rc) return gfail_member(g, k, rc);
        if (bytes != (size_t)g->rows[(size_t)k] * g->W * 3 * es) return gfail(g, RTIOW_E_STATE, "member framebuffer size mismatch");
    }
    hipStream_t s0 = g->stream[0];
    G_HIP(g, hipSetDevice(g->dev[0]));
    // the timed region starts when the LAST render has finished, so that it holds the exchange alone
    for (int k = 1; k < g->n; ++k) G_HIP(g, hipStreamWaitEvent(s0, g->done[(size_t)k], 0));
    G_HIP(g, hipEventRecord(g->g0, s0));
    if (g->gather_mode == RTIOW_GATHER_RCCL) {
        const ncclDataType_t dt = g->precision == 64 ? ncclDouble : ncclFloat;
        for (int k = 1; k < g->n; ++k) {          // a rank's send must not start before device 0 opened the timed region
            G_HIP(g, hipSetDevice(g->dev[(size_t)k]));
            G_HIP(g, hipStreamWaitEvent(g->stream[(size_t)k], g->g0, 0));
        }
        ncclResult_t r = g->rccl.GroupStart();
        for (int k = 0; k < g->n && r == ncclSuccess; ++k) {
            const size_t count = (size_t)g->rows[(size_t)k] * g->W * 3;
            if (count == 0) continue;
            r = g->rccl.Send(fb[(size_t)k], count, dt, 0, g->comms[(size_t)k], g->stream[(size_t)k]);
            if (r == ncclSuccess)
                r = g->rccl.Recv((char*)g->staged + g->host_offsets[(size_t)k] * es, count, dt, k, g->comms[0], s0);
        }
        const ncclResult_t r2 = g->rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) return gfail(g, RTIOW_E_STATE, std::string("RCCL gather failed: ") + g->rccl.GetErrorString(r));
    } else {
        for (int k = 0; k < g->n; ++k) {
            const size_t bytes = (size_t)g->rows[(size_t)k] * g->W * 3 * es;
            if (bytes == 0) continue;
            void* dst = (char*)g->staged + g->host_offsets[(size_t)k] * es;
            if (k == 0 || g->dev[(size_t)k] == g->dev[0]) {
                G_HIP(g, hipSetDevice(g->dev[0]));
                G_HIP(g, hipMemcpyAsync(dst, fb[(size_t)k], bytes, hipMemcpyDeviceToDevice, s0));   // s0 already waits for rank k's render
            } else {
                G_HIP(g, hipSetDevice(g->dev[(size_t)k]));
                G_HIP(g, hipStreamWaitEvent(g->stream[(size_t)k], g->g0, 0));
                G_HIP(g, hipMemcpyPeerAsync(dst, g->dev[0], fb[(size_t)k], g->dev[(size_t)k], bytes, g->stream[(size_t)k]));
                G_HIP(g, hipEventRecord(g->done[(size_t)k], g->stream[(size_t)k]));
                G_HIP(g, hipSetDevice(g->dev[0]));
                G_HIP(g, hipStreamWaitEvent(s0, g->done[(size_t)k], 0));
            }
        }
    }
    G_HIP(g, hipSetDevice(g->dev[0]));
    const size_t npix = (size_t)g->W * g->H;
    const unsigned blocks = (unsigned)((npix + 255) / 256);
    if (g->precision == 64)
        hipLaunchKernelGGL(place_strips_kernel<double>, dim3(blocks), dim3(256), 0, s0, (const double*)g->staged, g->offsets, (double*)g->full, g->W, g->H, g->n, g->strip_rows);
    else
        hipLaunchKernelGGL(place_strips_kernel<float>, dim3(blocks), dim3(256), 0, s0, (const float*)g->staged, g->offsets, (float*)g->full, g->W, g->H, g->n, g->strip_rows);
    G_HIP(g, hipGetLastError());
    G_HIP(g, hipEventRecord(g->g1, s0));
    G_HIP(g, hipEventSynchronize(g->g1));
    float ms = 0;
    G_HIP(g, hipEventElapsedTime(&ms, g->g0, g->g1));
    g->stats.gather_ms = ms;
    g->stats.gather_bytes = (uint64_t)npix * 3 * es;
    g->stats.gather_mode = g->gather_mode;
    g->stats.rccl_version = g->rccl_version;
    return 0;
}

int rtiow_group_framebuffer_device_ptr(rtiow_group g, void** device_ptr, size_t* bytes) {
    if (!g || !device_ptr) return RTIOW_E_BADARG;
    if (!g->full) return gfail(g, RTIOW_E_STATE, "rtiow_group_framebuffer_device_ptr before rtiow_group_gather");
    *device_ptr = g->full;
    if (bytes) *bytes = (size_t)g->W * g->H * 3 * gelem(g);
    return 0;
}

int rtiow_group_read_framebuffer(rtiow_group g, void* host_rgb, size_t bytes) {
    if (!g || !host_rgb) return RTIOW_E_BADARG;
    if (!g->rendered) return gfail(g, RTIOW_E_STATE, "rtiow_group_read_framebuffer before rtiow_group_render");
    const size_t need = (size_t)g->W * g->H * 3 * gelem(g);
    if (bytes < need) return gfail(g, RTIOW_E_BADARG, "rtiow_group_read_framebuffer: host buffer too small");
    const int rc = rtiow_group_gather(g);
    if (rc) return rc;
    G_HIP(g, hipSetDevice(g->dev[0]));
    G_HIP(g, hipMemcpy(host_rgb, g->full, need, hipMemcpyDeviceToHost));
    return 0;
}

int rtiow_group_get_stats(rtiow_group g, rtiow_group_stats* out) {
    if (!g || !out) return RTIOW_E_BADARG;
    *out = g->stats;
    return 0;
}

const char* rtiow_group_transport_note(rtiow_group g) { return g ? g->transport_note.c_str() : ""; }

}  // extern "C"
