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

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "rtiow.h"
#ifdef RTIOW_DEBUG_API
#include "rtiow_debug.h"
#endif

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
    decltype(&ncclCommAbort) CommAbort = nullptr;                    // optional (null in a library without them): the completion phase of the exchange
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;    // then watches the streams alone
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
        CommAbort = (decltype(CommAbort))dlsym(lib, "ncclCommAbort");
        CommGetAsyncError = (decltype(CommGetAsyncError))dlsym(lib, "ncclCommGetAsyncError");
        return true;
    }
};

// ---- the exchange as a schedule over a table of calls --------------------------------------------------------
// Everything rtiow_group_gather asks of HIP and RCCL between its two timing events goes through this table, so
// that the schedule itself -- which call, on which device / stream / communicator, with which counts and offsets,
// in which order, and which events fence it -- can run against a recorder on a machine without GPUs
// (rtiow_debug_gather_schedule, tests/test_group_schedule.py).  The real table (hip_backend) forwards to
// hip* / nccl*; every entry returns 0 or the failing call's error code.
struct GatherBackend {
    void* self;
    int (*set_device)(void* self, int dev);
    int (*stream_wait_event)(void* self, hipStream_t s, hipEvent_t e);
    int (*event_record)(void* self, hipEvent_t e, hipStream_t s);
    int (*copy_async)(void* self, void* dst, const void* src, size_t bytes, hipStream_t s);                                   // same device
    int (*copy_peer_async)(void* self, void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t s);  // across devices
    int (*group_start)(void* self);
    int (*group_end)(void* self);
    int (*send)(void* self, const void* buf, size_t count, int fp64, int peer, void* comm, hipStream_t s);
    int (*recv)(void* self, void* buf, size_t count, int fp64, int peer, void* comm, hipStream_t s);
    // host mode and the fallback between transports
    int (*stream_sync)(void* self, hipStream_t s);
    int (*copy_via_host)(void* self, void* dst, int dst_dev, hipStream_t dst_stream, const void* src, int src_dev, size_t bytes);   // blocking D2H into a bounce buffer, H2D on dst_stream, awaited
    int (*drain)(void* self, int ndev, const int* dev);      // after a failing transport: every device idle, sticky errors cleared (never fails the chain)
    // COMPLETION of the exchange: ncclSend / ncclRecv / ncclGroupEnd and hipMemcpyPeerAsync mostly fail or stall AFTER they were
    // enqueued.  await = wait until everything enqueued on `s` is done, watching the communicator's asynchronous error state while waiting
    // (comm null: peer copies) and giving up after a deadline; 0, or the error with *where naming its source.  abort_comms = ncclCommAbort
    // on every communicator (a transport that failed is never used again, and its kernels must be gone before the devices are drained).
    int (*await)(void* self, hipStream_t s, void* comm, const char** where);
    int (*abort_comms)(void* self);
};

struct GatherInputs {
    int n, mode, W, fp64;                        // ranks, RTIOW_GATHER_RCCL | _PEER, image width, element type
    const int* dev;                              // device of every rank
    const int* rows;                             // local rows of every rank
    const unsigned long long* offsets;           // element offset of every rank's block in the staging buffer (rank-major)
    void* const* fb;                             // every rank's strip framebuffer (on its device)
    void* staged;                                // device 0: rank-major staging buffer
    const hipStream_t* stream;                   // every rank's stream; stream[0] carries the exchange on device 0
    const hipEvent_t* done;                      // per rank: its render (and, peer mode, afterwards its copy) is complete
    hipEvent_t g0;                               // device 0: start of the timed exchange
    void* const* comms;                          // RCCL mode: every rank's communicator
};

// Rank k's strips -> staged + offsets[k], for every k, fenced so that (i) nothing starts before the LAST render has
// finished and device 0 has recorded g0 (the timed region then holds the exchange alone), and (ii) stream[0] has
// every block when this returns its last call (the caller launches the de-interleave on stream[0] next).
// Ranks without rows take no part on either side.  `where` names the failing call.
int run_gather_schedule(const GatherInputs& in, const GatherBackend& be, const char*& where) {
#define GB(call, name) do { const int rc_ = (call); if (rc_) { where = name; return rc_; } } while (0)
    const size_t es = in.fp64 ? 8 : 4;
    hipStream_t s0 = in.stream[0];
    GB(be.set_device(be.self, in.dev[0]), "hipSetDevice");
    for (int k = 1; k < in.n; ++k) GB(be.stream_wait_event(be.self, s0, in.done[k]), "hipStreamWaitEvent");
    GB(be.event_record(be.self, in.g0, s0), "hipEventRecord");
    if (in.mode == RTIOW_GATHER_RCCL) {
        for (int k = 1; k < in.n; ++k) {          // a rank's send must not start before device 0 opened the timed region
            GB(be.set_device(be.self, in.dev[k]), "hipSetDevice");
            GB(be.stream_wait_event(be.self, in.stream[k], in.g0), "hipStreamWaitEvent");
        }
        int r = be.group_start(be.self);
        const char* failed = "ncclGroupStart";
        for (int k = 0; k < in.n && r == 0; ++k) {
            const size_t count = (size_t)in.rows[k] * in.W * 3;
            if (count == 0) continue;
            failed = "ncclSend";
            r = be.send(be.self, in.fb[k], count, in.fp64, 0, in.comms[k], in.stream[k]);
            if (r == 0) { failed = "ncclRecv"; r = be.recv(be.self, (char*)in.staged + in.offsets[k] * es, count, in.fp64, k, in.comms[0], s0); }
        }
        const int r2 = be.group_end(be.self);      // always closed, also after a failing call inside the group
        if (r == 0 && r2 != 0) { r = r2; failed = "ncclGroupEnd"; }
        if (r != 0) { where = failed; return r; }
        // completion: every sender's stream, then stream 0 (the receives), each with its communicator's asynchronous error state
        for (int k = in.n - 1; k >= 0; --k) {
            if ((size_t)in.rows[k] * in.W * 3 == 0 && k != 0) continue;
            GB(be.set_device(be.self, in.dev[k]), "hipSetDevice");
            const char* aw = "exchange";
            const int ra = be.await(be.self, in.stream[k], in.comms[k], &aw);
            if (ra) { where = aw; return ra; }
        }
    } else if (in.mode == RTIOW_GATHER_HOST) {
        // last resort: blocking copies through the host, rank by rank (every rank's render awaited first; stream 0 holds every block at the end)
        for (int k = 0; k < in.n; ++k) {
            const size_t bytes = (size_t)in.rows[k] * in.W * 3 * es;
            if (bytes == 0) continue;
            GB(be.set_device(be.self, in.dev[k]), "hipSetDevice");
            GB(be.stream_sync(be.self, in.stream[k]), "hipStreamSynchronize");
            GB(be.copy_via_host(be.self, (char*)in.staged + in.offsets[k] * es, in.dev[0], s0, in.fb[k], in.dev[k], bytes), "hipMemcpy (host-staged)");
        }
    } else {
        for (int k = 0; k < in.n; ++k) {
            const size_t bytes = (size_t)in.rows[k] * in.W * 3 * es;
            if (bytes == 0) continue;
            void* dst = (char*)in.staged + in.offsets[k] * es;
            if (k == 0 || in.dev[k] == in.dev[0]) {
                GB(be.set_device(be.self, in.dev[0]), "hipSetDevice");
                GB(be.copy_async(be.self, dst, in.fb[k], bytes, s0), "hipMemcpyAsync");   // s0 already waits for rank k's render
            } else {
                GB(be.set_device(be.self, in.dev[k]), "hipSetDevice");
                GB(be.stream_wait_event(be.self, in.stream[k], in.g0), "hipStreamWaitEvent");
                GB(be.copy_peer_async(be.self, dst, in.dev[0], in.fb[k], in.dev[k], bytes, in.stream[k]), "hipMemcpyPeerAsync");
                GB(be.event_record(be.self, in.done[k], in.stream[k]), "hipEventRecord");
                GB(be.set_device(be.self, in.dev[0]), "hipSetDevice");
                GB(be.stream_wait_event(be.self, s0, in.done[k]), "hipStreamWaitEvent");
            }
        }
        // completion: the streams that carry a copy across devices, then stream 0
        for (int k = in.n - 1; k >= 0; --k) {
            const bool across = k != 0 && in.dev[k] != in.dev[0] && (size_t)in.rows[k] * in.W * 3 != 0;
            if (!across && k != 0) continue;
            GB(be.set_device(be.self, in.dev[k]), "hipSetDevice");
            const char* aw = "exchange";
            const int ra = be.await(be.self, in.stream[k], nullptr, &aw);
            if (ra) { where = aw; return ra; }
        }
    }
    GB(be.set_device(be.self, in.dev[0]), "hipSetDevice");
#undef GB
    return 0;
}

// A transport that fails AT GATHER TIME (the first ncclGroupEnd / send / recv between distinct devices, the first peer copy) must not end
// the frame when another one can carry it: with `fallback` the exchange is tried RCCL -> peer copies -> host-staged copies, in this
// process, each attempt from the top of the schedule after the devices have been drained.  `attempts` collects what failed and why
// (rtiow_group_transport_note); in.mode is left at the transport that carried the image.  Without `fallback` (an explicitly requested
// transport) the first failure is the result.
int run_gather_with_fallback(GatherInputs& in, const GatherBackend& be, bool fallback, const char*& where, std::string& attempts,
                             const std::function<std::string(int, const char*)>& describe) {
    for (;;) {
        where = "";
        const int rc = run_gather_schedule(in, be, where);
        if (rc == 0) return 0;
        if (!fallback || in.mode == RTIOW_GATHER_HOST) {
            if (in.mode == RTIOW_GATHER_RCCL) (void)be.abort_comms(be.self);      // a communicator that has failed is in no defined state: aborted, never destroyed
            return rc;
        }
        const int next = in.mode == RTIOW_GATHER_RCCL ? RTIOW_GATHER_PEER : RTIOW_GATHER_HOST;
        if (!attempts.empty()) attempts += "; ";
        attempts += std::string(in.mode == RTIOW_GATHER_RCCL ? "RCCL" : "peer copies") + " failed at gather time in " + where + " (" + describe(rc, where) + "), fell back to " +
                    (next == RTIOW_GATHER_PEER ? "peer copies" : "host-staged copies");
        if (in.mode == RTIOW_GATHER_RCCL) (void)be.abort_comms(be.self);      // its kernels must not be left waiting for a peer when the devices are drained
        (void)be.drain(be.self, in.n, in.dev);
        in.mode = next;
    }
}

// The real table.  self = the group's RcclApi (null entries are never reached in peer mode).
struct HipBackendSelf { RcclApi* rccl; int last_nccl; std::vector<unsigned char>* bounce; std::vector<ncclComm_t>* comms; bool* rccl_dead; };
// How long the completion phase waits for a stream before it calls the exchange stalled (RTIOW_GATHER_TIMEOUT_MS; the exchange of a 1080p frame takes < 0.1 ms).
inline double gather_timeout_ms() {
    const char* e = getenv("RTIOW_GATHER_TIMEOUT_MS");
    const double v = e ? atof(e) : 0.0;
    return v > 0 ? v : 20000.0;
}
GatherBackend hip_backend(HipBackendSelf* self) {
    GatherBackend b;
    b.self = self;
    b.set_device = [](void*, int dev) { return (int)hipSetDevice(dev); };
    b.stream_wait_event = [](void*, hipStream_t s, hipEvent_t e) { return (int)hipStreamWaitEvent(s, e, 0); };
    b.event_record = [](void*, hipEvent_t e, hipStream_t s) { return (int)hipEventRecord(e, s); };
    b.copy_async = [](void*, void* dst, const void* src, size_t bytes, hipStream_t s) { return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s); };
    b.copy_peer_async = [](void*, void* dst, int dd, const void* src, int sd, size_t bytes, hipStream_t s) { return (int)hipMemcpyPeerAsync(dst, dd, src, sd, bytes, s); };
    b.group_start = [](void* p) { auto* h = (HipBackendSelf*)p; return h->last_nccl = (int)h->rccl->GroupStart(); };
    b.group_end = [](void* p) { auto* h = (HipBackendSelf*)p; const int r = (int)h->rccl->GroupEnd(); if (r) h->last_nccl = r; return r; };
    b.send = [](void* p, const void* buf, size_t count, int fp64, int peer, void* comm, hipStream_t s) {
        auto* h = (HipBackendSelf*)p; return h->last_nccl = (int)h->rccl->Send(buf, count, fp64 ? ncclDouble : ncclFloat, peer, (ncclComm_t)comm, s); };
    b.recv = [](void* p, void* buf, size_t count, int fp64, int peer, void* comm, hipStream_t s) {
        auto* h = (HipBackendSelf*)p; return h->last_nccl = (int)h->rccl->Recv(buf, count, fp64 ? ncclDouble : ncclFloat, peer, (ncclComm_t)comm, s); };
    b.stream_sync = [](void*, hipStream_t s) { return (int)hipStreamSynchronize(s); };
    b.copy_via_host = [](void* p, void* dst, int dst_dev, hipStream_t dst_stream, const void* src, int src_dev, size_t bytes) {
        auto* h = (HipBackendSelf*)p;
        if (h->bounce->size() < bytes) h->bounce->resize(bytes);
        hipError_t e = hipSetDevice(src_dev);
        if (e == hipSuccess) e = hipMemcpy(h->bounce->data(), src, bytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipSetDevice(dst_dev);
        if (e == hipSuccess) e = hipMemcpyAsync(dst, h->bounce->data(), bytes, hipMemcpyHostToDevice, dst_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(dst_stream);      // the bounce buffer is reused by the next rank
        return (int)e; };
    b.await = [](void* p, hipStream_t s, void* comm, const char** where) {
        auto* h = (HipBackendSelf*)p;
        const auto t0 = std::chrono::steady_clock::now();
        const double limit = gather_timeout_ms();
        for (unsigned spins = 0;; ++spins) {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) { *where = "hipStreamQuery (exchange)"; return (int)q; }
            if (comm && h->rccl->CommGetAsyncError) {
                ncclResult_t ae = ncclSuccess;
                const ncclResult_t r = h->rccl->CommGetAsyncError((ncclComm_t)comm, &ae);
                if (r != ncclSuccess || (ae != ncclSuccess && ae != ncclInProgress)) { *where = "ncclCommGetAsyncError"; return h->last_nccl = (int)(r != ncclSuccess ? r : ae); }
            }
            if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > limit) { *where = "hipStreamQuery (exchange stalled past RTIOW_GATHER_TIMEOUT_MS)"; return (int)hipErrorNotReady; }
            if (spins > 2000) std::this_thread::yield();
        }
        if (comm && h->rccl->CommGetAsyncError) {           // a stream can drain although its communicator has recorded an error
            ncclResult_t ae = ncclSuccess;
            if (h->rccl->CommGetAsyncError((ncclComm_t)comm, &ae) == ncclSuccess && ae != ncclSuccess && ae != ncclInProgress) { *where = "ncclCommGetAsyncError"; return h->last_nccl = (int)ae; }
        }
        return 0; };
    b.abort_comms = [](void* p) {
        auto* h = (HipBackendSelf*)p;
        if (h->comms) {
            for (ncclComm_t& c : *h->comms) if (c) { if (h->rccl->CommAbort) (void)h->rccl->CommAbort(c); c = nullptr; }   // aborted communicators are gone: never destroyed again
            h->comms->clear();
        }
        if (h->rccl_dead) *h->rccl_dead = true;
        return 0; };
    b.drain = [](void*, int ndev, const int* dev) {
        for (int k = 0; k < ndev; ++k) { if (hipSetDevice(dev[k]) == hipSuccess) (void)hipDeviceSynchronize(); (void)hipGetLastError(); }
        (void)hipSetDevice(dev[0]); (void)hipGetLastError();
        return 0; };
    return b;
}

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
    bool rccl_dead = false;                     // RCCL failed at gather time: its communicators were aborted, the group never asks it again
    int rccl_version = 0;
    std::string transport_note;                 // why auto mode fell back, if it did (at creation / first gather, or at gather time)
    std::vector<unsigned char> bounce;          // RTIOW_GATHER_HOST: the host bounce buffer
    int peer_links = 0;                         // peer mode: ranks whose device has direct access to device 0 enabled
    double create_ms = 0;                       // wall time of rtiow_group_create (contexts, streams, communicator)
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

// NOTE for callers that own stdout: RCCL prints a version banner on STDOUT when a process creates its first
// communicator ("RCCL version : ... Librccl path : ...").  The library does not touch file descriptors (a dup2 on
// the process-wide fd 1 from inside a library call races with every other thread that writes there): the drop-in
// executables point fd 1 at stderr around rtiow_group_create themselves (csrc/host/main.cpp), bench.py does the
// same for its whole run.

bool distinct_devices(const rtiow_group_s* g) {
    std::vector<int> d = g->dev;
    std::sort(d.begin(), d.end());
    return std::adjacent_find(d.begin(), d.end()) == d.end();
}

// Peer mode, however it was chosen: direct xGMI copies where the devices can reach each other (else the runtime
// stages hipMemcpyPeerAsync through the host).  The copy runs on the SOURCE device's stream, so the source enables
// access to device 0.  "Already enabled" is success.
void enable_peer_access_to_root(rtiow_group_s* g) {
    for (int k = 1; k < g->n; ++k) {
        const int dk = g->dev[(size_t)k], d0 = g->dev[0];
        int can = 0;
        if (dk == d0 || hipDeviceCanAccessPeer(&can, dk, d0) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
        if (hipSetDevice(dk) != hipSuccess) { (void)hipGetLastError(); continue; }
        const hipError_t e = hipDeviceEnablePeerAccess(d0, 0);
        if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) ++g->peer_links;
        (void)hipGetLastError();
    }
    (void)hipSetDevice(g->dev[0]);              // leave device 0 current, as every caller finds it
    (void)hipGetLastError();
}

// Decide the transport once.  AUTO prefers RCCL and records why it did not get it.
int resolve_transport(rtiow_group_s* g) {
    if (g->gather_mode) return 0;
    if (g->gather_requested == RTIOW_GATHER_PEER) { g->gather_mode = RTIOW_GATHER_PEER; enable_peer_access_to_root(g); return 0; }
    if (g->gather_requested == RTIOW_GATHER_HOST) { g->gather_mode = RTIOW_GATHER_HOST; return 0; }
    std::string why;
    bool ok = true;
    if (!distinct_devices(g)) { ok = false; why = "the group maps several ranks to one device (ncclCommInitAll needs distinct devices)"; }
    if (ok && !g->rccl.load(why)) ok = false;
    if (ok) {
        g->comms.assign((size_t)g->n, nullptr);
        ncclResult_t r;
        r = g->rccl.CommInitAll(g->comms.data(), g->n, g->dev.data());
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
    enable_peer_access_to_root(g);
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

namespace { thread_local std::string g_create_error; }   // why the calling thread's last rtiow_group_create failed

extern "C" {

const char* rtiow_group_create_error(void) { return g_create_error.c_str(); }

int rtiow_group_create(int ngpus, const int* devices, int precision, int strip_rows, int gather, rtiow_group* out) {
    if (!out) return RTIOW_E_BADARG;
    *out = nullptr;
    if (ngpus < 1 || ngpus > 64 || (precision != 32 && precision != 64) || strip_rows < 1 ||
        (gather != RTIOW_GATHER_AUTO && gather != RTIOW_GATHER_RCCL && gather != RTIOW_GATHER_PEER && gather != RTIOW_GATHER_HOST)) return RTIOW_E_BADARG;
    rtiow_group_s* g = new (std::nothrow) rtiow_group_s();
    if (!g) return RTIOW_E_NOMEM;
    const auto t_create = std::chrono::steady_clock::now();
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
    g->stats.create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count();
    g->stats.peer_links = g->peer_links;
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
    {
        std::vector<void*> comms((size_t)g->n, nullptr);
        for (size_t k = 0; k < g->comms.size() && k < comms.size(); ++k) comms[k] = (void*)g->comms[k];
        GatherInputs in;
        in.n = g->n; in.mode = g->gather_mode; in.W = g->W; in.fp64 = g->precision == 64;
        in.dev = g->dev.data(); in.rows = g->rows.data(); in.offsets = g->host_offsets.data(); in.fb = fb.data(); in.staged = g->staged;
        in.stream = g->stream.data(); in.done = g->done.data(); in.g0 = g->g0; in.comms = comms.data();
        HipBackendSelf self{&g->rccl, 0, &g->bounce, &g->comms, &g->rccl_dead};
        const char* where = "";
        std::string attempts;
        auto describe = [g](int rc, const char* w) { return std::string(w[0] == 'n' ? g->rccl.GetErrorString((ncclResult_t)rc) : hipGetErrorString((hipError_t)rc)); };
        const int mode_before = in.mode;
        const int src = run_gather_with_fallback(in, hip_backend(&self), g->gather_requested == RTIOW_GATHER_AUTO, where, attempts, describe);
        if (in.mode != mode_before) {
            // the transport changed at gather time: it stays changed (peer access enabled for peer copies), and the note says why
            g->gather_mode = in.mode;
            if (in.mode == RTIOW_GATHER_PEER) enable_peer_access_to_root(g);
            g->transport_note = g->transport_note.empty() ? attempts : g->transport_note + "; " + attempts;
        }
        if (src) {
            if (where[0] == 'n') return gfail(g, RTIOW_E_STATE, std::string("RCCL gather failed in ") + where + ": " + g->rccl.GetErrorString((ncclResult_t)src));
            return gfail(g, src, std::string("HIP_SAFE_CALL: ") + hipGetErrorString((hipError_t)src) + " in the exchange (" + where + ")");
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

// ---- test hook: the exchange's schedule against a recorder (no GPU, no RCCL) ---------------------------------
// Runs run_gather_schedule -- the function rtiow_group_gather runs -- with made-up handles and a table that only
// writes down what it is asked: one record of 8 int64 per call, {op, device in effect, a, b, c, d, e, f}:
//   op 1 set_device        a = device
//   op 2 stream_wait_event a = stream id, b = event id
//   op 3 event_record      a = event id, b = stream id
//   op 4 copy_async        a = dst byte offset in the staging buffer, b = source rank, c = bytes, d = stream id
//   op 5 copy_peer_async   a = dst byte offset, b = source rank, c = bytes, d = stream id, e = dst device, f = src device
//   op 6 group_start       op 7 group_end
//   op 8 send              a = source rank (by its framebuffer), b = count (elements), c = fp64, d = peer, e = communicator id, f = stream id
//   op 9 recv              a = dst byte offset, b = count, c = fp64, d = peer, e = communicator id, f = stream id
//   op 10 stream_sync      a = stream id            op 11 copy_via_host  a = dst byte offset, b = source rank, c = bytes, d = dst stream, e = dst device, f = src device
//   op 12 (closing record of a fallback run) a = transport that carried the image, b = length of the note
//   op 13 drain            a = devices              op 15 abort_comms
//   op 14 await            a = stream id, b = communicator id (0: none) -- the completion phase: "enqueue ok, completion fails" is a failure HERE
// ids: stream of rank k = 1 + k; event done[k] = 100 + k, g0 = 99; communicator of rank k = 200 + k.
// fail_at >= 0 makes the (fail_at)-th call return an error (the schedule must stop, or -- inside an RCCL group --
// still close the group).  Returns the number of records, or a negative RTIOW_E_* code.
#ifdef RTIOW_DEBUG_API       // test hook (include/rtiow_debug.h): compiled into lib/librtiow_hip_debug.so only
int rtiow_debug_gather_schedule(int n, const int* devices, const int* rows, int W, int precision, int mode, int fail_at,
                                int64_t* records, size_t cap_records, int* schedule_rc) {
    if (n < 1 || n > 64 || !devices || !rows || W < 1 || (precision != 32 && precision != 64) || !records || !schedule_rc ||
        ((mode & 0xff) != RTIOW_GATHER_RCCL && (mode & 0xff) != RTIOW_GATHER_PEER && (mode & 0xff) != RTIOW_GATHER_HOST) || (mode & ~0x1ff)) return RTIOW_E_BADARG;
    // mode | 0x100: with the fallback chain of RTIOW_GATHER_AUTO (run_gather_with_fallback); the last record is then {12, -, transport that carried the image}
    const bool fallback = (mode & 0x100) != 0;
    mode &= 0xff;
    struct Rec { std::vector<int64_t> v; int device = -1; int calls = 0, fail_at = -1; char* staged; std::vector<void*> fb;
                 int hit() { return calls++ == fail_at ? 999 : 0; }
                 int64_t rank_of(const void* p) const { for (size_t k = 0; k < fb.size(); ++k) if (fb[k] == p) return (int64_t)k; return -1; }
                 void put(int64_t op, int64_t a = 0, int64_t b = 0, int64_t c = 0, int64_t d = 0, int64_t e = 0, int64_t f = 0) {
                     const int64_t r[8] = {op, device, a, b, c, d, e, f}; v.insert(v.end(), r, r + 8); } } rec;
    rec.fail_at = fail_at;
    rec.staged = (char*)(uintptr_t)0x10000000;
    const size_t es = precision == 64 ? 8 : 4;
    std::vector<unsigned long long> offsets((size_t)n, 0);
    std::vector<hipStream_t> stream((size_t)n);
    std::vector<hipEvent_t> done((size_t)n);
    std::vector<void*> comms((size_t)n);
    unsigned long long off = 0;
    for (int k = 0; k < n; ++k) {
        offsets[(size_t)k] = off; off += (unsigned long long)rows[k] * W * 3;
        rec.fb.push_back((void*)(uintptr_t)(0x1000 * (k + 1)));
        stream[(size_t)k] = (hipStream_t)(uintptr_t)(1 + k); done[(size_t)k] = (hipEvent_t)(uintptr_t)(100 + k); comms[(size_t)k] = (void*)(uintptr_t)(200 + k);
    }
    auto id = [](const void* p) { return (int64_t)(uintptr_t)p; };
    GatherBackend b;
    b.self = &rec;
    b.set_device = [](void* p, int dev) { auto* r = (Rec*)p; r->put(1, dev); r->device = dev; return r->hit(); };
    b.stream_wait_event = [](void* p, hipStream_t s, hipEvent_t e) { auto* r = (Rec*)p; r->put(2, (int64_t)(uintptr_t)s, (int64_t)(uintptr_t)e); return r->hit(); };
    b.event_record = [](void* p, hipEvent_t e, hipStream_t s) { auto* r = (Rec*)p; r->put(3, (int64_t)(uintptr_t)e, (int64_t)(uintptr_t)s); return r->hit(); };
    b.copy_async = [](void* p, void* dst, const void* src, size_t bytes, hipStream_t s) { auto* r = (Rec*)p; r->put(4, (char*)dst - r->staged, r->rank_of(src), (int64_t)bytes, (int64_t)(uintptr_t)s); return r->hit(); };
    b.copy_peer_async = [](void* p, void* dst, int dd, const void* src, int sd, size_t bytes, hipStream_t s) {
        auto* r = (Rec*)p; r->put(5, (char*)dst - r->staged, r->rank_of(src), (int64_t)bytes, (int64_t)(uintptr_t)s, dd, sd); return r->hit(); };
    b.group_start = [](void* p) { auto* r = (Rec*)p; r->put(6); return r->hit(); };
    b.group_end = [](void* p) { auto* r = (Rec*)p; r->put(7); return r->hit(); };
    b.send = [](void* p, const void* buf, size_t count, int fp64, int peer, void* comm, hipStream_t s) {
        auto* r = (Rec*)p; r->put(8, r->rank_of(buf), (int64_t)count, fp64, peer, (int64_t)(uintptr_t)comm, (int64_t)(uintptr_t)s); return r->hit(); };
    b.recv = [](void* p, void* buf, size_t count, int fp64, int peer, void* comm, hipStream_t s) {
        auto* r = (Rec*)p; r->put(9, (char*)buf - r->staged, (int64_t)count, fp64, peer, (int64_t)(uintptr_t)comm, (int64_t)(uintptr_t)s); return r->hit(); };
    b.stream_sync = [](void* p, hipStream_t s) { auto* r = (Rec*)p; r->put(10, (int64_t)(uintptr_t)s); return r->hit(); };
    b.copy_via_host = [](void* p, void* dst, int dd, hipStream_t ds, const void* src, int sd, size_t bytes) {
        auto* r = (Rec*)p; r->put(11, (char*)dst - r->staged, r->rank_of(src), (int64_t)bytes, (int64_t)(uintptr_t)ds, dd, sd); return r->hit(); };
    b.drain = [](void* p, int ndev, const int*) { auto* r = (Rec*)p; r->put(13, ndev); return 0; };      // never fails, not counted by fail_at
    b.await = [](void* p, hipStream_t s, void* comm, const char** where) { auto* r = (Rec*)p; r->put(14, (int64_t)(uintptr_t)s, (int64_t)(uintptr_t)comm); *where = "await"; return r->hit(); };
    b.abort_comms = [](void* p) { auto* r = (Rec*)p; r->put(15); return 0; };                             // never fails, not counted by fail_at
    (void)id;
    GatherInputs in;
    in.n = n; in.mode = mode; in.W = W; in.fp64 = precision == 64;
    in.dev = devices; in.rows = rows; in.offsets = offsets.data(); in.fb = rec.fb.data(); in.staged = rec.staged;
    in.stream = stream.data(); in.done = done.data(); in.g0 = (hipEvent_t)(uintptr_t)99; in.comms = comms.data();
    const char* where = "";
    if (fallback) {
        std::string attempts;
        *schedule_rc = run_gather_with_fallback(in, b, true, where, attempts, [](int, const char*) { return std::string("injected"); });
        rec.device = -1;
        rec.put(12, in.mode, (int64_t)attempts.size());
    } else {
        std::string attempts;          // a transport requested outright: the same entry rtiow_group_gather takes, without the chain
        *schedule_rc = run_gather_with_fallback(in, b, false, where, attempts, [](int, const char*) { return std::string("injected"); });
    }
    (void)es;
    const size_t nrec = rec.v.size() / 8;
    if (nrec > cap_records) return RTIOW_E_BADARG;
    std::memcpy(records, rec.v.data(), rec.v.size() * sizeof(int64_t));
    return (int)nrec;
}
#endif  // RTIOW_DEBUG_API

}  // extern "C"
