// main.cpp -- global-float-hip-raytrace / global-double-hip-raytrace.
//
// Drop-in for the reference executables built from
//   /root/reference/src/GlobalFloatCUDAInOneWeekend/main.cu   (RTIOW_PRECISION=32)
//   /root/reference/src/GlobalDoubleCUDAInOneWeekend/main.cu  (RTIOW_PRECISION=64)
// Same flags, defaults, exit codes, stdout line ("%15.8f,%15.8f\n" = render_only_ms,
// end_to_end_ms), output file name and P3 format, so the reference's *_benchmark.sh,
// process.py and ppm_diff tooling work unchanged.  The phases below are in the order of
// main.cu:37-400; each device phase is one call into the C-ABI (include/rtiow.h).
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "rtiow.h"
#include "rtiow_host.h"

#ifndef RTIOW_PRECISION
#define RTIOW_PRECISION 32
#endif

namespace {

struct Options {
    bool have_scene = false, help = false;
    int scene_id = 0, width = 320, height = 192, samples = 10, bounces = 25, threads = 8;   // main.cu:45-54
    int scene_source = RTIOW_SCENE_GRID;
    bool stats = false;
    bool binary_ppm = false;
    int schedule = RTIOW_SCHED_SORTED;
    // multi-GPU (not in the reference, which is cudaSetDevice(0), main.cu:81): optional, and the
    // default output does not change.  --gpus N renders interleaved row strips on devices 0..N-1 of
    // this node inside this process and gathers them on device 0 (rtiow_group_*, include/rtiow.h).
    int gpus = 0;                       // 0: flag absent -> the single-handle path
    std::vector<int> devices;           // --devices a,b,c: explicit device per rank (a device may repeat)
    int gather = RTIOW_GATHER_AUTO;
    int strip_rows = 0;                 // 0: 8 rows for N <= 2, 2 rows above
};

const char* kUsage =
    "Super Raytrace: Raytracing with HIP\n"
    "Usage:\n"
    "  ./hip-raytrace [OPTION...]\n"
    "\n"
    "      --scene_id arg  ID of the scene to render\n"
    "      --width arg     Width of the output image (default: 320)\n"
    "      --height arg    Height of the output image (default: 192)\n"
    "      --samples arg   Number of samples per pixel (default: 10)\n"
    "      --bounces arg   Maximum number of ray bounces (default: 25)\n"
    "      --threads arg   Number of threads per 2-D thread block row. (default: \n"
    "                      8)\n"
    "  -h, --help          Print usage\n";

[[noreturn]] void parse_abort(const std::string& what) {
    // cxxopts throws on a bad option and the reference does not catch: std::terminate.
    std::fprintf(stderr, "terminate called after throwing an instance of 'cxxopts::exceptions::parsing'\n  what():  %s\n", what.c_str());
    std::abort();
}

int parse_int(const std::string& name, const std::string& text) {
    char* end = nullptr;
    const long v = std::strtol(text.c_str(), &end, 10);
    if (text.empty() || *end != '\0') parse_abort("Argument '" + text + "' failed to parse");
    (void)name;
    return (int)v;
}

Options parse(int argc, char** argv) {
    Options o;
    for (int k = 1; k < argc; ++k) {
        std::string arg = argv[k];
        if (arg == "-h" || arg == "--help") { o.help = true; continue; }
        if (arg.rfind("--", 0) != 0) parse_abort("Option '" + arg + "' does not exist");
        std::string name = arg.substr(2), value;
        const size_t eq = name.find('=');
        bool have_value = false;
        if (eq != std::string::npos) { value = name.substr(eq + 1); name = name.substr(0, eq); have_value = true; }
        if (name == "stats") { o.stats = true; continue; }
        const bool known = name == "scene_id" || name == "width" || name == "height" || name == "samples" ||
                           name == "bounces" || name == "threads" || name == "scene_source" || name == "ppm_format" || name == "schedule" ||
                           name == "gpus" || name == "devices" || name == "gather" || name == "strip_rows";
        if (!known) parse_abort("Option '" + name + "' does not exist");
        if (!have_value) {
            if (k + 1 >= argc) parse_abort("Option '" + name + "' is missing an argument");
            value = argv[++k];
        }
        if (name == "scene_source") {
            if (value == "grid") o.scene_source = RTIOW_SCENE_GRID;
            else if (value == "lds") o.scene_source = RTIOW_SCENE_LDS;
            else if (value == "scalar") o.scene_source = RTIOW_SCENE_SCALAR;
            else parse_abort("Argument '" + value + "' failed to parse");
            continue;
        }
        if (name == "schedule") {
            if (value == "sorted") o.schedule = RTIOW_SCHED_SORTED;
            else if (value == "persistent") o.schedule = RTIOW_SCHED_PERSISTENT;
            else if (value == "static") o.schedule = RTIOW_SCHED_STATIC;
            else parse_abort("Argument '" + value + "' failed to parse");
            continue;
        }
        if (name == "gather") {
            if (value == "auto") o.gather = RTIOW_GATHER_AUTO;
            else if (value == "rccl") o.gather = RTIOW_GATHER_RCCL;
            else if (value == "peer") o.gather = RTIOW_GATHER_PEER;
            else if (value == "host") o.gather = RTIOW_GATHER_HOST;
            else parse_abort("Argument '" + value + "' failed to parse");
            continue;
        }
        if (name == "devices") {
            o.devices.clear();
            size_t pos = 0;
            while (pos <= value.size()) {
                const size_t comma = value.find(',', pos);
                const std::string tok = value.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
                o.devices.push_back(parse_int(name, tok));
                if (comma == std::string::npos) break;
                pos = comma + 1;
            }
            continue;
        }
        if (name == "ppm_format") {
            if (value == "p3") o.binary_ppm = false;
            else if (value == "p6") o.binary_ppm = true;
            else parse_abort("Argument '" + value + "' failed to parse");
            continue;
        }
        const int v = parse_int(name, value);
        if (name == "scene_id") { o.scene_id = v; o.have_scene = true; }
        else if (name == "width") o.width = v;
        else if (name == "height") o.height = v;
        else if (name == "samples") o.samples = v;
        else if (name == "bounces") o.bounces = v;
        else if (name == "gpus") o.gpus = v;
        else if (name == "strip_rows") o.strip_rows = v;
        else o.threads = v;
    }
    return o;
}

// main.cu:14-21: message on stderr, exit with the error code, stdout left as it is.
void check(rtiow_handle h, int rc) {
    if (rc == 0) return;
    std::fprintf(stderr, "%s\n", h ? rtiow_last_error_string(h) : "HIP_SAFE_CALL: device initialisation failed");
    std::exit(rc);
}

void check_group(rtiow_group g, int rc) {
    if (rc == 0) return;
    std::fprintf(stderr, "%s\n", g ? rtiow_group_last_error_string(g) : "HIP_SAFE_CALL: device initialisation failed");
    std::exit(rc);
}

// stdout is the reference's CSV fragment (main.cu:342-343, 397-398) and RCCL prints a version banner there when the
// process creates its first communicator: while the group is created, file descriptor 1 points at stderr.  This
// program owns its stdout and is single-threaded, so the redirection is safe here (it is not inside the library).
struct StdoutToStderr {
    int saved = -1;
    StdoutToStderr() { std::fflush(stdout); saved = dup(1); if (saved >= 0) dup2(2, 1); }
    ~StdoutToStderr() { std::fflush(stdout); if (saved >= 0) { dup2(saved, 1); close(saved); } }
};

// Releases the group (communicators, streams, device buffers) on every way out of main_multi_gpu.
struct GroupGuard {
    rtiow_group g = nullptr;
    ~GroupGuard() { if (g) (void)rtiow_group_destroy(g); }
};

// --gpus N: the same phases as main() below, each through the group twin of the call.
int main_multi_gpu(const Options& opt) {
    const int precision = RTIOW_PRECISION;
    const size_t elem = precision == 64 ? 8 : 4;
    const int n = opt.devices.empty() ? opt.gpus : (int)opt.devices.size();
    if (n < 1 || (opt.gpus > 0 && !opt.devices.empty() && opt.gpus != n)) {
        std::fputs("Error: --gpus must be >= 1 and match --devices.\n", stderr);
        return 1;
    }
    const int strip_rows = opt.strip_rows > 0 ? opt.strip_rows : (n <= 2 ? 8 : 2);
    rtiow_group g = nullptr;
    int rc;
    // Device contexts, streams and the RCCL communicator (ncclCommInitAll: seconds) are created before the end-to-end
    // timer starts, where the reference has cudaSetDevice + event creation (main.cu:81-92 vs :95); --stats reports
    // the time as wall_ms.group_create, and end_to_end does NOT contain it.
    { StdoutToStderr quiet; rc = rtiow_group_create(n, opt.devices.empty() ? nullptr : opt.devices.data(), precision, strip_rows, opt.gather, &g); }   // main.cu:81-92, per device
    if (rc) { std::fprintf(stderr, "HIP_SAFE_CALL: cannot open %d device(s) (error %d) %s\n", n, rc, rtiow_group_create_error()); return rc; }
    GroupGuard guard;
    guard.g = g;
    const auto e2e_start = std::chrono::steady_clock::now();                 // main.cu:95
    auto lap = [last = e2e_start]() mutable {
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - last).count();
        last = now;
        return ms;
    };
    rtiow_camera_f32 cam32; rtiow_camera_f64 cam64;
    void* cam = precision == 64 ? (void*)&cam64 : (void*)&cam32;
    if (rtiow_host_camera(precision, opt.width, opt.height, opt.samples, opt.bounces, cam) != 0) {
        std::fputs("Error: invalid image size.\n", stderr);
        return 1;
    }
    check_group(g, rtiow_group_set_camera(g, cam));
    check_group(g, rtiow_group_set_scene_source(g, opt.scene_source));
    check_group(g, rtiow_group_set_schedule(g, opt.schedule, 0));
    const int slots = rtiow_host_scene_slots(opt.scene_id);
    std::vector<unsigned char> cr(elem * 4 * slots), af(elem * 4 * slots), ri(elem * slots);
    std::vector<int32_t> type(slots), valid(slots);
    rtiow_host_build_scene(opt.scene_id, precision, cr.data(), af.data(), ri.data(), type.data(), valid.data());
    check_group(g, rtiow_group_set_scene(g, slots, cr.data(), af.data(), ri.data(), type.data(), valid.data()));
    const double t_setup = lap();
    check_group(g, rtiow_group_init_rng(g, 1227));
    const double t_rng = lap();
    float render_ms = 0;                                                     // max over devices of the kernel-only event time
    check_group(g, rtiow_group_render(g, opt.threads, &render_ms));
    std::printf("%15.8f,", (double)render_ms);
    std::fflush(stdout);
    const double t_render = lap();
    char name[256];
    rtiow_host_ppm_filename(precision, opt.scene_id, opt.width, opt.height, opt.samples, opt.bounces, opt.threads, name, sizeof name);
    std::vector<unsigned char> rgb(elem * 3 * (size_t)opt.width * opt.height);
    check_group(g, rtiow_group_read_framebuffer(g, rgb.data(), rgb.size()));   // the one exchange + de-interleave on device 0 + D2H
    const double t_read = lap();
    const int wrc = opt.binary_ppm ? rtiow_host_write_ppm_binary(name, precision, opt.width, opt.height, rgb.data())
                                   : rtiow_host_write_ppm(name, precision, opt.width, opt.height, rgb.data());
    if (wrc != 0) {
        std::fprintf(stderr, "Error: Could not open file for writing: %s\n", name);
        return -1;
    }
    const double t_write = lap();
    rtiow_group_stats gs;
    std::memset(&gs, 0, sizeof gs);
    rtiow_group_get_stats(g, &gs);
    const std::string note = rtiow_group_transport_note(g);
    guard.g = nullptr;
    check_group(g, rtiow_group_destroy(g));
    const double e2e_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - e2e_start).count();
    std::printf("%15.8f\n", e2e_ms);
    if (opt.stats) {
        const double rays = (double)opt.width * opt.height * opt.samples;
        std::string per;
        for (int k = 0; k < n && k < RTIOW_GROUP_MAX_STATS; ++k) { char b[32]; std::snprintf(b, sizeof b, "%s%.6f", k ? ", " : "", gs.kernel_ms[k]); per += b; }
        std::fprintf(stderr,
                     "{\"mrays_per_s\": %.3f, \"render_ms\": %.6f, \"gpus\": %d, \"strip_rows\": %d, \"kernel_ms\": [%s], "
                     "\"gather\": \"%s\", \"rccl_version\": %d, \"gather_ms\": %.6f, \"gather_bytes\": %llu, \"transport_note\": \"%s\", "
                     "\"peer_links\": %d, "
                     "\"wall_ms\": {\"group_create\": %.3f, \"setup\": %.3f, \"rng_init\": %.3f, \"render\": %.3f, \"gather_and_readback\": %.3f, \"ppm_write\": %.3f, \"end_to_end\": %.3f}, "
                     "\"end_to_end_excludes\": \"group_create\"}\n",
                     render_ms > 0 ? rays / render_ms / 1e3 : 0.0, (double)render_ms, n, strip_rows, per.c_str(),
                     gs.gather_mode == RTIOW_GATHER_RCCL ? "rccl" : (gs.gather_mode == RTIOW_GATHER_HOST ? "host" : "peer"), gs.rccl_version, gs.gather_ms, (unsigned long long)gs.gather_bytes, note.c_str(),
                     gs.peer_links, gs.create_ms, t_setup, t_rng, t_render, t_read, t_write, e2e_ms);
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    const Options opt = parse(argc, argv);                                   // main.cu:39-77
    if (opt.help) { std::fputs(kUsage, stdout); std::fputs("\n", stdout); return 0; }
    if (!opt.have_scene) {
        std::fputs("Error: --scene_id is required.\n", stderr);
        std::fputs(kUsage, stdout); std::fputs("\n", stdout);
        return 1;
    }
    if (opt.gpus != 0 || !opt.devices.empty()) return main_multi_gpu(opt);
    const int precision = RTIOW_PRECISION;
    const size_t elem = precision == 64 ? 8 : 4;

    rtiow_handle h = nullptr;
    int rc = rtiow_create(0, precision, &h);                                 // main.cu:81-92
    if (rc) { std::fprintf(stderr, "HIP_SAFE_CALL: cannot open device 0 (error %d)\n", rc); return rc; }
    const auto e2e_start = std::chrono::steady_clock::now();                 // main.cu:95
    auto lap = [last = e2e_start]() mutable {                                // --stats: wall time of each phase
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - last).count();
        last = now;
        return ms;
    };

    // image/camera configuration (main.cu:100-124)
    rtiow_camera_f32 cam32; rtiow_camera_f64 cam64;
    void* cam = precision == 64 ? (void*)&cam64 : (void*)&cam32;
    if (rtiow_host_camera(precision, opt.width, opt.height, opt.samples, opt.bounces, cam) != 0) {
        std::fputs("Error: invalid image size.\n", stderr);
        return 1;
    }
    check(h, rtiow_set_camera(h, cam));
    check(h, rtiow_set_scene_source(h, opt.scene_source));
    check(h, rtiow_set_schedule(h, opt.schedule, 0));

    // world creation (main.cu:142-321)
    const int slots = rtiow_host_scene_slots(opt.scene_id);
    std::vector<unsigned char> cr(elem * 4 * slots), af(elem * 4 * slots), ri(elem * slots);
    std::vector<int32_t> type(slots), valid(slots);
    rtiow_host_build_scene(opt.scene_id, precision, cr.data(), af.data(), ri.data(), type.data(), valid.data());
    check(h, rtiow_set_scene(h, slots, cr.data(), af.data(), ri.data(), type.data(), valid.data()));
    const double t_setup = lap();

    // device RNG (main.cu:324-330, rtweekend.h:49)
    check(h, rtiow_init_rng(h, 1227));
    const double t_rng = lap();

    // render, kernel-only timing (main.cu:332-345)
    float render_ms = 0;
    if (std::getenv("RTIOW_RENDER_TWICE")) {    // study knob (scripts/cold_process_study.sh): an untimed render first, so that the timed one finds the clocks up and the kernels loaded
        check(h, rtiow_render(h, opt.threads, &render_ms));
        check(h, rtiow_init_rng(h, 1227));
    }
    check(h, rtiow_render(h, opt.threads, &render_ms));
    std::printf("%15.8f,", (double)render_ms);
    std::fflush(stdout);
    const double t_render = lap();

    // .ppm output (main.cu:347-379)
    char name[256];
    rtiow_host_ppm_filename(precision, opt.scene_id, opt.width, opt.height, opt.samples, opt.bounces, opt.threads, name, sizeof name);
    // The writer's int(256 * clamp(c)) (main.cu:367, 374-376) runs on the device and one byte per channel comes back -- a quarter (fp32) or an
    // eighth (fp64) of the framebuffer -- unless a channel is NaN (never in these scenes): then the T framebuffer and the T writer, which
    // prints what the reference's build prints for it.
    const size_t nlev = 3 * (size_t)opt.width * opt.height;
    const std::unique_ptr<unsigned char[]> lev(new unsigned char[nlev]);      // not zero-filled: every byte is read back
    uint64_t nan_channels = 0;
    check(h, rtiow_read_levels(h, lev.get(), nlev, &nan_channels));
    std::unique_ptr<unsigned char[]> rgb;
    if (nan_channels != 0) {
        const size_t rgb_bytes = elem * nlev;
        rgb.reset(new unsigned char[rgb_bytes]);
        check(h, rtiow_read_framebuffer(h, rgb.get(), rgb_bytes));
    }
    const double t_read = lap();
    // The image is on the host: the device side is released (main.cu:384-391) on a thread of its own WHILE the file is written
    // (hipFree of ~250 MB of buffers takes 1.6-2.2 ms; the reference frees after the write, the order carries no meaning).
    rtiow_stats st;
    std::memset(&st, 0, sizeof st);
    rtiow_get_stats(h, &st);
    int destroy_rc = 0;
    double t_destroy = 0;
    std::thread releaser([&]() {
        const auto t0 = std::chrono::steady_clock::now();
        destroy_rc = rtiow_destroy(h);
        t_destroy = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    });
    // the thread is joined on EVERY way out of the write (a joinable std::thread's destructor terminates the process):
    // a writer that throws -- bad_alloc on its format buffers -- counts as a failed write
    int wrc = -1;
    try {
        wrc = nan_channels == 0 ? rtiow_host_write_ppm_levels(name, opt.width, opt.height, lev.get(), opt.binary_ppm ? 1 : 0)
            : (opt.binary_ppm ? rtiow_host_write_ppm_binary(name, precision, opt.width, opt.height, rgb.get())
                              : rtiow_host_write_ppm(name, precision, opt.width, opt.height, rgb.get()));
    } catch (...) { wrc = -1; }
    const double t_write = lap();
    releaser.join();
    // both results are reported: a failed release is not hidden behind a failed write
    if (destroy_rc != 0) std::fprintf(stderr, "HIP_SAFE_CALL: releasing the device failed (error %d)\n", destroy_rc);
    if (wrc != 0) {
        std::fprintf(stderr, "Error: Could not open file for writing: %s\n", name);
        return -1;
    }
    if (destroy_rc != 0) return destroy_rc;
    (void)lap();
    const auto e2e_stop = std::chrono::steady_clock::now();                  // main.cu:394
    const double e2e_ms = std::chrono::duration<double, std::milli>(e2e_stop - e2e_start).count();
    std::printf("%15.8f\n", e2e_ms);

    if (opt.stats) {   // extra metrics go to stderr so that stdout stays the reference's CSV fragment
        const double rays = (double)opt.width * opt.height * opt.samples;
        std::fprintf(stderr,
                     "{\"mrays_per_s\": %.3f, \"render_ms\": %.6f, \"rng_init_ms\": %.6f, \"spheres\": %d, \"block\": [%d, %d], "
                     "\"vgprs\": %d, \"lds_bytes\": %d, \"scene_source\": \"%s\", \"solo_waves\": %d, \"scene_prepare_ms\": %.3f, "
                     "\"launch_ms\": {\"prepass\": %.4f, \"main\": %.4f, \"place\": %.4f}, \"clock_mhz\": {\"nominal\": %d, \"prepass\": %.0f, \"main\": %.0f, \"main_wave0_ms\": %.3f}, "
                     "\"wall_ms\": {\"setup\": %.3f, \"rng_init\": %.3f, \"render\": %.3f, \"readback\": %.3f, \"ppm_write\": %.3f, \"destroy\": %.3f, \"end_to_end\": %.3f}, \"destroy_overlaps\": \"ppm_write\"}\n",
                     render_ms > 0 ? rays / render_ms / 1e3 : 0.0, (double)render_ms, st.rng_init_ms, st.num_spheres,
                     st.block_x, st.block_y, st.vgprs, st.lds_bytes, st.scene_source == RTIOW_SCENE_GRID ? "grid" : (st.scene_source == RTIOW_SCENE_SCALAR ? "scalar" : "lds"), st.solo_waves, st.scene_prepare_ms,
                     st.prepass_ms, st.main_ms, st.place_ms, st.clock_mhz, st.prepass_clock_mhz, st.main_clock_mhz, st.main_wave0_ms,
                     t_setup, t_rng, t_render, t_read, t_write, t_destroy, e2e_ms);
    }
    return 0;
}
