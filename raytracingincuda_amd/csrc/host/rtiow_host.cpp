// rtiow_host.cpp -- host half of the drop-in (no GPU code): scene tables, camera, PPM.
// Mirrors /root/reference/src/GlobalFloatCUDAInOneWeekend/main.cu (and the GlobalDouble twin);
// citations are relative to that directory.  Built with -ffp-contract=off: the reference's
// host code is compiled by g++ for x86-64 without FMA.
#include "rtiow_host.h"

#include <fcntl.h>
#include <unistd.h>
#include <sys/types.h>

#include <atomic>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

// glibc's rand() with the default seed (the reference never calls srand): additive feedback
// x[k] = x[k-3] + x[k-31] over 32-bit words, output x[k] >> 1, first 310 outputs dropped.
class LibcRandStream {
  public:
    LibcRandStream() {
        hist_.resize(34);
        int64_t w = 1;
        hist_[0] = 1;
        for (int k = 1; k < 31; ++k) {
            w = (16807 * w) % 2147483647;                 // minimal-standard LCG seeds the table
            hist_[k] = (uint32_t)w;
        }
        for (int k = 31; k < 34; ++k) hist_[k] = hist_[k - 31];
        for (int k = 34; k < 344; ++k) hist_.push_back(hist_[k - 31] + hist_[k - 3]);
    }
    int next() {
        const size_t k = hist_.size();
        const uint32_t x = hist_[k - 31] + hist_[k - 3];
        hist_.push_back(x);
        if (hist_.size() > 4096) hist_.erase(hist_.begin(), hist_.begin() + 2048);
        return (int)(x >> 1);
    }
  private:
    std::vector<uint32_t> hist_;
};

template <class T> struct HostRand;
template <> struct HostRand<float> {   // rtweekend.h:22-25
    static float draw(LibcRandStream& g) { return g.next() / (2147483647 + 1.0f); }
};
template <> struct HostRand<double> {  // GlobalDouble rtweekend.h:22-25
    static double draw(LibcRandStream& g) { return g.next() / (2147483647 + 1.0); }
};
template <class T> T draw_range(LibcRandStream& g, T lo, T hi) {   // rtweekend.h:27-30
    const T r = HostRand<T>::draw(g);
    return lo + (hi - lo) * r;
}

struct GridRange { int a_lo, a_hi, b_lo, b_hi; };
GridRange grid_of(int scene_id) {
    if (scene_id == 1) return {-11, 11, -11, 11};      // main.cu:163-164
    if (scene_id == 2) return {5, 11, 5, 11};          // main.cu:210-211
    return {-11, 0, -11, 0};                           // main.cu:253-254 (default:)
}

template <class T>
int build_scene_t(int scene_id, T* cr, T* af, T* ri, int32_t* type, int32_t* valid) {
    const GridRange gr = grid_of(scene_id);
    const int nb = gr.b_hi - gr.b_lo;
    const int slots = 1 + (gr.a_hi - gr.a_lo) * nb + 3;
    for (int s = 0; s < slots; ++s) {
        for (int k = 0; k < 4; ++k) { cr[4 * s + k] = 0; af[4 * s + k] = 0; }
        ri[s] = 0; type[s] = 0; valid[s] = 0;
    }
    auto put = [&](int s, T x, T y, T z, T r, int ty, T ar, T ag, T ab, T fuzz, T index) {
        cr[4 * s] = x; cr[4 * s + 1] = y; cr[4 * s + 2] = z; cr[4 * s + 3] = r;
        af[4 * s] = ar; af[4 * s + 1] = ag; af[4 * s + 2] = ab; af[4 * s + 3] = fuzz;
        ri[s] = index; type[s] = ty; valid[s] = 1;
    };
    LibcRandStream g;
    put(0, 0, -1000, 0, 1000, RTIOW_LAMBERTIAN, (T)0.5, (T)0.5, (T)0.5, 0, 0);       // main.cu:158-159
    for (int a = gr.a_lo; a < gr.a_hi; ++a) {
        for (int b = gr.b_lo; b < gr.b_hi; ++b) {
            const T choose = HostRand<T>::draw(g);                                   // main.cu:165
            // main.cu:166: constructor arguments are evaluated last-to-first by g++.
            const T for_z = HostRand<T>::draw(g);
            const T for_x = HostRand<T>::draw(g);
            const T x = (T)(a + 0.9 * for_x), y = (T)0.2, z = (T)(b + 0.9 * for_z);
            const T ex = x - (T)4, ey = y - (T)0.2, ez = z - (T)0;
            const T dist = std::sqrt(ex * ex + ey * ey + ez * ez);
            if (!(dist > 0.9)) continue;                                             // main.cu:168
            const int slot = (a - gr.a_lo) * nb + (b - gr.b_lo) + 1;                 // main.cu:172
            if (choose < 0.8) {                                                      // main.cu:175-179
                T rhs[3], lhs[3];
                for (int k = 2; k >= 0; --k) rhs[k] = HostRand<T>::draw(g);         // right operand first,
                for (int k = 2; k >= 0; --k) lhs[k] = HostRand<T>::draw(g);         // each as z, y, x
                put(slot, x, y, z, (T)0.2, RTIOW_LAMBERTIAN, lhs[0] * rhs[0], lhs[1] * rhs[1], lhs[2] * rhs[2], 0, 0);
            } else if (choose < 0.95) {                                              // main.cu:181-186
                T alb[3];
                for (int k = 2; k >= 0; --k) alb[k] = draw_range<T>(g, (T)0.5, (T)1.0);
                T fuzz = draw_range<T>(g, (T)0.0, (T)0.5);
                if (!(fuzz < (T)1)) fuzz = 1;                                        // material.h:29-30
                put(slot, x, y, z, (T)0.2, RTIOW_METAL, alb[0], alb[1], alb[2], fuzz, 0);
            } else {                                                                 // main.cu:188-191
                put(slot, x, y, z, (T)0.2, RTIOW_DIELECTRIC, 0, 0, 0, 0, (T)1.5);
            }
        }
    }
    const int s = slots - 3;                                                         // main.cu:287-296
    put(s, 0, 1, 0, 1, RTIOW_DIELECTRIC, 0, 0, 0, 0, (T)1.5);
    put(s + 1, -4, 1, 0, 1, RTIOW_LAMBERTIAN, (T)0.4, (T)0.2, (T)0.1, 0, 0);
    put(s + 2, 4, 1, 0, 1, RTIOW_METAL, (T)0.7, (T)0.6, (T)0.5, 0, 0);
    return slots;
}

template <class T> struct P3 { T e[3]; };
template <class T> P3<T> sub(P3<T> a, P3<T> b) { return {{a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]}}; }
template <class T> P3<T> add(P3<T> a, P3<T> b) { return {{a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]}}; }
template <class T> P3<T> mul(T t, P3<T> a) { return {{t * a.e[0], t * a.e[1], t * a.e[2]}}; }
template <class T> P3<T> over(P3<T> a, T t) { return mul((T)1 / t, a); }               // vec3.h:89-91
template <class T> P3<T> crossp(P3<T> u, P3<T> v) {                                    // vec3.h:99-103
    return {{u.e[1] * v.e[2] - u.e[2] * v.e[1], u.e[2] * v.e[0] - u.e[0] * v.e[2], u.e[0] * v.e[1] - u.e[1] * v.e[0]}};
}
template <class T> P3<T> normalize(P3<T> a) {                                          // vec3.h:105-107
    const T len = std::sqrt(a.e[0] * a.e[0] + a.e[1] * a.e[1] + a.e[2] * a.e[2]);
    return over(a, len);
}

template <class T, class CAM>
void camera_t(int width, int height, int samples, int bounces, CAM* cam) {
    const T pi = (T)3.1415926535897932385L;                     // rtweekend.h:14
    const T vfov = 20, focus_dist = (T)10.0, defocus_angle = (T)0.6;   // main.cu:114-121
    const P3<T> lookfrom = {{13, 2, 3}}, lookat = {{0, 0, 0}}, vup = {{0, 1, 0}};
    cam->img_width = width; cam->img_height = height;
    cam->samples_per_pixel = samples; cam->max_depth = bounces;
    cam->pixel_samples_scale = (T)1 / samples;                  // camera.h:34
    const T theta = vfov * pi / (T)180;                         // camera.h:42, rtweekend.h:18-20
    const T hh = std::tan(theta / 2);                           // :43
    const T vp_h = (T)2 * hh * focus_dist;                      // :44
    const T vp_w = vp_h * ((T)width / height);                  // :45
    const P3<T> w = normalize(sub(lookfrom, lookat));           // :48-50
    const P3<T> u = normalize(crossp(vup, w));
    const P3<T> v = crossp(w, u);
    const P3<T> vp_u = mul(vp_w, u);                            // :53-54
    const P3<T> vp_v = mul(vp_h, P3<T>{{-v.e[0], -v.e[1], -v.e[2]}});
    const P3<T> du = over(vp_u, (T)width), dv = over(vp_v, (T)height);   // :57-58
    const P3<T> ul = sub(sub(sub(lookfrom, mul(focus_dist, w)), over(vp_u, (T)2)), over(vp_v, (T)2));   // :61
    const P3<T> p00 = add(ul, mul((T)0.5, add(du, dv)));        // :62
    const T radius = focus_dist * std::tan((defocus_angle / 2) * pi / (T)180);    // :65
    const P3<T> ddu = mul(radius, u), ddv = mul(radius, v);     // :66-67
    for (int k = 0; k < 3; ++k) {
        cam->center[k] = lookfrom.e[k]; cam->pixel00_loc[k] = p00.e[k];
        cam->pixel_delta_u[k] = du.e[k]; cam->pixel_delta_v[k] = dv.e[k];
        cam->defocus_disk_u[k] = ddu.e[k]; cam->defocus_disk_v[k] = ddv.e[k];
    }
    cam->defocus_angle = defocus_angle;
}

template <class T> inline int to_level(T c) {                   // main.cu:367,374-376; interval.h:25-29
    const T lo = (T)0.000, hi = (T)0.999;
    // clamp() passes a NaN through and int(NaN) is undefined in the reference; its x86 build yields
    // INT_MIN ("-2147483648" in the file).  Same text here, without the undefined conversion.
    if (!(c == c)) return INT_MIN;
    const T cl = c < lo ? lo : (c > hi ? hi : c);
    return (int)(256 * cl);
}

// Levels 0..255 as text, four bytes each ("255 " style: digits then blank padding is NOT used -- the entry holds
// the digits and its length), so the writer copies a table entry per channel instead of dividing.
struct LevelText { char s[4]; unsigned char n; };
const LevelText* level_table() {
    static LevelText tab[256];
    static bool ready = false;
    if (!ready) {
        for (int v = 0; v < 256; ++v) {
            char buf[8];
            const int n = std::snprintf(buf, sizeof buf, "%d", v);
            std::memcpy(tab[v].s, buf, (size_t)n);
            tab[v].n = (unsigned char)n;
        }
        ready = true;
    }
    return tab;
}

// Text of the pixels [p0, p1): "r g b\n" per pixel (main.cu:372-377).  Returns the end of what was written, or
// nullptr when the buffer [w, w_end) may not hold the next pixel (36 bytes: "-2147483648 " is the longest channel).
template <class T>
char* format_pixels(const T* rgb, size_t p0, size_t p1, char* w, const char* w_end) {
    const LevelText* tab = level_table();
    for (size_t p = p0; p < p1; ++p) {
        if (w + 36 > w_end) return nullptr;
        for (int k = 0; k < 3; ++k) {
            const int level = to_level<T>(rgb[3 * p + k]);
            if (level >= 0) {                                       // 0..255 by construction (clamp to 0.999)
                const LevelText& t = tab[level & 255];
                std::memcpy(w, t.s, 4);                             // fixed-size copy, then advance by the real length
                w += t.n;
            } else {                                                // only the NaN level is negative; unsigned negation is defined for INT_MIN
                unsigned v = 0u - (unsigned)level;
                char digits[16];
                int n = 0;
                *w++ = '-';
                do { digits[n++] = (char)('0' + v % 10); v /= 10; } while (v);
                while (n) *w++ = digits[--n];
            }
            *w++ = k == 2 ? '\n' : ' ';
        }
    }
    return w;
}

// One contiguous range of pixels as text.  The buffer is sized for levels 0..255 (12 bytes per pixel) and not
// initialised (a zero-filled 36 bytes per pixel cost more than the formatting); a range with a NaN level is
// formatted again into the long form.
struct TextPart {
    std::unique_ptr<char[]> buf;
    size_t n = 0;
    template <class T> void format(const T* rgb, size_t p0, size_t p1) {
        for (size_t per_pixel : {(size_t)12, (size_t)36}) {
            const size_t cap = (p1 - p0) * per_pixel + 64;
            buf.reset(new char[cap]);
            if (const char* e = format_pixels<T>(rgb, p0, p1, buf.get(), buf.get() + cap)) { n = (size_t)(e - buf.get()); return; }
        }
    }
};

// The whole file, header first, through `sink(data, bytes)` in file order.  Up to 8 threads format contiguous
// pixel ranges; a range is handed to the sink as soon as it and every range before it are done, so writing the
// file overlaps the formatting of its rest (the text writer is the largest part of the end-to-end time after the
// render: 12 ms of 33 at 1280x768 with one thread and a zero-filled buffer).
template <class T, class Sink>
void format_ppm_stream(int width, int height, const T* rgb, Sink sink) {
    char head[64];
    const int hn = std::snprintf(head, sizeof head, "P3\n%d %d\n255\n", width, height);
    sink(head, (size_t)hn);
    const size_t npix = (size_t)width * height;
    (void)level_table();                                            // built before the threads start
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 8) nt = 8;
    if (nt < 1 || npix < 65536) nt = 1;
    std::vector<TextPart> parts(nt);
    std::vector<std::thread> th;
    for (unsigned k = 1; k < nt; ++k)
        th.emplace_back([&parts, rgb, npix, nt, k]() { parts[k].template format<T>(rgb, npix * k / nt, npix * (k + 1) / nt); });
    parts[0].template format<T>(rgb, 0, npix / nt);                 // the calling thread takes the first range
    sink(parts[0].buf.get(), parts[0].n);
    for (unsigned k = 1; k < nt; ++k) {
        th[k - 1].join();
        sink(parts[k].buf.get(), parts[k].n);
        parts[k].buf.reset();
    }
}

template <class T>
void format_ppm_t(int width, int height, const T* rgb, std::string& out) {
    out.clear();
    out.reserve((size_t)width * height * 12 + 64);
    format_ppm_stream<T>(width, height, rgb, [&out](const char* d, size_t n) { out.append(d, n); });
}

}  // namespace

extern "C" {

int rtiow_host_scene_slots(int scene_id) {
    const GridRange gr = grid_of(scene_id);
    return 1 + (gr.a_hi - gr.a_lo) * (gr.b_hi - gr.b_lo) + 3;
}

int rtiow_host_build_scene(int scene_id, int precision, void* center_radius, void* albedo_fuzz,
                           void* refraction_index, int32_t* type, int32_t* valid) {
    if (!center_radius || !albedo_fuzz || !refraction_index || !type || !valid) return RTIOW_E_BADARG;
    if (precision == 32) return build_scene_t<float>(scene_id, (float*)center_radius, (float*)albedo_fuzz, (float*)refraction_index, type, valid);
    if (precision == 64) return build_scene_t<double>(scene_id, (double*)center_radius, (double*)albedo_fuzz, (double*)refraction_index, type, valid);
    return RTIOW_E_BADARG;
}

int rtiow_host_camera(int precision, int width, int height, int samples, int bounces, void* out) {
    if (!out || width <= 0 || height <= 0) return RTIOW_E_BADARG;
    if (precision == 32) { camera_t<float>(width, height, samples, bounces, (rtiow_camera_f32*)out); return 0; }
    if (precision == 64) { camera_t<double>(width, height, samples, bounces, (rtiow_camera_f64*)out); return 0; }
    return RTIOW_E_BADARG;
}

int rtiow_host_ppm_filename(int precision, int scene_id, int width, int height, int samples,
                            int bounces, int threads, char* out, size_t cap) {
    if (!out || (precision != 32 && precision != 64)) return RTIOW_E_BADARG;
    const int n = std::snprintf(out, cap, "%s_scene%d_%dx%d_%dsamples_%dbounces_%dthreadsPerBlockRow.ppm",
                                precision == 32 ? "global_float" : "global_double",
                                scene_id, width, height, samples, bounces, threads);
    return (n < 0 || (size_t)n >= cap) ? RTIOW_E_BADARG : 0;
}

int rtiow_host_format_ppm(int precision, int width, int height, const void* rgb, char* out, size_t cap, size_t* len) {
    if (!rgb || width <= 0 || height <= 0) return RTIOW_E_BADARG;
    std::string s;
    if (precision == 32) format_ppm_t<float>(width, height, (const float*)rgb, s);
    else if (precision == 64) format_ppm_t<double>(width, height, (const double*)rgb, s);
    else return RTIOW_E_BADARG;
    if (len) *len = s.size();
    if (out) { if (cap < s.size()) return RTIOW_E_BADARG; std::memcpy(out, s.data(), s.size()); }
    return 0;
}

int rtiow_host_write_ppm(const char* path, int precision, int width, int height, const void* rgb) {
    if (!path || !rgb || width <= 0 || height <= 0) return RTIOW_E_BADARG;
    if (precision != 32 && precision != 64) return RTIOW_E_BADARG;
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return RTIOW_E_STATE;
    std::setvbuf(f, nullptr, _IONBF, 0);                            // whole ranges go straight to write(2)
    bool ok = true;
    auto sink = [&ok, f](const char* d, size_t n) { ok = ok && std::fwrite(d, 1, n, f) == n; };
    if (precision == 32) format_ppm_stream<float>(width, height, (const float*)rgb, sink);
    else format_ppm_stream<double>(width, height, (const double*)rgb, sink);
    return (std::fclose(f) == 0 && ok) ? 0 : RTIOW_E_STATE;
}

// Binary twin of the P3 writer: "P6\nW H\n255\n" + 3 bytes per pixel, same quantisation
// (the text writer dominates end-to-end time minus render).
int rtiow_host_write_ppm_binary(const char* path, int precision, int width, int height, const void* rgb) {
    if (!path || !rgb || width <= 0 || height <= 0 || (precision != 32 && precision != 64)) return RTIOW_E_BADARG;
    char head[64];
    const int hn = std::snprintf(head, sizeof head, "P6\n%d %d\n255\n", width, height);
    const size_t n = (size_t)width * height * 3;
    std::string s(head, (size_t)hn);
    s.resize((size_t)hn + n);
    unsigned char* out = reinterpret_cast<unsigned char*>(&s[(size_t)hn]);
    if (precision == 32) { const float* v = (const float*)rgb; for (size_t k = 0; k < n; ++k) out[k] = (unsigned char)to_level<float>(v[k]); }
    else { const double* v = (const double*)rgb; for (size_t k = 0; k < n; ++k) out[k] = (unsigned char)to_level<double>(v[k]); }
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return RTIOW_E_STATE;
    const bool ok = std::fwrite(s.data(), 1, s.size(), f) == s.size();
    return (std::fclose(f) == 0 && ok) ? 0 : RTIOW_E_STATE;
}

long long rtiow_host_levels(int precision, int width, int height, const void* rgb, unsigned char* levels) {
    if (!rgb || !levels || width <= 0 || height <= 0 || (precision != 32 && precision != 64)) return -1;
    const size_t n = (size_t)width * height * 3;
    long long nans = 0;
    for (size_t k = 0; k < n; ++k) {
        const int level = precision == 32 ? to_level<float>(((const float*)rgb)[k]) : to_level<double>(((const double*)rgb)[k]);
        if (level < 0) { ++nans; levels[k] = 0; } else levels[k] = (unsigned char)level;
    }
    return nans;
}

int rtiow_host_write_ppm_levels(const char* path, int width, int height, const unsigned char* levels, int binary) {
    if (!path || !levels || width <= 0 || height <= 0) return RTIOW_E_BADARG;
    const size_t npix = (size_t)width * height;
    char head[64];
    const int hn = std::snprintf(head, sizeof head, "%s\n%d %d\n255\n", binary ? "P6" : "P3", width, height);
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);        // header and body go out through pwrite
    if (fd < 0) return RTIOW_E_STATE;
    auto write_all = [fd](const char* d, size_t n, off_t at) {
        while (n) {
            const ssize_t w = ::pwrite(fd, d, n, at);
            if (w <= 0) return false;
            d += w; n -= (size_t)w; at += w;
        }
        return true;
    };
    bool ok = write_all(head, (size_t)hn, 0);
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 1 || npix < 65536) nt = 1;
    if (binary) {
        // the levels ARE the file's body: ranges of it written by nt threads (page-cache copies scale with the threads)
        std::vector<std::thread> th;
        std::vector<char> good(nt, 1);
        const size_t n = npix * 3;
        for (unsigned k = 0; k < nt; ++k) {
            const size_t a = n * k / nt, b = n * (k + 1) / nt;
            auto job = [&, a, b, k]() { good[k] = write_all((const char*)levels + a, b - a, (off_t)hn + (off_t)a); };
            if (k + 1 < nt) th.emplace_back(job); else job();
        }
        for (auto& t : th) t.join();
        for (char g : good) ok = ok && g;
    } else {
        // text: the frame is cut into ranges; worker threads format them ("r g b\n", main.cu:372-377) while this thread writes every
        // finished range in file order.  The file system's buffered write is the floor (23 MB of text at 1080p: 6.5-9 ms on the GPU
        // boxes' overlay; several threads writing one file are serialised by its lock, and formatting through a shared mapping of the
        // file took twice as long: profiles/r04/e2e_*.log), so the formatting -- 1-2 ms on 15 threads -- hides behind it.
        const LevelText* tab = level_table();
        const unsigned workers = nt > 1 ? nt - 1 : 1;
        const unsigned nr = nt > 1 ? 4 * workers : 1;                 // ranges: four rounds per worker, so that the first write starts early
        std::vector<std::unique_ptr<char[]>> buf(nr);
        std::vector<size_t> len(nr, 0);
        auto format = [&](unsigned k) {
            const size_t p0 = npix * k / nr, p1 = npix * (k + 1) / nr;
            buf[k].reset(new char[(p1 - p0) * 12 + 16]);
            char* w = buf[k].get();
            const unsigned char* v = levels + 3 * p0;
            for (size_t p = p0; p < p1; ++p, v += 3) {
                const LevelText& r = tab[v[0]]; std::memcpy(w, r.s, 4); w += r.n; *w++ = ' ';
                const LevelText& g = tab[v[1]]; std::memcpy(w, g.s, 4); w += g.n; *w++ = ' ';
                const LevelText& b = tab[v[2]]; std::memcpy(w, b.s, 4); w += b.n; *w++ = '\n';
            }
            len[k] = (size_t)(w - buf[k].get());
        };
        off_t pos = hn;
        if (nt == 1) {
            format(0);
            ok = ok && write_all(buf[0].get(), len[0], pos);
        } else {
            std::vector<std::atomic<int>> done(nr);
            for (auto& d : done) d.store(0);
            std::vector<std::thread> th;
            for (unsigned t = 0; t < workers; ++t)
                th.emplace_back([&, t]() { for (unsigned k = t; k < nr; k += workers) { format(k); done[k].store(1, std::memory_order_release); } });
            for (unsigned k = 0; k < nr; ++k) {
                while (!done[k].load(std::memory_order_acquire)) std::this_thread::yield();
                ok = ok && write_all(buf[k].get(), len[k], pos);
                pos += (off_t)len[k];
                buf[k].reset();
            }
            for (auto& t : th) t.join();
        }
    }
    return (::close(fd) == 0 && ok) ? 0 : RTIOW_E_STATE;
}

int rtiow_host_shard_rows(int height, int rank, int nranks, int strip_rows, int32_t* rows_out) {
    if (height <= 0 || nranks < 1 || rank < 0 || rank >= nranks || strip_rows < 1) return RTIOW_E_BADARG;
    const int nstrips = (height + strip_rows - 1) / strip_rows;
    int n = 0;
    for (int s = rank; s < nstrips; s += nranks)
        for (int r = s * strip_rows; r < (s + 1) * strip_rows && r < height; ++r, ++n)
            if (rows_out) rows_out[n] = r;
    return n;
}

int rtiow_host_place_rows(int precision, int width, int height, int rank, int nranks, int strip_rows,
                          const void* local_rgb, void* full_rgb) {
    if (!local_rgb || !full_rgb || width <= 0 || height <= 0 || nranks < 1 || rank < 0 || rank >= nranks || strip_rows < 1) return RTIOW_E_BADARG;
    if (precision != 32 && precision != 64) return RTIOW_E_BADARG;
    const size_t row_bytes = (size_t)width * 3 * (precision == 64 ? 8 : 4);
    const int nstrips = (height + strip_rows - 1) / strip_rows;
    size_t local_row = 0;
    for (int s = rank; s < nstrips; s += nranks)
        for (int r = s * strip_rows; r < (s + 1) * strip_rows && r < height; ++r, ++local_row)
            std::memcpy((char*)full_rgb + (size_t)r * row_bytes, (const char*)local_rgb + local_row * row_bytes, row_bytes);
    return 0;
}

}  // extern "C"
