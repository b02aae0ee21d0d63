// rtiow_oracle.cpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// This file is the checker for the HIP render path, never the product: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
// Nothing under raytracingincuda_amd/ links, imports or calls it.
//
// It restates, on the CPU, the per-pixel `render` path of the reference
//   src/GlobalFloatCUDAInOneWeekend  (fp32)  and  src/GlobalDoubleCUDAInOneWeekend (fp64)
// ("CUDA semantics": per-pixel XORWOW stream, iterative ray_color, sky from the PRIMARY ray)
// and, separately, the serial program src/InOneWeekend ("serial semantics": fp64, one
// std::rand() stream, recursive ray_color, sky from the current ray).
// All file:line citations are relative to /root/reference/.
//
// PINNING (see DESIGN.md "Oracle"):
//  * serial semantics : byte-identical P3 output to the reference's own serial sources
//    compiled with g++ (oracle/_ref/, tests/test_oracle_vs_ref.py, tests/golden/serial_*.json).
//    This pins every function the two semantics share (hit_sphere, the three scatters,
//    reflect/refract/Schlick, camera::initialize, scene generation, the P3 writer).
//  * CUDA semantics   : the CUDA sources cannot be built or run here (no nvcc, no cuRAND).
//    The deltas to the serial semantics are restated line by line below; the device RNG
//    (cuRAND XORWOW, a third-party library absent from /root/reference) is restated from its
//    published algorithm and checked against rocRAND known answers (tests/golden/xorwow_kat.json).
//    Bit-level equality with cuRAND / nvcc's FMA contraction choices is "parity unpinned".
//
// FLOATING-POINT CONTRACT of the CUDA-semantics path (shared with the HIP kernel, so that
// kernel == oracle bit for bit): every +,-,*,/ and sqrt is IEEE-754 correctly rounded in T;
// a*b+c patterns that nvcc (-fmad=true default) may contract are written as explicit fma()
// at the places listed in DESIGN.md §"Canonical operation sequence"; nothing else is fused
// (build with -ffp-contract=off).  powf(x,5) (material.h:65) is evaluated as the float
// product chain ((x*x)*(x*x))*x.  The serial-semantics path uses NO fma (g++ -O3 on x86-64
// without -mfma emits none) and std::pow, exactly like the reference build.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

namespace {

// =====================================================================================
// glibc rand() restatement (TYPE_3 additive feedback generator, unseeded => seed 1).
// The reference never calls srand (SURVEY §0 finding 3), so scene tables are fixed.
// Checked against the C library's own rand() in tests/test_oracle_host.py.
// =====================================================================================
struct GlibcRand {
    uint32_t r[34];
    int f, b;  // front / back indices into the 31-word ring r[3..33]
    explicit GlibcRand(uint32_t seed = 1) { reset(seed); }
    void reset(uint32_t seed) {
        int32_t tbl[31];
        if (seed == 0) seed = 1;
        tbl[0] = (int32_t)seed;
        for (int i = 1; i < 31; ++i) {
            int64_t hi = tbl[i - 1] / 127773, lo = tbl[i - 1] % 127773;
            int64_t word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            tbl[i] = (int32_t)word;
        }
        for (int i = 0; i < 31; ++i) r[i] = (uint32_t)tbl[i];
        f = 3; b = 0;
        for (int i = 0; i < 310; ++i) next();
    }
    int32_t next() {
        r[f] += r[b];
        uint32_t result = r[f] >> 1;
        if (++f >= 31) f = 0;
        if (++b >= 31) b = 0;
        return (int32_t)result;
    }
};

// rtweekend.h:22-25 (float: rand()/(RAND_MAX+1.0f)); GlobalDouble rtweekend.h:22-25 and
// src/InOneWeekend/rtweekend.h:37-40 (double: rand()/(RAND_MAX+1.0)).
template <class T> inline T host_random(GlibcRand& g);
template <> inline float host_random<float>(GlibcRand& g) { return (float)g.next() / (2147483647 + 1.0f); }
template <> inline double host_random<double>(GlibcRand& g) { return g.next() / (2147483647 + 1.0); }
// rtweekend.h:27-30
template <class T> inline T host_random(GlibcRand& g, T mn, T mx) { T r = host_random<T>(g); return mn + (mx - mn) * r; }

// =====================================================================================
// XORWOW (cuRAND's curandStateXORWOW_t; third-party, not vendored in the reference).
// Call sites: rtweekend.h:34,49; vec3.h:119-121; camera.h:145-146.
// Published algorithm: Marsaglia xorwow, 5x32-bit xorshift state + Weyl counter d.
// curand_init(seed, subsequence, offset): seed scrambling, then skip subsequence*2^67
// steps, then offset steps.
// =====================================================================================
struct Xorwow { uint32_t v[5]; uint32_t d; };

inline uint32_t xorwow_next(Xorwow& s) {
    uint32_t t = s.v[0] ^ (s.v[0] >> 2);
    s.v[0] = s.v[1]; s.v[1] = s.v[2]; s.v[2] = s.v[3]; s.v[3] = s.v[4];
    s.v[4] = (s.v[4] ^ (s.v[4] << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v[4] + s.d;
}

// 160x160 GF(2) matrix: col[i] = image of basis bit i (5 words).
struct Mat160 { uint32_t col[160][5]; };

inline void mat_apply(const Mat160& m, const uint32_t in[5], uint32_t out[5]) {
    uint32_t acc[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 5; ++w)
        for (int bit = 0; bit < 32; ++bit)
            if ((in[w] >> bit) & 1u) {
                const uint32_t* c = m.col[w * 32 + bit];
                for (int k = 0; k < 5; ++k) acc[k] ^= c[k];
            }
    for (int k = 0; k < 5; ++k) out[k] = acc[k];
}

inline void mat_square(const Mat160& m, Mat160& out) {
    for (int i = 0; i < 160; ++i) mat_apply(m, m.col[i], out.col[i]);
}

struct XorwowJump {
    // jump[b] = L^(2^(67+b)) : advance by 2^b subsequences.
    std::vector<Mat160> jump;
    // step[b] = L^(2^b) : advance by 2^b draws (offset).
    std::vector<Mat160> step;
    XorwowJump() {
        Mat160 one;
        for (int i = 0; i < 160; ++i) {
            Xorwow s; std::memset(&s, 0, sizeof s);
            s.v[i / 32] = 1u << (i % 32);
            xorwow_next(s);
            for (int k = 0; k < 5; ++k) one.col[i][k] = s.v[k];
        }
        Mat160 cur = one, nxt;
        for (int b = 0; b < 67 + 32; ++b) {
            if (b < 32) step.push_back(cur);
            if (b >= 67) jump.push_back(cur);
            mat_square(cur, nxt);
            cur = nxt;
        }
    }
};

const XorwowJump& jump_tables() { static XorwowJump j; return j; }

// salt 0 = cuRAND's published seed scrambling (what the reference runs: rtweekend.h:49);
// salt 1 = rocRAND's (rocrand_xorwow.h: same engine and 2^67 stride, different "arbitrary"
// constants) -- used ONLY to check engine + jump matrices against rocRAND known answers.
inline void xorwow_init(Xorwow& s, uint64_t seed, uint64_t subsequence, uint64_t offset, int salt = 0) {
    const uint32_t x0 = salt ? 0x2c7f967fu : 0xaad26b49u, x1 = salt ? 0xa03697cbu : 0xf7dcefddu;
    const uint32_t m0 = salt ? 1228688033u : 1099087573u, m1 = salt ? 2073658381u : 2591861531u;
    uint32_t s0 = (uint32_t)seed ^ x0;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ x1;
    uint32_t t0 = m0 * s0;
    uint32_t t1 = m1 * s1;
    s.d = 6615241u + t1 + t0;
    s.v[0] = 123456789u + t0;
    s.v[1] = 362436069u ^ t0;
    s.v[2] = 521288629u + t1;
    s.v[3] = 88675123u ^ t1;
    s.v[4] = 5783321u + t0;
    const XorwowJump& J = jump_tables();
    for (int b = 0; b < 32 && (subsequence >> b); ++b)
        if ((subsequence >> b) & 1u) { uint32_t o[5]; mat_apply(J.jump[b], s.v, o); std::memcpy(s.v, o, sizeof o); }
    // d is a Weyl counter: +362437 per draw; 2^67*k draws add 0 mod 2^32.
    for (int b = 0; b < 32 && (offset >> b); ++b)
        if ((offset >> b) & 1u) { uint32_t o[5]; mat_apply(J.step[b], s.v, o); std::memcpy(s.v, o, sizeof o); }
    s.d += 362437u * (uint32_t)offset;
}

// curand_uniform: (0,1], x*2^-32 + 2^-33 in float.  curand_uniform_double (XORWOW): two
// draws x,y -> z = x ^ (y << 21) (53 bits), z*2^-53 + 2^-54.
template <class T> inline T device_uniform(Xorwow& s);
template <> inline float device_uniform<float>(Xorwow& s) {
    uint32_t x = xorwow_next(s);
    return std::fma((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}
template <> inline double device_uniform<double>(Xorwow& s) {
    uint32_t x = xorwow_next(s);
    uint32_t y = xorwow_next(s);
    uint64_t z = (uint64_t)x ^ ((uint64_t)y << (53 - 32));
    return std::fma((double)z, 1.1102230246251565e-16, 5.5511151231257827e-17);
}

// =====================================================================================
// vec3 (vec3.h:7-107).  FUSED selects the CUDA-semantics contraction contract.
// =====================================================================================
template <class T> struct V3 { T x, y, z; };

template <bool FUSED, class T> inline T mad(T a, T b, T c) {  // a*b + c
    if (FUSED) return std::fma(a, b, c);
    return a * b + c;
}
template <class T> inline V3<T> operator+(V3<T> u, V3<T> v) { return {u.x + v.x, u.y + v.y, u.z + v.z}; }
template <class T> inline V3<T> operator-(V3<T> u, V3<T> v) { return {u.x - v.x, u.y - v.y, u.z - v.z}; }
template <class T> inline V3<T> operator-(V3<T> u) { return {-u.x, -u.y, -u.z}; }
template <class T> inline V3<T> operator*(V3<T> u, V3<T> v) { return {u.x * v.x, u.y * v.y, u.z * v.z}; }
template <class T> inline V3<T> scale(T t, V3<T> v) { return {t * v.x, t * v.y, t * v.z}; }  // vec3.h:81-83
// vec3.h:93-97: e0*e0' + e1*e1' + e2*e2'  (left to right; contracted as fma(z, fma(y, x*x)))
template <bool F, class T> inline T dot(V3<T> u, V3<T> v) { return mad<F>(u.z, v.z, mad<F>(u.y, v.y, u.x * v.x)); }
template <bool F, class T> inline T len2(V3<T> v) { return dot<F>(v, v); }  // vec3.h:44-46
// w + t*v  (ray::at ray.h:19-21 and every "vec + scalar*vec" in the path)
template <bool F, class T> inline V3<T> madd(T t, V3<T> v, V3<T> w) { return {mad<F>(t, v.x, w.x), mad<F>(t, v.y, w.y), mad<F>(t, v.z, w.z)}; }
// vec3.h:89-91: v / t == (1/t) * v
template <class T> inline V3<T> vdiv(V3<T> v, T t) { return scale((T)1 / t, v); }
// vec3.h:105-107
template <bool F, class T> inline V3<T> unit(V3<T> v) { return vdiv(v, (T)std::sqrt(len2<F>(v))); }
// vec3.h:99-103
template <class T> inline V3<T> cross(V3<T> u, V3<T> v) {
    return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
// vec3.h:129-131: v - 2*dot(v,n)*n
template <bool F, class T> inline V3<T> reflect(V3<T> v, V3<T> n) {
    T k = (T)2 * dot<F>(v, n);
    if (F) return madd<true>(-k, n, v);
    return v - scale(k, n);
}
// vec3.h:133-138
template <bool F, class T> inline V3<T> refract(V3<T> uv, V3<T> n, T eta) {
    T cos_theta = std::fmin(dot<F>(-uv, n), (T)1);
    V3<T> perp = scale(eta, F ? madd<true>(cos_theta, n, uv) : uv + scale(cos_theta, n));
    T k = -(T)std::sqrt(std::fabs((T)1 - len2<F>(perp)));
    if (F) return madd<true>(k, n, perp);
    return perp + scale(k, n);
}

// =====================================================================================
// Scene tables (main.cu:148-296) -- gcc argument-evaluation order made explicit.
// =====================================================================================
enum { LAMBERTIAN = 0, METAL = 1, DIELECTRIC = 2 };  // material.h:11-15

template <class T> struct Scene {
    int n = 0;
    std::vector<V3<T>> center, albedo;
    std::vector<T> radius, fuzz, ri;
    std::vector<int> type, valid;
    void resize(int k) {
        n = k; center.assign(k, {0, 0, 0}); albedo.assign(k, {0, 0, 0});
        radius.assign(k, 0); fuzz.assign(k, 0); ri.assign(k, 0); type.assign(k, 0); valid.assign(k, 0);
    }
    void set(int i, V3<T> c, T r, int ty, V3<T> alb, T fz, T idx) {
        center[i] = c; radius[i] = r; type[i] = ty; albedo[i] = alb; fuzz[i] = fz; ri[i] = idx; valid[i] = 1;
    }
};

// SERIAL selects src/InOneWeekend/main.cc:25-66 (same draws; objects simply appended).
template <class T> void build_scene(int scene_id, Scene<T>& sc, GlibcRand& g) {
    int a0, a1, b0, b1;
    if (scene_id == 1) { a0 = -11; a1 = 11; b0 = -11; b1 = 11; }       // main.cu:149-194
    else if (scene_id == 2) { a0 = 5; a1 = 11; b0 = 5; b1 = 11; }      // main.cu:196-240
    else { a0 = -11; a1 = 0; b0 = -11; b1 = 0; }                        // main.cu:241-284 (default:)
    const int nb = b1 - b0;
    sc.resize(1 + (a1 - a0) * nb + 3);
    sc.set(0, {0, -1000, 0}, 1000, LAMBERTIAN, {(T)0.5, (T)0.5, (T)0.5}, 0, 0);  // main.cu:158-159
    for (int a = a0; a < a1; ++a)
        for (int b = b0; b < b1; ++b) {
            T choose_mat = host_random<T>(g);                       // main.cu:165
            // main.cu:166 `point3 center(a+0.9*rf(), 0.2, b+0.9*rf())` : g++ evaluates the
            // constructor arguments right to left => z-term draw, then x-term draw.
            T zdraw = host_random<T>(g);
            T xdraw = host_random<T>(g);
            V3<T> c = {(T)(a + 0.9 * (double)xdraw), (T)0.2, (T)(b + 0.9 * (double)zdraw)};
            V3<T> d = c - V3<T>{4, (T)0.2, 0};
            T len = (T)std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);   // host code: no FMA
            if ((double)len > 0.9) {                                 // main.cu:168
                int i = (a - a0) * nb + (b - b0) + 1;                 // main.cu:172
                if ((double)choose_mat < 0.8) {
                    // main.cu:176 color::random()*color::random(): right operand first, each z,y,x
                    V3<T> R, L;
                    R.z = host_random<T>(g); R.y = host_random<T>(g); R.x = host_random<T>(g);
                    L.z = host_random<T>(g); L.y = host_random<T>(g); L.x = host_random<T>(g);
                    sc.set(i, c, (T)0.2, LAMBERTIAN, L * R, 0, 0);
                } else if ((double)choose_mat < 0.95) {
                    V3<T> A;                                          // main.cu:182 random(0.5,1.0): z,y,x
                    A.z = host_random<T>(g, (T)0.5, (T)1.0);
                    A.y = host_random<T>(g, (T)0.5, (T)1.0);
                    A.x = host_random<T>(g, (T)0.5, (T)1.0);
                    T fz = host_random<T>(g, (T)0.0, (T)0.5);         // main.cu:183
                    sc.set(i, c, (T)0.2, METAL, A, fz < (T)1 ? fz : (T)1, 0);   // material.h:29-30
                } else {
                    sc.set(i, c, (T)0.2, DIELECTRIC, {0, 0, 0}, 0, (T)1.5);      // main.cu:188-189
                }
            }
        }
    int i = sc.n - 3;                                                 // main.cu:287-296
    sc.set(i, {0, 1, 0}, 1, DIELECTRIC, {0, 0, 0}, 0, (T)1.5);
    sc.set(i + 1, {-4, 1, 0}, 1, LAMBERTIAN, {(T)0.4, (T)0.2, (T)0.1}, 0, 0);
    sc.set(i + 2, {4, 1, 0}, 1, METAL, {(T)0.7, (T)0.6, (T)0.5}, 0, 0);
}

// =====================================================================================
// Camera (camera.h:33-68; src/InOneWeekend/camera.h:68-103). Host code => no FMA.
// =====================================================================================
template <class T> struct Camera {
    int W, H, S, B;
    T pixel_samples_scale;
    V3<T> center, pixel00_loc, pixel_delta_u, pixel_delta_v;
    T defocus_angle;
    V3<T> defocus_disk_u, defocus_disk_v;
};

template <class T> void camera_init(Camera<T>& cam, int W, int H, int S, int B) {
    const T pi = (T)3.1415926535897932385L;                          // rtweekend.h:14
    cam.W = W; cam.H = H; cam.S = S; cam.B = B;
    cam.pixel_samples_scale = (T)1 / (T)S;                            // camera.h:34
    const T vfov = 20, defocus_angle = (T)0.6, focus_dist = 10;       // main.cu:114-121
    const V3<T> lookfrom = {13, 2, 3}, lookat = {0, 0, 0}, vup = {0, 1, 0};
    cam.center = lookfrom;
    T theta = vfov * pi / (T)180;                                     // rtweekend.h:18-20
    T h = std::tan(theta / 2);
    T viewport_height = (T)2 * h * focus_dist;
    T viewport_width = viewport_height * ((T)W / (T)H);               // camera.h:45: float(W)/H
    V3<T> w = unit<false>(lookfrom - lookat);
    V3<T> u = unit<false>(cross(vup, w));
    V3<T> v = cross(w, u);
    V3<T> viewport_u = scale(viewport_width, u);
    V3<T> viewport_v = scale(viewport_height, -v);
    cam.pixel_delta_u = vdiv(viewport_u, (T)W);
    cam.pixel_delta_v = vdiv(viewport_v, (T)H);
    V3<T> upper_left = cam.center - scale(focus_dist, w) - vdiv(viewport_u, (T)2) - vdiv(viewport_v, (T)2);
    cam.pixel00_loc = upper_left + scale((T)0.5, cam.pixel_delta_u + cam.pixel_delta_v);
    T defocus_radius = focus_dist * std::tan((defocus_angle / 2) * pi / (T)180);
    cam.defocus_angle = defocus_angle;
    cam.defocus_disk_u = scale(defocus_radius, u);                    // u * r == r * u (vec3.h:85-87)
    cam.defocus_disk_v = scale(defocus_radius, v);
}

// =====================================================================================
// hit_sphere / hit_world (hittable.h:40-98; src/InOneWeekend/sphere.h:22-49)
// =====================================================================================
template <class T> struct Hit { V3<T> p, normal; T t; bool front; int idx; };

template <bool F, class T>
inline bool hit_sphere(V3<T> center, T radius, V3<T> O, V3<T> D, T tmin, T tmax, Hit<T>& rec) {
    V3<T> oc = center - O;                                            // hittable.h:42
    T a = len2<F>(D);                                                 // :43
    T h = dot<F>(D, oc);                                              // :44
    T c = len2<F>(oc) - radius * radius;                              // :45
    T disc = F ? std::fma(h, h, -(a * c)) : h * h - a * c;            // :47
    if (disc < 0) return false;                                       // :48
    T sqrtd = std::sqrt(disc);                                        // :50
    T root = (h - sqrtd) / a;                                         // :53
    if (!(tmin < root && root < tmax)) {                              // :54 surrounds (open interval)
        root = (h + sqrtd) / a;
        if (!(tmin < root && root < tmax)) return false;
    }
    rec.t = root;                                                     // :59
    rec.p = F ? madd<true>(root, D, O) : O + scale(root, D);          // :60 ray::at
    V3<T> outward = vdiv(rec.p - center, radius);                     // :61
    rec.front = dot<F>(D, outward) < 0;                               // :24
    rec.normal = rec.front ? outward : -outward;                      // :25
    return true;
}

// Compact world: skipped grid cells are dropped (serial main.cc simply never adds them;
// in the CUDA variant the slot keeps an uninitialised radius -- UB).
template <class T> struct World { std::vector<V3<T>> center, albedo; std::vector<T> radius, fuzz, ri; std::vector<int> type; int n = 0; };

template <class T> void compact(const Scene<T>& sc, World<T>& w) {
    for (int i = 0; i < sc.n; ++i) if (sc.valid[i]) {
        w.center.push_back(sc.center[i]); w.albedo.push_back(sc.albedo[i]); w.radius.push_back(sc.radius[i]);
        w.fuzz.push_back(sc.fuzz[i]); w.ri.push_back(sc.ri[i]); w.type.push_back(sc.type[i]); ++w.n;
    }
}

template <bool F, class T>
inline bool hit_world(const World<T>& w, V3<T> O, V3<T> D, T tmin, T tmax, Hit<T>& rec) {
    Hit<T> tmp; bool any = false; T closest = tmax;                   // hittable.h:82-84
    for (int i = 0; i < w.n; ++i)
        if (hit_sphere<F>(w.center[i], w.radius[i], O, D, tmin, closest, tmp)) {
            any = true; closest = tmp.t; rec = tmp; rec.idx = i;      // :88-92
        }
    return any;
}

// =====================================================================================
// ONE scatter / sky / ray_color implementation for both semantics.
//
// Everything after hit_world -- material dispatch, the three scatters, random_unit_vector,
// Schlick, the sky blend, the bounce loop -- exists ONCE below, as templates over a POLICY
// that holds only what differs between the reference's two programs:
//
//                        SerialPolicy (src/InOneWeekend)         CudaPolicy<T> (src/Global{Float,Double}CUDA...)
//   real type            double                                  T = float | double
//   a*b+c contraction    none (g++ -O3, x86-64, no -mfma)        explicit fma at the DESIGN.md places
//   RNG                  one glibc rand() stream, [0,1)          per-pixel XORWOW, (0,1]
//   cube point           vec3::random(-1,1): z,y,x (g++ order),  x,y,z, 2u-1 (vec3.h:119-121)
//                        min+(max-min)*r  (vec3.h:124-131)
//   near_zero / ruv eps  1e-8 / 1e-160                           1e-6f / 1e-8f (fp64: 1e-8 / 1e-160)
//   pow(1-cos, 5)        std::pow                                powf, as the float chain ((x*x)*(x*x))*x
//
// plus two run-time switches that are independent of the policy:
//   LoopForm  RECURSIVE (src/InOneWeekend/camera.h:137-156: att * ray_color(...))
//             ITERATIVE (Global*/camera.h:84-127: running attenuation product, then * sky)
//   SkyMode   SKY_CURRENT (serial camera.h:153: the ray that missed)
//             SKY_PRIMARY (Global*/camera.h:121: the PRIMARY ray r, not curr_ray)
//
// The reference programs are the corners {Serial, RECURSIVE, SKY_CURRENT} -- pinned BYTE FOR
// BYTE against the reference's own sources built into oracle/_ref -- and {Cuda, ITERATIVE,
// SKY_PRIMARY}, the oracle of the HIP kernel.  Because the code is shared, the byte-exact pin
// covers the material dispatch the hot path uses; the other corners exist so that tests can
// walk from the pinned corner to the CUDA corner one switch at a time
// (tests/test_oracle_pins.py: loop form with identical random numbers, then sky mode, then
// policy against reference-run images with SURVEY A.5 noise floors).
// =====================================================================================
enum LoopForm { RECURSIVE = 0, ITERATIVE = 1 };
enum SkyMode { SKY_CURRENT = 0, SKY_PRIMARY = 1 };

struct SerialPolicy {
    typedef double real;
    typedef GlibcRand Rng;
    static constexpr bool fused = false;
    static constexpr double near_zero = 1e-8;                          // src/InOneWeekend/vec3.h:50
    static constexpr double ruv_eps = 1e-160;                          // src/InOneWeekend/vec3.h:128
    static double uniform(Rng& g) { return host_random<double>(g); }   // rtweekend.h:37-40
    static V3<double> cube_point(Rng& g) {                             // vec3::random(-1,1): z,y,x (g++ order)
        V3<double> p;
        p.z = host_random<double>(g, -1.0, 1.0);
        p.y = host_random<double>(g, -1.0, 1.0);
        p.x = host_random<double>(g, -1.0, 1.0);
        return p;
    }
    static double pow5(double x) { return std::pow(x, 5); }            // material.h:105
};

template <class T> struct CudaEps;
template <> struct CudaEps<float> { static constexpr float near_zero = 1e-6f, ruv = 1e-8f; };    // vec3.h:50,124
template <> struct CudaEps<double> { static constexpr double near_zero = 1e-8, ruv = 1e-160; };  // GlobalDouble vec3.h:50,125

template <class T> struct CudaPolicy {
    typedef T real;
    typedef Xorwow Rng;
    static constexpr bool fused = true;
    static constexpr T near_zero = CudaEps<T>::near_zero;
    static constexpr T ruv_eps = CudaEps<T>::ruv;
    static T uniform(Rng& s) { return device_uniform<T>(s); }          // rtweekend.h:32-35
    static V3<T> cube_point(Rng& s) {                                  // vec3.h:119-121: x, y, z
        T x = std::fma(device_uniform<T>(s), (T)2, (T)-1);
        T y = std::fma(device_uniform<T>(s), (T)2, (T)-1);
        T z = std::fma(device_uniform<T>(s), (T)2, (T)-1);
        return {x, y, z};
    }
    // material.h:65 powf(1 - cosine, 5) -- powf even in the double build (GlobalDouble material.h:68)
    static T pow5(T x) { float f = (float)x; float f2 = f * f; return (T)((f2 * f2) * f); }
};

// vec3.h:117-127 (CUDA) / src/InOneWeekend/vec3.h:124-131
template <class P> inline V3<typename P::real> random_unit_vector(typename P::Rng& rng) {
    typedef typename P::real T;
    for (;;) {
        V3<T> p = P::cube_point(rng);
        T lensq = len2<P::fused>(p);
        if (P::ruv_eps < lensq && lensq <= (T)1) return vdiv(p, (T)std::sqrt(lensq));
    }
}

// material.h:62-66 (CUDA) / src/InOneWeekend/material.h:101-106:  r0 + (1-r0)*pow(1-cosine, 5)
template <class P> inline typename P::real reflectance(typename P::real cosine, typename P::real ri) {
    typedef typename P::real T;
    T r0 = ((T)1 - ri) / ((T)1 + ri);
    r0 = r0 * r0;
    return mad<P::fused>((T)1 - r0, P::pow5((T)1 - cosine), r0);
}

// material.h:38-89 (CUDA) / src/InOneWeekend/material.h:32-92.  Returns false when the path is
// absorbed (metal scattering below the surface).
template <class P>
inline bool scatter(const World<typename P::real>& w, V3<typename P::real> D, const Hit<typename P::real>& rec,
                    typename P::Rng& rng, V3<typename P::real>& att, V3<typename P::real>& nd) {
    typedef typename P::real T;
    constexpr bool F = P::fused;
    const int m = rec.idx;
    if (w.type[m] == LAMBERTIAN) {                                    // material.h:38-49
        nd = rec.normal + random_unit_vector<P>(rng);
        if (std::fabs(nd.x) < P::near_zero && std::fabs(nd.y) < P::near_zero && std::fabs(nd.z) < P::near_zero) nd = rec.normal;
        att = w.albedo[m];
        return true;
    }
    if (w.type[m] == METAL) {                                         // material.h:51-59
        V3<T> ur = unit<F>(reflect<F>(D, rec.normal));
        nd = madd<F>(w.fuzz[m], random_unit_vector<P>(rng), ur);     // unit(reflected) + fuzz * random_unit_vector
        att = w.albedo[m];
        return dot<F>(nd, rec.normal) > 0;
    }
    att = {1, 1, 1};                                                  // material.h:68-89
    T ri = rec.front ? ((T)1 / w.ri[m]) : w.ri[m];
    V3<T> ud = unit<F>(D);
    T cos_theta = std::fmin(dot<F>(-ud, rec.normal), (T)1);
    T sin_theta = std::sqrt(mad<F>(-cos_theta, cos_theta, (T)1));    // sqrt(1 - cos*cos)
    bool cannot = ri * sin_theta > (T)1;
    if (cannot || reflectance<P>(cos_theta, ri) > P::uniform(rng)) nd = reflect<F>(ud, rec.normal);   // no draw on TIR
    else nd = refract<F>(ud, rec.normal, ri);
    return true;
}

// camera.h:121-123 (CUDA; `a` is double even in the float build) / src/InOneWeekend/camera.h:153-155:
//   (1-a)*white + a*(0.5,0.7,1.0),  a = 0.5*(unit(dir).y + 1)
template <class P> inline V3<typename P::real> sky_colour(V3<typename P::real> dir) {
    typedef typename P::real T;
    V3<T> ud = unit<P::fused>(dir);
    double a = 0.5 * ((double)ud.y + 1.0);
    T w1 = (T)(1.0 - a), w2 = (T)a;                                   // operator*(T, vec3): the weights are converted to T
    return {mad<P::fused>(w2, (T)0.5, w1), mad<P::fused>(w2, (T)0.7, w1), mad<P::fused>(w2, (T)1.0, w1)};
}

struct RenderStats { uint64_t primary_rays, segments, sphere_tests, rng_draws; };

// Global*/camera.h:78-128: iterative form.
template <class P>
inline V3<typename P::real> ray_color_iterative(V3<typename P::real> O0, V3<typename P::real> D0, int max_depth,
                                                const World<typename P::real>& w, typename P::Rng& rng, SkyMode sky, RenderStats& st) {
    typedef typename P::real T;
    V3<T> O = O0, D = D0;                                             // camera.h:82
    V3<T> atten = {1, 1, 1};                                          // :83
    for (int depth = 0; depth < max_depth; ++depth) {                 // :84
        Hit<T> rec;
        ++st.segments; st.sphere_tests += (uint64_t)w.n;
        if (hit_world<P::fused>(w, O, D, (T)0.001, std::numeric_limits<T>::infinity(), rec)) {   // :87
            V3<T> att, nd;
            if (!scatter<P>(w, D, rec, rng, att, nd)) return {0, 0, 0};   // :117
            atten = atten * att;                                      // :112
            O = rec.p; D = nd;                                        // :114
        } else {
            return atten * sky_colour<P>(sky == SKY_PRIMARY ? D0 : D);   // :120-124 (:121 uses the PRIMARY ray r)
        }
    }
    return {0, 0, 0};                                                 // :127
}

// src/InOneWeekend/camera.h:137-156: recursive form.  D0 is only read by SKY_PRIMARY.
template <class P>
V3<typename P::real> ray_color_recursive(V3<typename P::real> O, V3<typename P::real> D, V3<typename P::real> D0, int depth,
                                         const World<typename P::real>& w, typename P::Rng& rng, SkyMode sky, RenderStats& st) {
    typedef typename P::real T;
    if (depth <= 0) return {0, 0, 0};                                 // :139-140
    Hit<T> rec;
    ++st.segments; st.sphere_tests += (uint64_t)w.n;
    if (hit_world<P::fused>(w, O, D, (T)0.001, std::numeric_limits<T>::infinity(), rec)) {   // :144
        V3<T> att, nd;
        if (scatter<P>(w, D, rec, rng, att, nd)) return att * ray_color_recursive<P>(rec.p, nd, D0, depth - 1, w, rng, sky, st);   // :147-149
        return {0, 0, 0};                                             // :150
    }
    return sky_colour<P>(sky == SKY_PRIMARY ? D0 : D);                // :153-155
}

template <class P>
inline V3<typename P::real> ray_color(LoopForm form, V3<typename P::real> O, V3<typename P::real> D, int max_depth,
                                      const World<typename P::real>& w, typename P::Rng& rng, SkyMode sky, RenderStats& st) {
    if (form == ITERATIVE) return ray_color_iterative<P>(O, D, max_depth, w, rng, sky, st);
    return ray_color_recursive<P>(O, D, D, max_depth, w, rng, sky, st);
}

// =====================================================================================
// CUDA-semantics render (camera.h:130-172): per-pixel XORWOW streams, so rows are independent
// and are rendered by `threads` host threads (the result does not depend on the split).
// =====================================================================================
template <class T> inline V3<T> dev_random_in_unit_disk(Xorwow& s) {  // vec3.h:109-115; rtweekend.h:37-40
    for (;;) {
        T px = std::fma((T)2, device_uniform<T>(s), (T)-1);          // first argument drawn first (documented choice)
        T py = std::fma((T)2, device_uniform<T>(s), (T)-1);
        if (std::fma(py, py, px * px) < (T)1) return {px, py, 0};
    }
}

template <class T>
void render_cuda_rows(const World<T>& w, const Camera<T>& cam, uint64_t seed, int row0, int row1, int out_row0,
                      T* out_rgb, RenderStats& st, uint32_t* seg_per_pixel, LoopForm form, SkyMode sky) {
    typedef CudaPolicy<T> P;
    for (int j = row0; j < row1; ++j)
        for (int i = 0; i < cam.W; ++i) {
            const int pixel_index = j * cam.W + i;                    // camera.h:134
            Xorwow s; xorwow_init(s, seed, (uint64_t)pixel_index, 0); // rtweekend.h:49
            V3<T> pc = {0, 0, 0};
            const uint64_t seg0 = st.segments;
            for (int sample = 0; sample < cam.S; ++sample) {          // camera.h:141
                T ox = device_uniform<T>(s) - (T)0.5;                 // :145 (first argument drawn first)
                T oy = device_uniform<T>(s) - (T)0.5;                 // :146
                T fi = (T)i + ox, fj = (T)j + oy;                     // :149-150
                V3<T> ps = madd<true>(fj, cam.pixel_delta_v, madd<true>(fi, cam.pixel_delta_u, cam.pixel00_loc));
                V3<T> org = cam.center;
                if (!(cam.defocus_angle <= 0)) {                      // :152-153, :73-76
                    V3<T> p = dev_random_in_unit_disk<T>(s);
                    org = madd<true>(p.y, cam.defocus_disk_v, madd<true>(p.x, cam.defocus_disk_u, cam.center));
                }
                V3<T> dir = ps - org;                                 // :154
                ++st.primary_rays;
                pc = pc + ray_color<P>(form, org, dir, cam.B, w, s, sky, st);   // :160
            }
            const size_t o_idx = (size_t)(j - out_row0) * cam.W + i;
            if (seg_per_pixel) seg_per_pixel[o_idx] = (uint32_t)(st.segments - seg0);
            pc = scale(cam.pixel_samples_scale, pc);                  // :167
            T* o = out_rgb + o_idx * 3;
            o[0] = pc.x > 0 ? (T)std::sqrt(pc.x) : 0;                 // color.h:10-13
            o[1] = pc.y > 0 ? (T)std::sqrt(pc.y) : 0;
            o[2] = pc.z > 0 ? (T)std::sqrt(pc.z) : 0;
        }
}

int g_threads = 0;   // 0: std::thread::hardware_concurrency(), capped at 16 (oracle_set_threads)

template <class T>
void render_cuda_semantics(const World<T>& w, const Camera<T>& cam, uint64_t seed, int row0, int row1,
                           T* out_rgb /* (row1-row0)*W*3 */, RenderStats& st, uint32_t* seg_per_pixel = nullptr,
                           LoopForm form = ITERATIVE, SkyMode sky = SKY_PRIMARY) {
    jump_tables();                                                    // build once, before the threads read it
    int nt = g_threads > 0 ? g_threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    const int rows = row1 - row0;
    if ((long long)rows * cam.W * cam.S < 20000) nt = 1;
    if (nt > rows) nt = rows > 0 ? rows : 1;
    if (nt <= 1) { render_cuda_rows<T>(w, cam, seed, row0, row1, row0, out_rgb, st, seg_per_pixel, form, sky); return; }
    // rows are dealt in small interleaved chunks so that sky rows and ground rows mix per thread
    std::vector<RenderStats> part((size_t)nt, RenderStats{0, 0, 0, 0});
    std::vector<std::thread> pool;
    std::atomic<int> next(row0);
    for (int t = 0; t < nt; ++t)
        pool.emplace_back([&, t]() {
            for (;;) {
                const int r = next.fetch_add(2);
                if (r >= row1) break;
                render_cuda_rows<T>(w, cam, seed, r, std::min(r + 2, row1), row0, out_rgb, part[(size_t)t], seg_per_pixel, form, sky);
            }
        });
    for (std::thread& th : pool) th.join();
    for (const RenderStats& p : part) { st.primary_rays += p.primary_rays; st.segments += p.segments; st.sphere_tests += p.sphere_tests; }
}

// =====================================================================================
// Serial-semantics render (src/InOneWeekend/camera.h:34-54, 105-135): fp64, ONE rand() stream
// shared by scene generation and rendering, pixels in row-major order.
// =====================================================================================
typedef V3<double> D3;

inline D3 ser_random_in_unit_disk(GlibcRand& g) {                     // vec3.h:116-122 (serial)
    for (;;) {
        D3 p; p.z = 0;
        p.y = host_random<double>(g, -1.0, 1.0);                      // right-to-left
        p.x = host_random<double>(g, -1.0, 1.0);
        if (len2<false>(p) < 1) return p;
    }
}

inline int to_byte(double c) {                                        // color.h:29-47 (serial), main.cu:374-376
    double g = c > 0 ? std::sqrt(c) : 0;
    double cl = g < 0.000 ? 0.000 : (g > 0.999 ? 0.999 : g);
    return (int)(256 * cl);
}

// The single rand() stream makes a row subset a different (equally valid) sample, so the
// byte-exact pin always renders all rows.  form/sky default to the reference's program.
void render_serial_semantics(int scene_id, int W, int H, int S, int depth, std::string& p3, RenderStats& st,
                             LoopForm form = RECURSIVE, SkyMode sky = SKY_CURRENT) {
    GlibcRand g(1);
    Scene<double> sc; build_scene<double>(scene_id, sc, g);
    World<double> w; compact(sc, w);
    Camera<double> cam; camera_init<double>(cam, W, H, S, depth);
    char buf[64];
    std::snprintf(buf, sizeof buf, "P3\n%d %d\n255\n", W, H);
    p3 = buf;
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            D3 pc = {0, 0, 0};
            for (int s = 0; s < S; ++s) {
                double oy = host_random<double>(g) - 0.5;             // camera.h:120: right-to-left
                double ox = host_random<double>(g) - 0.5;
                D3 ps = cam.pixel00_loc + scale(i + ox, cam.pixel_delta_u) + scale(j + oy, cam.pixel_delta_v);
                D3 org = cam.center;
                if (!(cam.defocus_angle <= 0)) {
                    D3 p = ser_random_in_unit_disk(g);
                    org = cam.center + scale(p.x, cam.defocus_disk_u) + scale(p.y, cam.defocus_disk_v);
                }
                ++st.primary_rays;
                pc = pc + ray_color<SerialPolicy>(form, org, ps - org, depth, w, g, sky, st);
            }
            pc = scale(cam.pixel_samples_scale, pc);
            std::snprintf(buf, sizeof buf, "%d %d %d\n", to_byte(pc.x), to_byte(pc.y), to_byte(pc.z));
            p3 += buf;
        }
}

template <class T> void scene_to_arrays(const Scene<T>& sc, T* center_radius, T* albedo_fuzz, T* ri, int* type, int* valid) {
    for (int i = 0; i < sc.n; ++i) {
        center_radius[4 * i + 0] = sc.center[i].x; center_radius[4 * i + 1] = sc.center[i].y;
        center_radius[4 * i + 2] = sc.center[i].z; center_radius[4 * i + 3] = sc.radius[i];
        albedo_fuzz[4 * i + 0] = sc.albedo[i].x; albedo_fuzz[4 * i + 1] = sc.albedo[i].y;
        albedo_fuzz[4 * i + 2] = sc.albedo[i].z; albedo_fuzz[4 * i + 3] = sc.fuzz[i];
        ri[i] = sc.ri[i]; type[i] = sc.type[i]; valid[i] = sc.valid[i];
    }
}

template <class T> void world_from_arrays(int n, const T* cr, const T* af, const T* ri, const int* type, World<T>& w) {
    for (int i = 0; i < n; ++i) {
        w.center.push_back({cr[4 * i], cr[4 * i + 1], cr[4 * i + 2]}); w.radius.push_back(cr[4 * i + 3]);
        w.albedo.push_back({af[4 * i], af[4 * i + 1], af[4 * i + 2]}); w.fuzz.push_back(af[4 * i + 3]);
        w.ri.push_back(ri[i]); w.type.push_back(type[i]);
    }
    w.n = n;
}

// Flat camera record: 4 ints then 1+3+3+3+3+1+3+3 = 20 T.
template <class T> void camera_to_flat(const Camera<T>& c, int* ints, T* f) {
    ints[0] = c.W; ints[1] = c.H; ints[2] = c.S; ints[3] = c.B;
    int k = 0;
    f[k++] = c.pixel_samples_scale;
    for (V3<T> v : {c.center, c.pixel00_loc, c.pixel_delta_u, c.pixel_delta_v}) { f[k++] = v.x; f[k++] = v.y; f[k++] = v.z; }
    f[k++] = c.defocus_angle;
    for (V3<T> v : {c.defocus_disk_u, c.defocus_disk_v}) { f[k++] = v.x; f[k++] = v.y; f[k++] = v.z; }
}
template <class T> void camera_from_flat(Camera<T>& c, const int* ints, const T* f) {
    c.W = ints[0]; c.H = ints[1]; c.S = ints[2]; c.B = ints[3];
    int k = 0;
    c.pixel_samples_scale = f[k++];
    V3<T>* vs[] = {&c.center, &c.pixel00_loc, &c.pixel_delta_u, &c.pixel_delta_v};
    for (V3<T>* v : vs) { v->x = f[k++]; v->y = f[k++]; v->z = f[k++]; }
    c.defocus_angle = f[k++];
    V3<T>* ds[] = {&c.defocus_disk_u, &c.defocus_disk_v};
    for (V3<T>* v : ds) { v->x = f[k++]; v->y = f[k++]; v->z = f[k++]; }
}

}  // namespace

// =====================================================================================
// C entry points (ctypes).  precision: 32 or 64.
// =====================================================================================
extern "C" {

int oracle_glibc_rand(int n, int* out) { GlibcRand g(1); for (int i = 0; i < n; ++i) out[i] = g.next(); return 0; }

int oracle_scene_slots(int scene_id) { return scene_id == 1 ? 488 : scene_id == 2 ? 40 : 125; }

int oracle_build_scene(int scene_id, int precision, void* center_radius, void* albedo_fuzz, void* ri, int* type, int* valid) {
    GlibcRand g(1);
    if (precision == 32) { Scene<float> sc; build_scene<float>(scene_id, sc, g); scene_to_arrays(sc, (float*)center_radius, (float*)albedo_fuzz, (float*)ri, type, valid); return sc.n; }
    if (precision == 64) { Scene<double> sc; build_scene<double>(scene_id, sc, g); scene_to_arrays(sc, (double*)center_radius, (double*)albedo_fuzz, (double*)ri, type, valid); return sc.n; }
    return -1;
}

int oracle_camera_init(int precision, int W, int H, int S, int B, int* ints4, void* flat20) {
    if (precision == 32) { Camera<float> c; camera_init<float>(c, W, H, S, B); camera_to_flat(c, ints4, (float*)flat20); return 0; }
    if (precision == 64) { Camera<double> c; camera_init<double>(c, W, H, S, B); camera_to_flat(c, ints4, (double*)flat20); return 0; }
    return -1;
}

void oracle_xorwow_init(unsigned long long seed, unsigned long long subsequence, unsigned long long offset, int salt, unsigned int* state6) {
    Xorwow s; xorwow_init(s, seed, subsequence, offset, salt);
    for (int k = 0; k < 5; ++k) state6[k] = s.v[k];
    state6[5] = s.d;
}
unsigned int oracle_xorwow_next(unsigned int* state6) {
    Xorwow s; for (int k = 0; k < 5; ++k) s.v[k] = state6[k]; s.d = state6[5];
    unsigned int r = xorwow_next(s);
    for (int k = 0; k < 5; ++k) state6[k] = s.v[k]; state6[5] = s.d;
    return r;
}
float oracle_uniform_f32(unsigned int* state6) {
    Xorwow s; for (int k = 0; k < 5; ++k) s.v[k] = state6[k]; s.d = state6[5];
    float r = device_uniform<float>(s);
    for (int k = 0; k < 5; ++k) state6[k] = s.v[k]; state6[5] = s.d;
    return r;
}
double oracle_uniform_f64(unsigned int* state6) {
    Xorwow s; for (int k = 0; k < 5; ++k) s.v[k] = state6[k]; s.d = state6[5];
    double r = device_uniform<double>(s);
    for (int k = 0; k < 5; ++k) state6[k] = s.v[k]; state6[5] = s.d;
    return r;
}

// Host threads used by the CUDA-semantics renders (0 = all cores, at most 16).  The result does
// not depend on it (per-pixel streams).
void oracle_set_threads(int n) { g_threads = n < 0 ? 0 : n; }

// CUDA-policy render of rows [row0,row1) of a compact world given as arrays, with the loop form
// (0 recursive, 1 iterative) and sky mode (0 current ray, 1 primary ray) switches.  The
// reference's GPU program is (1, 1).   stats4 = {primary_rays, segments, sphere_tests, 0}.
int oracle_render_modes(int precision, int n, const void* center_radius, const void* albedo_fuzz, const void* ri, const int* type,
                        const int* cam_ints4, const void* cam_flat20, unsigned long long seed, int row0, int row1,
                        int loop_form, int sky_mode, void* out_rgb, unsigned long long* stats4, unsigned int* seg_per_pixel) {
    RenderStats st = {0, 0, 0, 0};
    const LoopForm form = loop_form ? ITERATIVE : RECURSIVE;
    const SkyMode sky = sky_mode ? SKY_PRIMARY : SKY_CURRENT;
    if (precision == 32) {
        World<float> w; world_from_arrays(n, (const float*)center_radius, (const float*)albedo_fuzz, (const float*)ri, type, w);
        Camera<float> c; camera_from_flat(c, cam_ints4, (const float*)cam_flat20);
        render_cuda_semantics<float>(w, c, seed, row0, row1, (float*)out_rgb, st, seg_per_pixel, form, sky);
    } else if (precision == 64) {
        World<double> w; world_from_arrays(n, (const double*)center_radius, (const double*)albedo_fuzz, (const double*)ri, type, w);
        Camera<double> c; camera_from_flat(c, cam_ints4, (const double*)cam_flat20);
        render_cuda_semantics<double>(w, c, seed, row0, row1, (double*)out_rgb, st, seg_per_pixel, form, sky);
    } else return -1;
    if (stats4) { stats4[0] = st.primary_rays; stats4[1] = st.segments; stats4[2] = st.sphere_tests; stats4[3] = 0; }
    return 0;
}

// The reference's GPU program: CUDA policy, iterative loop, sky from the primary ray.
int oracle_render(int precision, int n, const void* center_radius, const void* albedo_fuzz, const void* ri, const int* type,
                  const int* cam_ints4, const void* cam_flat20, unsigned long long seed, int row0, int row1,
                  void* out_rgb, unsigned long long* stats4) {
    return oracle_render_modes(precision, n, center_radius, albedo_fuzz, ri, type, cam_ints4, cam_flat20, seed, row0, row1, 1, 1, out_rgb, stats4, nullptr);
}

// As oracle_render (fp32) plus the number of path segments (hit_world calls) of every pixel.
int oracle_render_segments_f32(int n, const float* center_radius, const float* albedo_fuzz, const float* ri, const int* type,
                               const int* cam_ints4, const float* cam_flat20, unsigned long long seed, int row0, int row1,
                               float* out_rgb, unsigned int* seg_per_pixel) {
    return oracle_render_modes(32, n, center_radius, albedo_fuzz, ri, type, cam_ints4, cam_flat20, seed, row0, row1, 1, 1, out_rgb, nullptr, seg_per_pixel);
}

// Serial-policy render (one rand() stream, all rows) with the same two switches; the reference's
// serial program is (0, 0).  Returns the P3 text length, copies up to cap bytes.
long long oracle_render_serial_modes(int scene_id, int W, int H, int S, int depth, int loop_form, int sky_mode,
                                     char* out, long long cap, unsigned long long* stats4) {
    std::string p3; RenderStats st = {0, 0, 0, 0};
    render_serial_semantics(scene_id, W, H, S, depth, p3, st, loop_form ? ITERATIVE : RECURSIVE, sky_mode ? SKY_PRIMARY : SKY_CURRENT);
    if (out && cap > 0) std::memcpy(out, p3.data(), (size_t)std::min<long long>(cap, (long long)p3.size()));
    if (stats4) { stats4[0] = st.primary_rays; stats4[1] = st.segments; stats4[2] = st.sphere_tests; stats4[3] = 0; }
    return (long long)p3.size();
}
long long oracle_render_serial(int scene_id, int W, int H, int S, int depth, char* out, long long cap, unsigned long long* stats4) {
    return oracle_render_serial_modes(scene_id, W, H, S, depth, 0, 0, out, cap, stats4);
}

// The sky term alone (camera.h:121-123 / serial camera.h:153-155) for one direction.
// policy: 0 serial (fp64, unfused), 32 / 64 CUDA policy in that precision.  out3 is double.
int oracle_sky(int policy, const double* dir3, double* out3) {
    if (policy == 0) { D3 c = sky_colour<SerialPolicy>({dir3[0], dir3[1], dir3[2]}); out3[0] = c.x; out3[1] = c.y; out3[2] = c.z; return 0; }
    if (policy == 64) { D3 c = sky_colour<CudaPolicy<double>>({dir3[0], dir3[1], dir3[2]}); out3[0] = c.x; out3[1] = c.y; out3[2] = c.z; return 0; }
    if (policy == 32) {
        V3<float> c = sky_colour<CudaPolicy<float>>({(float)dir3[0], (float)dir3[1], (float)dir3[2]});
        out3[0] = c.x; out3[1] = c.y; out3[2] = c.z; return 0;
    }
    return -1;
}

// hit_world (hittable.h:80-98, CUDA policy: explicit fma, tmin 0.001, tmax infinity) alone on n caller-supplied
// rays {ox,oy,oz,dx,dy,dz}: nearest root (+inf: none) and sphere index (-1) per ray.  Twin of the library's
// rtiow_debug_hit_world, for ray-by-ray comparison (tests/test_gpu_parity.py).
int oracle_hit_world(int precision, int n_spheres, const void* center_radius, int n_rays, const void* rays, void* t_out, int* index_out) {
    auto run = [&](auto tag) {
        using T = decltype(tag);
        const T* cr = (const T*)center_radius; const T* r = (const T*)rays; T* t = (T*)t_out;
        World<T> w;
        for (int i = 0; i < n_spheres; ++i) { w.center.push_back({cr[4 * i], cr[4 * i + 1], cr[4 * i + 2]}); w.radius.push_back(cr[4 * i + 3]); }
        w.n = n_spheres;
        for (int k = 0; k < n_rays; ++k) {
            Hit<T> rec;
            const bool any = hit_world<true, T>(w, {r[6 * k], r[6 * k + 1], r[6 * k + 2]}, {r[6 * k + 3], r[6 * k + 4], r[6 * k + 5]}, (T)0.001, std::numeric_limits<T>::infinity(), rec);
            t[k] = any ? rec.t : std::numeric_limits<T>::infinity();
            index_out[k] = any ? rec.idx : -1;
        }
    };
    if (precision == 32) run(float());
    else if (precision == 64) run(double());
    else return -1;
    return 0;
}

// Single-primitive probes used by the analytic known-answer tests (tests/test_oracle_kat.py).
int oracle_hit_sphere_f64(const double* center, double radius, const double* O, const double* D, double tmin, double tmax,
                          double* t, double* p3, double* n3, int* front) {
    Hit<double> rec;
    bool ok = hit_sphere<true, double>({center[0], center[1], center[2]}, radius, {O[0], O[1], O[2]}, {D[0], D[1], D[2]}, tmin, tmax, rec);
    if (!ok) return 0;
    *t = rec.t; p3[0] = rec.p.x; p3[1] = rec.p.y; p3[2] = rec.p.z; n3[0] = rec.normal.x; n3[1] = rec.normal.y; n3[2] = rec.normal.z; *front = rec.front;
    return 1;
}
void oracle_reflect_f64(const double* v, const double* n, double* out) { D3 r = reflect<true, double>({v[0], v[1], v[2]}, {n[0], n[1], n[2]}); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void oracle_refract_f64(const double* v, const double* n, double eta, double* out) { D3 r = refract<true, double>({v[0], v[1], v[2]}, {n[0], n[1], n[2]}, eta); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
double oracle_reflectance_f64(double cosine, double ri) { return reflectance<CudaPolicy<double>>(cosine, ri); }

}  // extern "C"
