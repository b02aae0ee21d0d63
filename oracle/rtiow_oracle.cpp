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
#include <string>
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
// CUDA-semantics render (camera.h:73-172)
// =====================================================================================
template <class T> struct Eps;
template <> struct Eps<float> { static constexpr float near_zero = 1e-6f, ruv = 1e-8f; };    // vec3.h:50,124
template <> struct Eps<double> { static constexpr double near_zero = 1e-8, ruv = 1e-160; };  // GlobalDouble vec3.h:50,125

template <class T> inline V3<T> dev_random_unit_vector(Xorwow& s) {  // vec3.h:117-127
    for (;;) {
        T x = std::fma(device_uniform<T>(s), (T)2, (T)-1);
        T y = std::fma(device_uniform<T>(s), (T)2, (T)-1);
        T z = std::fma(device_uniform<T>(s), (T)2, (T)-1);
        V3<T> p = {x, y, z};
        T lensq = dot<true>(p, p);
        if (Eps<T>::ruv < lensq && lensq <= (T)1) return vdiv(p, (T)std::sqrt(lensq));
    }
}

template <class T> inline V3<T> dev_random_in_unit_disk(Xorwow& s) {  // vec3.h:109-115; rtweekend.h:37-40
    for (;;) {
        T px = std::fma((T)2, device_uniform<T>(s), (T)-1);          // first argument drawn first (documented choice)
        T py = std::fma((T)2, device_uniform<T>(s), (T)-1);
        if (std::fma(py, py, px * px) < (T)1) return {px, py, 0};
    }
}

// material.h:62-66. powf even in the double build (GlobalDouble material.h:68).
template <class T> inline T dev_reflectance(T cosine, T ri) {
    T r0 = ((T)1 - ri) / ((T)1 + ri);
    r0 = r0 * r0;
    float x = (float)((T)1 - cosine);
    float x2 = x * x;
    float p5 = (x2 * x2) * x;
    return std::fma((T)1 - r0, (T)p5, r0);
}

struct RenderStats { uint64_t primary_rays, segments, sphere_tests, rng_draws; };

template <class T>
inline V3<T> dev_ray_color(V3<T> O0, V3<T> D0, int max_depth, const World<T>& w, Xorwow& s, RenderStats& st) {
    V3<T> O = O0, D = D0;                                             // camera.h:82
    V3<T> atten = {1, 1, 1};                                          // :83
    for (int depth = 0; depth < max_depth; ++depth) {                 // :84
        Hit<T> rec;
        ++st.segments; st.sphere_tests += (uint64_t)w.n;
        if (hit_world<true>(w, O, D, (T)0.001, std::numeric_limits<T>::infinity(), rec)) {   // :87
            V3<T> nd, att; bool ok;
            const int m = rec.idx;
            if (w.type[m] == LAMBERTIAN) {                            // material.h:38-49
                nd = rec.normal + dev_random_unit_vector<T>(s);
                if (std::fabs(nd.x) < Eps<T>::near_zero && std::fabs(nd.y) < Eps<T>::near_zero &&
                    std::fabs(nd.z) < Eps<T>::near_zero) nd = rec.normal;
                att = w.albedo[m]; ok = true;
            } else if (w.type[m] == METAL) {                          // material.h:51-59
                V3<T> refl = reflect<true>(D, rec.normal);
                V3<T> ur = unit<true>(refl);
                V3<T> ruv = dev_random_unit_vector<T>(s);
                nd = madd<true>(w.fuzz[m], ruv, ur);
                att = w.albedo[m];
                ok = dot<true>(nd, rec.normal) > 0;
            } else {                                                  // material.h:68-89
                att = {1, 1, 1};
                T eta = w.ri[m];
                T ri = rec.front ? ((T)1 / eta) : eta;
                V3<T> ud = unit<true>(D);
                T cos_theta = std::fmin(dot<true>(-ud, rec.normal), (T)1);
                T sin_theta = std::sqrt(std::fma(-cos_theta, cos_theta, (T)1));
                bool cannot = ri * sin_theta > (T)1;
                if (cannot || dev_reflectance<T>(cos_theta, ri) > device_uniform<T>(s)) nd = reflect<true>(ud, rec.normal);
                else nd = refract<true>(ud, rec.normal, ri);
                ok = true;
            }
            if (!ok) return {0, 0, 0};                                // camera.h:117
            atten = atten * att;                                      // :112
            O = rec.p; D = nd;                                        // :114
        } else {
            V3<T> ud = unit<true>(D0);                                // :121 -- PRIMARY ray r, not curr_ray
            double a = 0.5 * ((double)ud.y + 1.0);                    // :122 (double even in the float build)
            T w1 = (T)(1.0 - a), w2 = (T)a;                           // :123 operator*(T, vec3)
            V3<T> sky = {std::fma(w2, (T)0.5, w1), std::fma(w2, (T)0.7, w1), std::fma(w2, (T)1.0, w1)};
            return atten * sky;
        }
    }
    return {0, 0, 0};                                                 // :127
}

template <class T>
void render_cuda_semantics(const World<T>& w, const Camera<T>& cam, uint64_t seed, int row0, int row1,
                           T* out_rgb /* (row1-row0)*W*3 */, RenderStats& st, uint32_t* seg_per_pixel = nullptr) {
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
                pc = pc + dev_ray_color<T>(org, dir, cam.B, w, s, st); // :160
            }
            if (seg_per_pixel) seg_per_pixel[(size_t)(j - row0) * cam.W + i] = (uint32_t)(st.segments - seg0);
            pc = scale(cam.pixel_samples_scale, pc);                  // :167
            T* o = out_rgb + ((size_t)(j - row0) * cam.W + i) * 3;
            o[0] = pc.x > 0 ? (T)std::sqrt(pc.x) : 0;                 // color.h:10-13
            o[1] = pc.y > 0 ? (T)std::sqrt(pc.y) : 0;
            o[2] = pc.z > 0 ? (T)std::sqrt(pc.z) : 0;
        }
}

// =====================================================================================
// Serial semantics (src/InOneWeekend): fp64, one rand() stream, recursion, sky from current ray.
// =====================================================================================
typedef V3<double> D3;

inline D3 ser_random_unit_vector(GlibcRand& g) {                      // vec3.h:124-131 (serial)
    for (;;) {
        D3 p;                                                         // vec3::random(-1,1): z,y,x (g++ order)
        p.z = host_random<double>(g, -1.0, 1.0);
        p.y = host_random<double>(g, -1.0, 1.0);
        p.x = host_random<double>(g, -1.0, 1.0);
        double lensq = len2<false>(p);
        if (1e-160 < lensq && lensq <= 1.0) return vdiv(p, std::sqrt(lensq));
    }
}

inline D3 ser_random_in_unit_disk(GlibcRand& g) {                     // vec3.h:116-122 (serial)
    for (;;) {
        D3 p; p.z = 0;
        p.y = host_random<double>(g, -1.0, 1.0);                      // right-to-left
        p.x = host_random<double>(g, -1.0, 1.0);
        if (len2<false>(p) < 1) return p;
    }
}

inline double ser_reflectance(double cosine, double ri) {             // material.h:101-106 (serial)
    double r0 = (1 - ri) / (1 + ri);
    r0 = r0 * r0;
    return r0 + (1 - r0) * std::pow((1 - cosine), 5);
}

D3 ser_ray_color(D3 O, D3 D, int depth, const World<double>& w, GlibcRand& g, RenderStats& st) {   // camera.h:137-156
    if (depth <= 0) return {0, 0, 0};
    Hit<double> rec;
    ++st.segments; st.sphere_tests += (uint64_t)w.n;
    if (hit_world<false>(w, O, D, 0.001, std::numeric_limits<double>::infinity(), rec)) {
        const int m = rec.idx;
        D3 nd, att;
        if (w.type[m] == LAMBERTIAN) {                                // material.h:32-43
            nd = rec.normal + ser_random_unit_vector(g);
            if (std::fabs(nd.x) < 1e-8 && std::fabs(nd.y) < 1e-8 && std::fabs(nd.z) < 1e-8) nd = rec.normal;
            att = w.albedo[m];
        } else if (w.type[m] == METAL) {                              // material.h:50-58
            D3 refl = reflect<false>(D, rec.normal);
            D3 ur = unit<false>(refl);
            nd = ur + scale(w.fuzz[m], ser_random_unit_vector(g));
            att = w.albedo[m];
            if (!(dot<false>(nd, rec.normal) > 0)) return {0, 0, 0};
        } else {                                                      // material.h:70-92
            att = {1, 1, 1};
            double ri = rec.front ? (1.0 / w.ri[m]) : w.ri[m];
            D3 ud = unit<false>(D);
            double cos_theta = std::fmin(dot<false>(-ud, rec.normal), 1.0);
            double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
            bool cannot = ri * sin_theta > 1.0;
            if (cannot || ser_reflectance(cos_theta, ri) > host_random<double>(g)) nd = reflect<false>(ud, rec.normal);
            else nd = refract<false>(ud, rec.normal, ri);
        }
        return att * ser_ray_color(rec.p, nd, depth - 1, w, g, st);   // camera.h:149
    }
    D3 ud = unit<false>(D);                                           // camera.h:153 -- CURRENT ray
    double a = 0.5 * (ud.y + 1.0);
    return scale(1.0 - a, D3{1.0, 1.0, 1.0}) + scale(a, D3{0.5, 0.7, 1.0});
}

inline int to_byte(double c) {                                        // color.h:29-47 (serial), main.cu:374-376
    double g = c > 0 ? std::sqrt(c) : 0;
    double cl = g < 0.000 ? 0.000 : (g > 0.999 ? 0.999 : g);
    return (int)(256 * cl);
}

// Renders rows [row0,row1) with stride `row_step`... the single rand() stream makes a row
// subset a different (equally valid) sample, so the byte-exact pin always renders all rows.
void render_serial_semantics(int scene_id, int W, int H, int S, int depth, std::string& p3, RenderStats& st) {
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
                pc = pc + ser_ray_color(org, ps - org, depth, w, g, st);
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

// CUDA-semantics render of rows [row0,row1) of a compact world given as arrays.
// stats4 = {primary_rays, segments, sphere_tests, 0}.
int oracle_render(int precision, int n, const void* center_radius, const void* albedo_fuzz, const void* ri, const int* type,
                  const int* cam_ints4, const void* cam_flat20, unsigned long long seed, int row0, int row1,
                  void* out_rgb, unsigned long long* stats4) {
    RenderStats st = {0, 0, 0, 0};
    if (precision == 32) {
        World<float> w; world_from_arrays(n, (const float*)center_radius, (const float*)albedo_fuzz, (const float*)ri, type, w);
        Camera<float> c; camera_from_flat(c, cam_ints4, (const float*)cam_flat20);
        render_cuda_semantics<float>(w, c, seed, row0, row1, (float*)out_rgb, st);
    } else if (precision == 64) {
        World<double> w; world_from_arrays(n, (const double*)center_radius, (const double*)albedo_fuzz, (const double*)ri, type, w);
        Camera<double> c; camera_from_flat(c, cam_ints4, (const double*)cam_flat20);
        render_cuda_semantics<double>(w, c, seed, row0, row1, (double*)out_rgb, st);
    } else return -1;
    if (stats4) { stats4[0] = st.primary_rays; stats4[1] = st.segments; stats4[2] = st.sphere_tests; stats4[3] = 0; }
    return 0;
}

// As oracle_render (fp32) plus the number of path segments (hit_world calls) of every pixel.
int oracle_render_segments_f32(int n, const float* center_radius, const float* albedo_fuzz, const float* ri, const int* type,
                               const int* cam_ints4, const float* cam_flat20, unsigned long long seed, int row0, int row1,
                               float* out_rgb, unsigned int* seg_per_pixel) {
    RenderStats st = {0, 0, 0, 0};
    World<float> w; world_from_arrays(n, center_radius, albedo_fuzz, ri, type, w);
    Camera<float> c; camera_from_flat(c, cam_ints4, cam_flat20);
    render_cuda_semantics<float>(w, c, seed, row0, row1, out_rgb, st, seg_per_pixel);
    return 0;
}

// Serial-semantics render; returns the P3 text length, copies up to cap bytes.
long long oracle_render_serial(int scene_id, int W, int H, int S, int depth, char* out, long long cap, unsigned long long* stats4) {
    std::string p3; RenderStats st = {0, 0, 0, 0};
    render_serial_semantics(scene_id, W, H, S, depth, p3, st);
    if (out && cap > 0) std::memcpy(out, p3.data(), (size_t)std::min<long long>(cap, (long long)p3.size()));
    if (stats4) { stats4[0] = st.primary_rays; stats4[1] = st.segments; stats4[2] = st.sphere_tests; stats4[3] = 0; }
    return (long long)p3.size();
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
double oracle_reflectance_f64(double cosine, double ri) { return dev_reflectance<double>(cosine, ri); }

}  // extern "C"
