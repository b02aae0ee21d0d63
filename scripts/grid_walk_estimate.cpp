// Design probe (CPU only, statistics, NOT bit-exact and not part of the product or the oracle):
// how long would a per-lane uniform-grid walk be, per wave-iteration of the flattened render loop?
//
// A wave of 64 lanes runs 64 pixels of an 8x8 tile in lockstep, one path segment per iteration
// (the kernel's structure).  For every iteration this records the MAXIMUM over the live lanes of
//   * the number of xz grid cells the ray crosses while inside the y-slab of the small spheres,
//   * the number of 4-sphere trips those cells hold,
// which is what a SIMT grid walk would cost, next to the brute-force trip count (N/4).
//
//   g++ -O2 -o /tmp/grid_walk scripts/grid_walk_estimate.cpp && /tmp/grid_walk 3 1.0
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

struct V { double x, y, z; };
static V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V operator*(double t, V a) { return {t * a.x, t * a.y, t * a.z}; }
static double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V unit(V a) { return (1.0 / std::sqrt(dot(a, a))) * a; }
static V cross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct Sphere { V c; double r; int type; double fuzz, ri; };
static std::mt19937_64 gen(12345);
static double rnd() { return std::uniform_real_distribution<double>(0.0, 1.0)(gen); }

static std::vector<Sphere> build(int scene) {
    std::vector<Sphere> s;
    s.push_back({{0, -1000, 0}, 1000, 0, 0, 0});
    int a0 = -11, a1 = 11, b0 = -11, b1 = 11;
    if (scene == 2) { a0 = 5; a1 = 11; b0 = 5; b1 = 11; }
    if (scene == 3) { a0 = -11; a1 = 0; b0 = -11; b1 = 0; }
    for (int a = a0; a < a1; ++a)
        for (int b = b0; b < b1; ++b) {
            double m = rnd();
            V c = {a + 0.9 * rnd(), 0.2, b + 0.9 * rnd()};
            V d = c - V{4, 0.2, 0};
            if (std::sqrt(dot(d, d)) > 0.9) s.push_back({c, 0.2, m < 0.8 ? 0 : (m < 0.95 ? 1 : 2), 0.5 * rnd(), 1.5});
        }
    s.push_back({{0, 1, 0}, 1, 2, 0, 1.5});
    s.push_back({{-4, 1, 0}, 1, 0, 0, 0});
    s.push_back({{4, 1, 0}, 1, 1, 0, 0});
    return s;
}

struct Grid {
    double x0, z0, cell, ylo, yhi; int nx, nz;
    std::vector<std::vector<int>> cells;
    std::vector<int> big;
};

static Grid make_grid(const std::vector<Sphere>& s, double cell, double margin) {
    Grid g; g.cell = cell;
    double xl = 1e30, xh = -1e30, zl = 1e30, zh = -1e30; g.ylo = 1e30; g.yhi = -1e30;
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i].r > 0.5 * cell) { g.big.push_back((int)i); continue; }
        double R = s[i].r + margin;
        xl = std::min(xl, s[i].c.x - R); xh = std::max(xh, s[i].c.x + R);
        zl = std::min(zl, s[i].c.z - R); zh = std::max(zh, s[i].c.z + R);
        g.ylo = std::min(g.ylo, s[i].c.y - R); g.yhi = std::max(g.yhi, s[i].c.y + R);
    }
    g.x0 = xl; g.z0 = zl; g.nx = (int)std::ceil((xh - xl) / cell); g.nz = (int)std::ceil((zh - zl) / cell);
    g.cells.assign((size_t)g.nx * g.nz, {});
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i].r > 0.5 * cell) continue;
        double R = s[i].r + margin;
        int ix0 = (int)std::floor((s[i].c.x - R - xl) / cell), ix1 = (int)std::floor((s[i].c.x + R - xl) / cell);
        int iz0 = (int)std::floor((s[i].c.z - R - zl) / cell), iz1 = (int)std::floor((s[i].c.z + R - zl) / cell);
        for (int ix = std::max(ix0, 0); ix <= std::min(ix1, g.nx - 1); ++ix)
            for (int iz = std::max(iz0, 0); iz <= std::min(iz1, g.nz - 1); ++iz) g.cells[(size_t)iz * g.nx + ix].push_back((int)i);
    }
    return g;
}

// cells crossed by the ray inside slab and grid box, and the trips (ceil(count/4)) they hold
static void walk(const Grid& g, V O, V D, int& ncells, int& ntrips) {
    ncells = 0; ntrips = 0;
    double t0 = 0.0, t1 = 1e30;
    const double lo[3] = {g.x0, g.ylo, g.z0}, hi[3] = {g.x0 + g.nx * g.cell, g.yhi, g.z0 + g.nz * g.cell};
    const double o[3] = {O.x, O.y, O.z}, d[3] = {D.x, D.y, D.z};
    for (int k = 0; k < 3; ++k) {
        if (d[k] == 0) { if (o[k] < lo[k] || o[k] > hi[k]) return; continue; }
        double a = (lo[k] - o[k]) / d[k], b = (hi[k] - o[k]) / d[k];
        if (a > b) std::swap(a, b);
        t0 = std::max(t0, a); t1 = std::min(t1, b);
    }
    if (t0 > t1) return;
    V p = O + t0 * D, q = O + t1 * D;
    int ix = std::min(std::max((int)std::floor((p.x - g.x0) / g.cell), 0), g.nx - 1);
    int iz = std::min(std::max((int)std::floor((p.z - g.z0) / g.cell), 0), g.nz - 1);
    int jx = std::min(std::max((int)std::floor((q.x - g.x0) / g.cell), 0), g.nx - 1);
    int jz = std::min(std::max((int)std::floor((q.z - g.z0) / g.cell), 0), g.nz - 1);
    ncells = std::abs(jx - ix) + std::abs(jz - iz) + 1;
    // trips along an (approximate) DDA: step through the cells of the straight line
    int cx = ix, cz = iz;
    const int sx = D.x > 0 ? 1 : -1, sz = D.z > 0 ? 1 : -1;
    double tx = D.x != 0 ? ((g.x0 + (cx + (sx > 0)) * g.cell) - O.x) / D.x : 1e30;
    double tz = D.z != 0 ? ((g.z0 + (cz + (sz > 0)) * g.cell) - O.z) / D.z : 1e30;
    const double dx = D.x != 0 ? g.cell / std::fabs(D.x) : 1e30, dz = D.z != 0 ? g.cell / std::fabs(D.z) : 1e30;
    for (int k = 0; k < ncells; ++k) {
        if (cx < 0 || cz < 0 || cx >= g.nx || cz >= g.nz) break;
        ntrips += (int)(g.cells[(size_t)cz * g.nx + cx].size() + 3) / 4;
        if (tx < tz) { cx += sx; tx += dx; } else { cz += sz; tz += dz; }
    }
}

static bool hit(const std::vector<Sphere>& s, V O, V D, double& t, int& idx) {
    t = 1e30; idx = -1;
    const double a = dot(D, D);
    for (size_t i = 0; i < s.size(); ++i) {
        V oc = s[i].c - O;
        double h = dot(D, oc), c = dot(oc, oc) - s[i].r * s[i].r, disc = h * h - a * c;
        if (disc < 0) continue;
        double sq = std::sqrt(disc), r = (h - sq) / a;
        if (r <= 0.001 || r >= t) { r = (h + sq) / a; if (r <= 0.001 || r >= t) continue; }
        t = r; idx = (int)i;
    }
    return idx >= 0;
}

static V ruv() { for (;;) { V p = {2 * rnd() - 1, 2 * rnd() - 1, 2 * rnd() - 1}; double l = dot(p, p); if (l > 1e-12 && l <= 1) return (1 / std::sqrt(l)) * p; } }

struct Lane { V O, D; int depth, sample; bool alive; int px, py; };

int main(int argc, char** argv) {
    const int scene = argc > 1 ? atoi(argv[1]) : 3;
    const double cell = argc > 2 ? atof(argv[2]) : 1.0;
    const int W = 480, H = 270, S = argc > 3 ? atoi(argv[3]) : 8, B = 50;
    std::vector<Sphere> s = build(scene);
    Grid g = make_grid(s, cell, 0.02);
    size_t reg = 0, maxc = 0; for (auto& c : g.cells) { reg += c.size(); maxc = std::max(maxc, c.size()); }
    printf("scene %d: %zu spheres, grid %dx%d cell %.2f, %zu registrations (max %zu per cell), %zu big, slab y [%.2f, %.2f]\n",
           scene, s.size(), g.nx, g.nz, cell, reg, maxc, g.big.size(), g.ylo, g.yhi);
    // camera (main.cu:114-121)
    V from = {13, 2, 3}, at = {0, 0, 0}, vup = {0, 1, 0};
    double theta = 20 * M_PI / 180, hh = std::tan(theta / 2), focus = 10, vh = 2 * hh * focus, vw = vh * W / H;
    V w = unit(from - at), u = unit(cross(vup, w)), v = cross(w, u);
    V du = (vw / W) * u, dv = (-vh / H) * v;
    V p00 = from - focus * w - 0.5 * (vw * u) - 0.5 * ((-vh) * (-1.0 * v)) ;
    p00 = from - focus * w - (vw / 2) * u + (vh / 2) * v + 0.5 * (du + dv);
    double dr = focus * std::tan(0.3 * M_PI / 180);
    double it = 0, sum_maxcells = 0, sum_maxtrips = 0, sum_meancells = 0, live = 0, sum_sumtrips = 0;
    std::vector<double> hist(64, 0);
    for (int ty = 0; ty < H / 8; ++ty)
        for (int tx = 0; tx < W / 8; ++tx) {
            Lane L[64];
            for (int k = 0; k < 64; ++k) { L[k].alive = true; L[k].sample = 0; L[k].depth = -1; L[k].px = tx * 8 + (k & 7); L[k].py = ty * 8 + (k >> 3); }
            for (;;) {
                int nalive = 0, maxcells = 0, maxtrips = 0; double sc = 0, st = 0;
                for (int k = 0; k < 64; ++k) {
                    Lane& l = L[k];
                    if (!l.alive) continue;
                    if (l.depth < 0) {
                        double ox = rnd() - 0.5, oy = rnd() - 0.5, a, b;
                        do { a = 2 * rnd() - 1; b = 2 * rnd() - 1; } while (a * a + b * b >= 1);
                        V ps = p00 + (l.px + ox) * du + (l.py + oy) * dv;
                        l.O = from + (a * dr) * u + (b * dr) * v; l.D = ps - l.O; l.depth = 0;
                    }
                    ++nalive;
                    int nc, nt; walk(g, l.O, l.D, nc, nt);
                    maxcells = std::max(maxcells, nc); maxtrips = std::max(maxtrips, nt); sc += nc; st += nt;
                    double t; int idx; bool end = false;
                    if (!hit(s, l.O, l.D, t, idx)) end = true;
                    else {
                        V P = l.O + t * l.D, n = (1 / s[idx].r) * (P - s[idx].c);
                        bool front = dot(l.D, n) < 0; if (!front) n = -1.0 * n;
                        if (s[idx].type == 0) { l.D = n + ruv(); }
                        else if (s[idx].type == 1) { V r = unit(l.D - 2 * dot(l.D, n) * n) + s[idx].fuzz * ruv(); if (dot(r, n) <= 0) end = true; l.D = r; }
                        else {
                            double ri = front ? 1 / 1.5 : 1.5; V ud = unit(l.D); double ct = std::min(-dot(ud, n), 1.0), stt = std::sqrt(1 - ct * ct);
                            double r0 = (1 - ri) / (1 + ri); r0 *= r0;
                            if (ri * stt > 1 || r0 + (1 - r0) * std::pow(1 - ct, 5) > rnd()) l.D = ud - 2 * dot(ud, n) * n;
                            else { V perp = ri * (ud + ct * n); l.D = perp + (-std::sqrt(std::fabs(1 - dot(perp, perp)))) * n; }
                        }
                        l.O = P;
                        if (++l.depth >= B) end = true;
                    }
                    if (end) { l.depth = -1; if (++l.sample >= S) l.alive = false; }
                }
                if (!nalive) break;
                it += 1; sum_maxcells += maxcells; sum_maxtrips += maxtrips; sum_meancells += sc / nalive; live += nalive; sum_sumtrips += st;
                hist[std::min(maxcells, 63)] += 1;
            }
        }
    printf("wave-iterations %.0f, mean live lanes %.1f\n", it, live / it);
    printf("per wave-iteration: max cells %.2f, max trips %.2f, mean-lane cells %.2f, mean-lane trips %.2f; brute force trips %zu\n",
           sum_maxcells / it, sum_maxtrips / it, sum_meancells / it, sum_sumtrips / live, (s.size() + 3) / 4);
    printf("histogram of max cells per wave-iteration:");
    for (int k = 0; k < 64; ++k) if (hist[k] > 0) printf(" %d:%.3f", k, hist[k] / it);
    printf("\n");
    return 0;
}
