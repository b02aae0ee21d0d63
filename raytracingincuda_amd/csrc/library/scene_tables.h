// scene_tables.h -- device tables of a scene: exact / shade tables (upload_scene), screening table, uniform-grid plan and blob
// Host side of librtiow_hip.so; part of the single translation unit rtiow_hip.hip (internal linkage).
#pragma once
#include "handle.h"

namespace {

template <class T>
int upload_scene(rtiow_handle_s* h, int n, const T* cr, const T* af, const T* ri, const int32_t* type, const int32_t* valid) {
    std::vector<T> ga, st;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (valid && !valid[i]) continue;
        const T cx = cr[4 * i], cy = cr[4 * i + 1], cz = cr[4 * i + 2], r = cr[4 * i + 3];
        if (type[i] < 0 || type[i] > 2) return fail_arg(h, RTIOW_E_BADARG, "material type out of range");
        ga.insert(ga.end(), {cx, cy, cz, (T)(r * r)});          // hittable.h:45 radius*radius
        // words 4..7: albedo and fuzz; a dielectric uses neither (material.h:70), its record carries Schlick's
        // r0^2 = ((1 - ri) / (1 + ri))^2 for ri = 1/eta (front face) and ri = eta (back face) instead, computed here
        // with the operations of material.h:62-66 in T (no contraction on the host either)
        auto r0sq = [](T ri_) { T r0 = ((T)1 - ri_) / ((T)1 + ri_); return (T)(r0 * r0); };
        const bool glass = type[i] == RTIOW_DIELECTRIC;
        st.insert(st.end(), {cx, cy, cz, (T)((T)1 / r),         // vec3.h:89-91 (1/t)*v
                             glass ? r0sq((T)((T)1 / ri[i])) : af[4 * i], glass ? r0sq(ri[i]) : af[4 * i + 1], af[4 * i + 2], af[4 * i + 3],
                             ri[i], (T)((T)1 / ri[i]),          // material.h:73 1.0f/refraction_index
                             (T)type[i], (T)0});
        ++m;
    }
    if (m == 0) return fail_arg(h, RTIOW_E_BADARG, "scene has no valid spheres");
    const int mp = (m + 4) / 4 * 4;                         // >= one padding entry: index m is the never-hit sphere that grid cells pad with
    for (int i = m; i < mp; ++i) ga.insert(ga.end(), {(T)0, (T)0, (T)0, (T)-1e12});   // c = |oc|^2 + 1e12 => disc < 0: never hit
    if (sizeof(T) == 4) {                                    // fp32: pair-interleave for v_pk_*_f32 (trip_discriminants)
        std::vector<T> pi(ga.size());
        for (int q = 0; q < mp / 2; ++q)
            for (int k = 0; k < 4; ++k) { pi[8 * q + 2 * k] = ga[8 * q + k]; pi[8 * q + 2 * k + 1] = ga[8 * q + 4 + k]; }
        ga.swap(pi);
    }
    h->host_cr.clear();
    for (int i = 0; i < n; ++i)
        if (!valid || valid[i]) for (int k = 0; k < 4; ++k) h->host_cr.push_back((double)cr[4 * i + k]);
    h->screen_dirty = true;
    void** bufs[] = {&h->geom_a, &h->shade_tbl};
    for (void** b : bufs) if (*b) { HIP_TRY(h, hipFree(*b)); *b = nullptr; }
    HIP_TRY(h, hipMalloc(&h->geom_a, sizeof(T) * 4 * mp));
    HIP_TRY(h, hipMalloc(&h->shade_tbl, sizeof(T) * 12 * m));
    HIP_TRY(h, hipMemcpy(h->geom_a, ga.data(), sizeof(T) * 4 * mp, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->shade_tbl, st.data(), sizeof(T) * 12 * m, hipMemcpyHostToDevice));
    h->n = m; h->n_padded = mp;
    h->stats.num_spheres = m;
    h->stats.scene_prepare_ms = 0;
    return 0;
}

// Builds the screening table of hit_world_screened for the current scene: centres recentred on
// the scene's centroid (ground-like spheres excluded), q' = |C'|^2 - r^2 - 2^-17 (|C'|^2 + r^2)
// rounded DOWN; always fp32 and pair-interleaved like the fp32 geom_a (the fp64 kernel screens in
// fp32 too).  2 Cmax goes to the kernel for the per-ray share of the margin.
template <class T>
int build_screen_table(rtiow_handle_s* h) {
    typedef float S;                                        // the screen runs in fp32 for both precisions
    const int m = h->n, mp = h->n_padded;
    const std::vector<double>& cr = h->host_cr;
    double ctr[3] = {0, 0, 0};
    int cnt = 0;
    for (int i = 0; i < m; ++i)
        if (cr[4 * i + 3] < 100.0) { for (int k = 0; k < 3; ++k) ctr[k] += cr[4 * i + k]; ++cnt; }
    if (cnt) for (int k = 0; k < 3; ++k) ctr[k] /= cnt;
    for (int k = 0; k < 3; ++k) h->ctr[k] = (double)(T)ctr[k];
    std::vector<S> lin((size_t)mp * 4);
    double cmax = 0;                                        // max |C'| over the spheres that are screened
    for (int i = 0; i < mp; ++i) {
        if (i >= m) { lin[4 * i] = lin[4 * i + 1] = lin[4 * i + 2] = 0; lin[4 * i + 3] = (S)1e12; continue; }   // padding: c~ huge => never a candidate
        double c2 = 0;
        for (int k = 0; k < 3; ++k) {
            const S cp = (S)(cr[4 * i + k] - h->ctr[k]);    // what the kernel will use as C'
            lin[4 * i + k] = cp;
            c2 += (double)cp * (double)cp;
        }
        if (std::sqrt(c2) > 64.0) { lin[4 * i + 3] = (S)-1e30; continue; }         // e.g. the ground: always re-tested exactly
        cmax = std::max(cmax, std::sqrt(c2));
        const double r = cr[4 * i + 3], r2 = r * r;
        const double kappa = std::ldexp(1.0, -17) * (c2 + r2);                     // the sphere's share of the margin
        S q = (S)(c2 - r2 - kappa);
        if ((double)q > c2 - r2 - kappa) q = std::nextafter(q, (S)-INFINITY);
        lin[4 * i + 3] = q;
    }
    h->omax2 = 2.0 * cmax * 1.000001;                       // per-ray share uses 2 Cmax |O'| + |O'|^2; the slack covers the raw sqrt (<= 2^-22 relative) in the kernel
    {
        std::vector<S> pi(lin.size());
        for (int q = 0; q < mp / 2; ++q)
            for (int k = 0; k < 4; ++k) { pi[8 * q + 2 * k] = lin[8 * q + k]; pi[8 * q + 2 * k + 1] = lin[8 * q + 4 + k]; }
        lin.swap(pi);
    }
    if (h->geom_s) { HIP_TRY(h, hipFree(h->geom_s)); h->geom_s = nullptr; }
    HIP_TRY(h, hipMalloc(&h->geom_s, lin.size() * sizeof(S)));
    HIP_TRY(h, hipMemcpy(h->geom_s, lin.data(), lin.size() * sizeof(S), hipMemcpyHostToDevice));
    h->screen_dirty = false;
    return 0;
}

// The plan of the uniform grid of hit_world_grid for a scene: pure host arithmetic (no GPU), shared by
// build_grid_tables and the rtiow_debug_grid_plan test hook.
//
//  small sphere    : registration half-width w_i = sqrt(r_i^2 + E_i) + eps <= cell / 2, where
//                    E_i = 18 * 2^-24 ((Rfar + Cmax)^2 + r_i^2) bounds the reference's discriminant
//                    noise for every origin within Rfar of the recentring point (hit_world_grid);
//  cell            : about one small sphere per cell, never narrower than the widest small sphere;
//  registration    : sphere i goes into every cell its square [c - w, c + w]^2 touches (<= 2 x 2),
//                    in index order; a sphere that meets a full cell (4 entries) joins the direct list;
//  direct list     : everything else (ground, big spheres, overflow), tested exactly by every ray.
// Candidate "small" sets: every finite sphere, then without the largest radii, and so on; each
// candidate whose cells are at least as wide as its widest member is registered, and the plan with the
// shortest direct list wins.  `usable` stays false when the grid would not pay (few small spheres,
// or a direct list that is no shorter than a fraction of the scene): the scene keeps the screened loop.
struct GridPlan {
    bool usable = false;
    int nx = 0, nz = 0, registered = 0;
    float cellf = 0, x0f = 0, z0f = 0;
    double rfar = 0, eps = 0, ylo = 1e300, yhi = -1e300, core_lo[3] = {1e300, 1e300, 1e300}, core_hi[3] = {-1e300, -1e300, -1e300}, rmax_g = 0, cmax_g = 0;
    std::vector<uint16_t> cells;        // [nz][nx][4]: sphere indices, 0xffff x4 = empty cell, index m = never-hit pad
    std::vector<int> direct;
    std::vector<double> halfwidth;      // w_i of the registered spheres (0 for the direct list)
};

GridPlan plan_grid(int m, const std::vector<double>& cr, const double* ctr) {
    GridPlan best;
    if (m < 24 || m > 60000) return best;
    std::vector<double> radii(m);
    for (int i = 0; i < m; ++i) radii[i] = cr[4 * i + 3];
    std::vector<int> small;
    for (int i = 0; i < m; ++i) {
        bool ok = radii[i] > 0 && std::isfinite(radii[i]);
        for (int k = 0; k < 3; ++k) ok = ok && std::isfinite(cr[4 * i + k]);
        if (ok) small.push_back(i);
    }
    std::sort(small.begin(), small.end(), [&](int a, int b) { return radii[a] < radii[b] || (radii[a] == radii[b] && a < b); });
    const double u18 = 18.0 * std::ldexp(1.0, -24) * 1.01;
    bool have = false;
    std::vector<double> w(m, 0.0);
    for (int attempt = 0; attempt < 12 && (int)small.size() >= 16; ++attempt) {
        double cmax = 0, lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, rmax = 0;
        for (int i : small) {
            double d2 = 0;
            for (int k = 0; k < 3; ++k) { const double d = cr[4 * i + k] - ctr[k]; d2 += d * d; lo[k] = std::min(lo[k], cr[4 * i + k]); hi[k] = std::max(hi[k], cr[4 * i + k]); }
            cmax = std::max(cmax, std::sqrt(d2));
            rmax = std::max(rmax, radii[i]);
        }
        const double cut = 0.9 * rmax;                                  // the next candidate drops the largest radii
        cmax *= 1.0001;
        const double rfar = std::max(64.0, 4.0 * cmax);
        double cabs = 0;
        for (int k = 0; k < 3; ++k) cabs = std::max(cabs, std::fabs(ctr[k]));
        const double L = 2.0 * (rfar + cmax) + cabs + rmax;            // every coordinate the walk handles is smaller
        const double eps = std::ldexp(L, -16);
        double wmax = 0;
        for (int i : small) {
            const double E = u18 * ((rfar + cmax) * (rfar + cmax) + radii[i] * radii[i]);
            w[i] = std::sqrt(radii[i] * radii[i] + E) + eps;
            wmax = std::max(wmax, w[i]);
        }
        const double ext_x = (hi[0] - lo[0]) + 2 * wmax, ext_z = (hi[2] - lo[2]) + 2 * wmax;
        double cell = std::sqrt(ext_x * ext_z / (double)small.size());
        if (L < 1e6 && cell >= 2.0 * (wmax + eps) * 1.02) {
            GridPlan pl;
            for (;;) {
                pl.nx = (int)std::ceil(ext_x / cell) + 1; pl.nz = (int)std::ceil(ext_z / cell) + 1;
                if ((long long)pl.nx * pl.nz <= 4096) break;
                cell *= 1.25;
            }
            pl.cellf = (float)cell; pl.rfar = rfar; pl.eps = eps;
            pl.x0f = (float)(lo[0] - wmax - 0.25 * cell); pl.z0f = (float)(lo[2] - wmax - 0.25 * cell);
            // registration against the cell edges the KERNEL will use (fp32 origin and width), widened by eps again
            auto cell_of = [&](double v, float origin) { return (int)std::floor((v - (double)origin) / (double)pl.cellf); };
            const int nx = pl.nx, nz = pl.nz;
            pl.cells.assign((size_t)nx * nz * 4, 0xffff);
            pl.halfwidth.assign(m, 0.0);
            std::vector<int> count((size_t)nx * nz, 0);
            std::vector<char> is_small(m, 0);
            for (int i : small) is_small[i] = 1;
            for (int i = 0; i < m; ++i) {
                if (!is_small[i]) { pl.direct.push_back(i); continue; }
                const double cx = cr[4 * i], cy = cr[4 * i + 1], cz = cr[4 * i + 2];
                const int ix0 = cell_of(cx - w[i] - eps, pl.x0f), ix1 = cell_of(cx + w[i] + eps, pl.x0f);
                const int iz0 = cell_of(cz - w[i] - eps, pl.z0f), iz1 = cell_of(cz + w[i] + eps, pl.z0f);
                bool fits = ix0 >= 0 && iz0 >= 0 && ix1 < nx && iz1 < nz;
                for (int iz = iz0; fits && iz <= iz1; ++iz)
                    for (int ix = ix0; ix <= ix1; ++ix) if (count[(size_t)iz * nx + ix] >= 4) fits = false;
                if (!fits) { pl.direct.push_back(i); continue; }
                for (int iz = iz0; iz <= iz1; ++iz)
                    for (int ix = ix0; ix <= ix1; ++ix) { const size_t c = (size_t)iz * nx + ix; pl.cells[4 * c + count[c]++] = (uint16_t)i; }
                ++pl.registered;
                pl.halfwidth[i] = w[i];
                pl.ylo = std::min(pl.ylo, cy - w[i]); pl.yhi = std::max(pl.yhi, cy + w[i]);
                const double c3[3] = {cx, cy, cz};
                double d2 = 0;
                for (int k = 0; k < 3; ++k) { pl.core_lo[k] = std::min(pl.core_lo[k], c3[k]); pl.core_hi[k] = std::max(pl.core_hi[k], c3[k]); const double d = c3[k] - ctr[k]; d2 += d * d; }
                pl.rmax_g = std::max(pl.rmax_g, radii[i]);
                pl.cmax_g = std::max(pl.cmax_g, std::sqrt(d2));
            }
            // a partly filled cell pads with index m, the never-hit entry behind the table (upload_scene)
            for (size_t c = 0; c < count.size(); ++c)
                if (count[c] > 0) for (int k = count[c]; k < 4; ++k) pl.cells[4 * c + k] = (uint16_t)m;
            if (pl.registered >= 16 && (!have || pl.direct.size() < best.direct.size())) { best = std::move(pl); have = true; }
        }
        while (!small.empty() && radii[small.back()] >= cut) small.pop_back();
    }
    best.usable = have && (int)best.direct.size() <= std::max(8, m / 6);
    return best;
}

// Builds the device tables of hit_world_grid for the current scene (after build_screen_table, whose
// recentring point the plan shares).  On return h->grid.use_grid says whether the scene has a grid.
template <class T>
int build_grid_tables(rtiow_handle_s* h) {
    GridParams& g = h->grid;
    g = GridParams{};
    if (h->grid_blob) { HIP_TRY(h, hipFree(h->grid_blob)); h->grid_blob = nullptr; }
    const int m = h->n;
    const std::vector<double>& cr = h->host_cr;
    const GridPlan best = plan_grid(m, cr, h->ctr);
    if (!best.usable) return 0;
    const std::vector<uint16_t>& cells = best.cells;
    const std::vector<int>& direct = best.direct;
    const int nx = best.nx, nz = best.nz, registered = best.registered;
    const float cellf = best.cellf, x0f = best.x0f, z0f = best.z0f;
    const double rfar = best.rfar, ylo = best.ylo, yhi = best.yhi, rmax_g = best.rmax_g, cmax_g = best.cmax_g;
    const double* core_lo = best.core_lo; const double* core_hi = best.core_hi;
    // ---- blob: cells | aos (fp32 only) | direct table | direct ids
    const int nd = (int)direct.size(), ndp = (nd + 3) / 4 * 4;
    std::vector<T> dtab((size_t)ndp * 4);
    std::vector<int> ids(ndp, m);
    for (int k = 0; k < ndp; ++k) {
        if (k < nd) {
            const int i = direct[k];
            const T r = (T)cr[4 * i + 3];
            dtab[4 * k] = (T)cr[4 * i]; dtab[4 * k + 1] = (T)cr[4 * i + 1]; dtab[4 * k + 2] = (T)cr[4 * i + 2]; dtab[4 * k + 3] = (T)(r * r);   // as upload_scene
            ids[k] = i;
        } else { dtab[4 * k] = dtab[4 * k + 1] = dtab[4 * k + 2] = (T)0; dtab[4 * k + 3] = (T)-1e12; }
    }
    if (sizeof(T) == 4) {                                    // pair-interleave like geom_a (trip_discriminants)
        std::vector<T> pi(dtab.size());
        for (int q = 0; q < ndp / 2; ++q)
            for (int k = 0; k < 4; ++k) { pi[8 * q + 2 * k] = dtab[8 * q + k]; pi[8 * q + 2 * k + 1] = dtab[8 * q + 4 + k]; }
        dtab.swap(pi);
    }
    std::vector<float> aos;
    if (sizeof(T) == 4) {
        aos.resize((size_t)(m + 1) * 4);
        for (int i = 0; i < m; ++i) {
            const float r = (float)cr[4 * i + 3];
            aos[4 * i] = (float)cr[4 * i]; aos[4 * i + 1] = (float)cr[4 * i + 1]; aos[4 * i + 2] = (float)cr[4 * i + 2]; aos[4 * i + 3] = r * r;
        }
        aos[4 * m] = aos[4 * m + 1] = aos[4 * m + 2] = 0.0f; aos[4 * m + 3] = -1e12f;
    }
    auto align16 = [](size_t v) { return (v + 15) / 16 * 16; };
    const size_t cells_bytes = align16(cells.size() * sizeof(uint16_t));
    const size_t aos_bytes = align16(aos.size() * sizeof(float));
    const size_t dtab_bytes = align16(dtab.size() * sizeof(T));
    const size_t ids_bytes = align16(ids.size() * sizeof(int));
    std::vector<unsigned char> blob(cells_bytes + aos_bytes + dtab_bytes + ids_bytes, 0);
    std::memcpy(blob.data(), cells.data(), cells.size() * sizeof(uint16_t));
    if (!aos.empty()) std::memcpy(blob.data() + cells_bytes, aos.data(), aos.size() * sizeof(float));
    std::memcpy(blob.data() + cells_bytes + aos_bytes, dtab.data(), dtab.size() * sizeof(T));
    std::memcpy(blob.data() + cells_bytes + aos_bytes + dtab_bytes, ids.data(), ids.size() * sizeof(int));
    HIP_TRY(h, hipMalloc(&h->grid_blob, blob.size()));
    HIP_TRY(h, hipMemcpy(h->grid_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    h->grid_cells_bytes = (int)cells_bytes; h->grid_aos_bytes = (int)aos_bytes; h->grid_direct_bytes = (int)dtab_bytes; h->grid_ids_bytes = (int)ids_bytes;
    h->grid_direct = nd; h->grid_registered = registered;
    g.use_grid = 1;
    g.nx = nx; g.nz = nz;
    g.x0 = x0f; g.z0 = z0f; g.cell = cellf; g.inv_cell = (float)(1.0 / (double)cellf);
    g.ylo = std::nextafter((float)ylo, -INFINITY); g.yhi = std::nextafter((float)yhi, INFINITY);
    g.far2 = (float)(rfar * rfar * 0.999);
    for (int k = 0; k < 3; ++k) { g.core_lo[k] = std::nextafter((float)core_lo[k], -INFINITY); g.core_hi[k] = std::nextafter((float)core_hi[k], INFINITY); }
    g.rmax2 = (float)(rmax_g * rmax_g * 1.0001);
    g.cmax = (float)(cmax_g * 1.0001);
    g.n_direct_padded = ndp;
    g.blob = (const unsigned char*)h->grid_blob;
    g.blob_bytes = (int)blob.size();
    return 0;
}

template <class T>
void fill_screen_params(RenderParams<T>& p, const rtiow_handle_s* h) {
    p.screen.geom_s = (const float*)h->geom_s;
    p.use_screen = ((h->scene_source == RTIOW_SCENE_LDS || h->scene_source == RTIOW_SCENE_GRID) && h->geom_s) ? 1 : 0;
    p.screen.ctr_x = (T)h->ctr[0]; p.screen.ctr_y = (T)h->ctr[1]; p.screen.ctr_z = (T)h->ctr[2]; p.screen.omax2 = (T)h->omax2;
}

}  // namespace
