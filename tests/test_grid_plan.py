"""The uniform-grid plan of RTIOW_SCENE_GRID (rtiow_debug_grid_plan: host arithmetic of librtiow_hip.so, no
GPU): the invariants hit_world_grid's exactness argument rests on, re-derived here with numpy.

  * every sphere is either on the direct list or registered, never both; the reference scenes put exactly
    the ground and the three unit spheres on the direct list;
  * a registered sphere sits in EVERY cell its square [c - w, c + w] touches (cell edges computed from the
    fp32 origin and width the kernel uses), at most four spheres share a cell, pads are index n;
  * its half-width w covers the reference's discriminant noise for origins within Rfar:
    w >= sqrt(r^2 + 18 * 2^-24 ((Rfar + Cmax)^2 + r^2)) + eps, eps = 2^-16 of the largest coordinate in play;
  * the grid box and the slab contain every registered square;
  * scenes the grid does not suit (too few, too dense) are declined.
"""
import ctypes

import numpy as np
import pytest


def _plan(native, cr, centre=None):
    lib = native.load_hip_library(debug=True)       # the plan is a test hook (include/rtiow_debug.h): the test build exports it
    cr = np.ascontiguousarray(cr, np.float64)
    n = len(cr)
    if centre is None:                                    # the library's recentring point: centroid of the spheres with r < 100
        centre = cr[cr[:, 3] < 100.0, :3].mean(axis=0)
    centre = np.ascontiguousarray(centre, np.float64)
    dims = (ctypes.c_int32 * 4)()
    params = (ctypes.c_double * 8)()
    cells = np.zeros(4096 * 4, np.uint16)
    direct = np.zeros(n, np.int32)
    hw = np.zeros(n, np.float64)
    lib.rtiow_debug_grid_plan.restype = ctypes.c_int
    lib.rtiow_debug_grid_plan.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    rc = lib.rtiow_debug_grid_plan(n, cr.ctypes.data, centre.ctypes.data, ctypes.addressof(dims), ctypes.addressof(params),
                                   cells.ctypes.data, cells.size, direct.ctypes.data, direct.size, hw.ctypes.data)
    assert rc >= 0
    nx, nz, registered, nd = list(dims)
    keys = ("x0", "z0", "cell", "ylo", "yhi", "rfar", "eps", "cmax")
    return {"usable": rc == 1, "nx": nx, "nz": nz, "registered": registered, "direct": direct[:nd].copy(),
            "cells": cells[:nx * nz * 4].reshape(nz, nx, 4).copy(), "hw": hw, "centre": centre, **dict(zip(keys, params))}


def _check_invariants(cr, pl):
    n = len(cr)
    cells, hw = pl["cells"], pl["hw"]
    direct = set(int(i) for i in pl["direct"])
    in_cells = set(int(i) for i in cells.ravel() if i < n)
    assert direct.isdisjoint(in_cells) and direct | in_cells == set(range(n))
    assert len(in_cells) == pl["registered"] and (hw[list(in_cells)] > 0).all() and (hw[list(direct)] == 0).all()
    # cells: at most 4, filled from slot 0, pads = n, empty = 0xffff x 4
    for rec in cells.reshape(-1, 4):
        real = rec[rec < n]
        if len(real) == 0:
            assert (rec == 0xFFFF).all()
        else:
            assert (rec[:len(real)] < n).all() and (rec[len(real):] == n).all() and len(set(real)) == len(real)
    # coverage with the kernel's fp32 cell edges
    x0, z0, cell = np.float32(pl["x0"]), np.float32(pl["z0"]), np.float32(pl["cell"])
    for i in in_cells:
        cx, cy, cz, r = cr[i]
        w = hw[i]
        E = 18.0 * 2.0 ** -24 * ((pl["rfar"] + pl["cmax"]) ** 2 + r * r)
        assert w >= np.sqrt(r * r + E) + pl["eps"] * 0.999, (i, w)
        assert 2 * w <= float(cell)
        ix0, ix1 = int(np.floor((cx - w - float(x0)) / float(cell))), int(np.floor((cx + w - float(x0)) / float(cell)))
        iz0, iz1 = int(np.floor((cz - w - float(z0)) / float(cell))), int(np.floor((cz + w - float(z0)) / float(cell)))
        assert 0 <= ix0 <= ix1 < pl["nx"] and 0 <= iz0 <= iz1 < pl["nz"], i            # the box contains the square
        for iz in range(iz0, iz1 + 1):
            for ix in range(ix0, ix1 + 1):
                assert i in cells[iz, ix], (i, ix, iz)
        assert pl["ylo"] <= cy - w and cy + w <= pl["yhi"]
    # eps: 2^-16 of a bound on every coordinate the walk handles
    L = 2 * (pl["rfar"] + pl["cmax"]) + np.abs(pl["centre"]).max()
    assert pl["eps"] >= L * 2.0 ** -16 * 0.999 and pl["rfar"] >= 64.0 and pl["rfar"] >= 4.0 * pl["cmax"] * 0.999


@pytest.mark.parametrize("scene_id,expect_cells", [(1, (21, 25)), (2, (6, 9)), (3, (11, 14))])
def test_reference_scenes(native, oracle, scene_id, expect_cells):
    from tests.conftest import compact
    sc = compact(oracle.build_scene(scene_id, 64))
    cr = sc["center_radius"]
    pl = _plan(native, cr)
    assert pl["usable"]
    n = len(cr)
    assert sorted(pl["direct"]) == [0, n - 3, n - 2, n - 1]              # the ground and the three unit spheres (main.cu:290-296)
    assert pl["registered"] == n - 4 and expect_cells[0] <= pl["nx"] <= expect_cells[1] and expect_cells[0] <= pl["nz"] <= expect_cells[1]
    assert 0.9 < pl["cell"] < 1.3 and -0.1 < pl["ylo"] < 0.0 and 0.4 < pl["yhi"] < 0.5   # the small spheres sit at y = 0.2, r = 0.2
    _check_invariants(cr, pl)
    # fp32 tables give the same plan (the library plans from the values it uploaded)
    pl32 = _plan(native, compact(oracle.build_scene(scene_id, 32))["center_radius"])
    assert pl32["usable"] and (pl32["nx"], pl32["nz"], pl32["registered"]) == (pl["nx"], pl["nz"], pl["registered"])


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes(native, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(30, 900))
    half = float(rng.uniform(2, 40))
    off = rng.uniform(-50, 50, 3) if seed % 3 == 0 else np.zeros(3)
    r = [np.full(n, 0.2), rng.uniform(0.02, 0.6, n), np.exp(rng.uniform(np.log(0.005), np.log(2.0), n)), rng.choice([0.1, 0.45, 1.0], n)][seed % 4]
    cr = np.column_stack([rng.uniform(-half, half, n) + off[0], r + rng.uniform(0, 3, n) * (seed % 2) + off[1], rng.uniform(-half, half, n) + off[2], r])
    cr = np.vstack([[off[0], off[1] - 1000.0, off[2], 1000.0], cr])
    pl = _plan(native, cr)
    if pl["usable"]:
        _check_invariants(cr, pl)
        assert len(pl["direct"]) <= max(8, len(cr) // 6) and pl["registered"] >= 16
    else:
        assert pl["registered"] == 0 or len(pl["direct"]) > max(8, len(cr) // 6)


def test_declines_unsuitable_scenes(native):
    rng = np.random.default_rng(0)
    few = np.column_stack([rng.uniform(-5, 5, 20), np.full(20, 0.2), rng.uniform(-5, 5, 20), np.full(20, 0.2)])
    assert not _plan(native, few)["usable"]                                   # fewer than 24 spheres
    dense = np.column_stack([rng.uniform(-1, 1, 400), np.full(400, 0.2), rng.uniform(-1, 1, 400), np.full(400, 0.2)])
    assert not _plan(native, dense)["usable"]                                 # 100 spheres per unit area: every cell overflows
    bad = np.column_stack([rng.uniform(-5, 5, 60), np.full(60, 0.2), rng.uniform(-5, 5, 60), np.full(60, 0.2)])
    bad[5, 0] = np.nan; bad[6, 3] = -0.2; bad[7, 3] = np.inf
    pl = _plan(native, bad, centre=np.zeros(3))
    if pl["usable"]:
        assert {5, 6, 7} <= set(int(i) for i in pl["direct"])                 # non-finite or non-positive spheres are never gridded
