"""Host-side half of the drop-in (no GPU): scene tables, camera, PPM writer, file naming,
CLI behaviour, and that the C-ABI libraries load and export every declared symbol."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtiow_[a-z0-9_]+)\s*\(", text)))


def test_c_abi_exports_every_declared_symbol(native):
    from raytracingincuda_amd import api
    paths = native.lib_paths()
    hip_decl, host_decl = _declared("rtiow.h"), _declared("rtiow_host.h")
    dbg_decl = [d for d in _declared("rtiow_debug.h") if d.startswith("rtiow_debug_") and d not in ("rtiow_debug_region_cycles", "rtiow_debug_path_stats")]   # those two: instrumented builds only
    assert sorted(api.HIP_SYMBOLS) == hip_decl
    assert sorted(api.HOST_SYMBOLS) == host_decl
    assert sorted(api.DEBUG_SYMBOLS) == dbg_decl and not [d for d in hip_decl if "debug" in d]
    # dlopen both (no compute call: there is no GPU here) and resolve each symbol
    hip = ctypes.CDLL(paths["hip"])
    host = ctypes.CDLL(paths["host"])
    for s in hip_decl:
        assert getattr(hip, s) is not None
    for s in host_decl:
        assert getattr(host, s) is not None
    assert hip.rtiow_abi_version() == native.ABI_VERSION == 6
    # exported from the shared objects with C linkage
    syms = subprocess.run(["nm", "-D", "--defined-only", paths["hip"]], capture_output=True, text=True, check=True).stdout
    for s in hip_decl:
        assert re.search(r"\bT %s\b" % s, syms), s
    # test hooks ship in the test build only: the product exports none, the debug build exports the product's ABI plus every hook
    assert "rtiow_debug_" not in syms
    dsyms = subprocess.run(["nm", "-D", "--defined-only", paths["hip_debug"]], capture_output=True, text=True, check=True).stdout
    for s in hip_decl + dbg_decl:
        assert re.search(r"\bT %s\b" % s, dsyms), s
    # ... and nothing but the ABI: the library is linked into someone else's main.cu (main.cu:335), so no helper function, global
    # or libstdc++ instantiation may leak out of it (csrc/librtiow_hip.map); the host library's names are rtiow_host_* likewise
    for table, decl in ((syms, hip_decl), (dsyms, hip_decl + dbg_decl)):
        exported = [l.split()[-1] for l in table.splitlines() if l.strip()]
        assert sorted(exported) == sorted(decl), sorted(set(exported) ^ set(decl))
    hsyms = subprocess.run(["nm", "-D", "--defined-only", paths["host"]], capture_output=True, text=True, check=True).stdout
    stray = [l.split()[-1] for l in hsyms.splitlines() if l.split()[1] in "TBD" and not l.split()[-1].startswith("rtiow_host_")]
    assert not stray, stray


def test_no_gpu_means_loud_failure_not_fallback(native):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(native.RtiowError):
        native.Renderer(0, 32)


def test_product_code_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "raytracingincuda_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in text and "oracle_lib" not in text and "import oracle" not in text, f


def test_host_scene_equals_oracle_bitwise(native, oracle):
    for prec in (32, 64):
        for sid in (1, 2, 3, 0, -5, 99):
            a, b = native.build_scene(sid, prec), oracle.build_scene(sid, prec)
            assert native.scene_slots(sid) == len(b["type"])
            for k in ("center_radius", "albedo_fuzz", "refraction_index", "type", "valid"):
                assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), (prec, sid, k)
    sc = native.compact_scene(native.build_scene(1, 32))
    assert len(sc["type"]) == 487 and sc["valid"].all()


def test_host_camera_equals_oracle_bitwise(native, oracle):
    from tests.oracle_lib import Oracle
    for prec in (32, 64):
        for cfg in [(320, 192, 10, 25), (1280, 720, 100, 50), (1920, 1080, 100, 50), (1920, 1080, 500, 50), (7, 5, 1, 1)]:
            cam = native.camera(prec, *cfg)
            ints, flat = Oracle.camera_to_flat(cam, prec)
            oints, oflat = oracle.camera_flat(prec, *cfg)
            assert np.array_equal(ints, oints) and np.array_equal(flat.view(np.uint8), oflat.view(np.uint8))
    with pytest.raises(native.RtiowError):
        native.camera(32, 0, 10, 1, 1)


def test_ppm_filename_matches_reference_pattern(native):
    # main.cu:349-358 and GlobalDouble main.cu:351
    assert native.ppm_filename(32, 1, 320, 192, 10, 25, 8) == "global_float_scene1_320x192_10samples_25bounces_8threadsPerBlockRow.ppm"
    assert native.ppm_filename(64, 3, 1920, 1080, 500, 50, 16) == "global_double_scene3_1920x1080_500samples_50bounces_16threadsPerBlockRow.ppm"


def test_ppm_writer_semantics(native, tmp_path, oracle):
    # int(256*clamp(c,0,0.999)), "r g b\n", header "P3\nW H\n255\n" (main.cu:367-379)
    for dt in (np.float32, np.float64):
        img = np.array([[[0.0, 0.5, 1.0], [0.999, 0.9989, -0.25]], [[2.0, 1e-9, 0.25], [0.00390625, 0.0039, 0.7]]], dt)
        text = native.format_ppm(img)
        assert text == b"P3\n2 2\n255\n0 128 255\n255 255 0\n255 0 64\n1 0 179\n"
        path = str(tmp_path / "x.ppm")
        native.write_ppm(path, img)
        assert open(path, "rb").read() == text
        # binary twin: same levels, one byte per channel; ppm_diff reads both as equal
        native.write_ppm(path + "6", img, binary=True)
        assert open(path + "6", "rb").read() == b"P6\n2 2\n255\n" + bytes([0, 128, 255, 255, 255, 0, 255, 0, 64, 1, 0, 179])
    # same quantisation as the reference serial writer on a real image (color.h:40-43)
    p3, _ = oracle.render_serial(3, 32, 18, 2, 5)
    vals = np.array(p3.split()[4:], np.int64)
    assert vals.min() >= 0 and vals.max() <= 255
    with pytest.raises(native.RtiowError):
        native.write_ppm(str(tmp_path / "no_such_dir" / "x.ppm"), np.zeros((1, 1, 3), np.float32))


def test_ppm_writer_threaded_ranges_keep_the_text(native, tmp_path):
    """Frames of 65536 pixels or more are formatted on several threads, range by range, from buffers sized for levels
    0..255; a range that holds a NaN level ("-2147483648") is formatted again in the long form.  Same bytes as a
    per-channel restatement."""
    rng = np.random.default_rng(3)
    for dt in (np.float32, np.float64):
        img = rng.uniform(-0.2, 1.2, (301, 257, 3)).astype(dt)
        img[0, 0, 0] = np.nan; img[17, 5, 1] = np.nan; img[300, 256, 2] = np.nan; img[150, :, :] = np.nan   # first, last and a whole row
        with np.errstate(invalid="ignore"):
            lv = (dt(256) * np.clip(img, dt(0), dt(0.999))).astype(np.float64)
        lv = np.where(np.isnan(img), -2147483648, np.trunc(np.nan_to_num(lv))).astype(np.int64).reshape(-1, 3)
        want = ("P3\n257 301\n255\n" + "".join("%d %d %d\n" % tuple(r) for r in lv)).encode()
        assert native.format_ppm(img) == want
        path = str(tmp_path / "big.ppm")
        native.write_ppm(path, img)
        assert open(path, "rb").read() == want


def test_level_writers_are_byte_identical_to_the_float_writers(native, tmp_path):
    """The drop-in executable quantises on the device and writes the file from one byte per channel (rtiow_read_levels +
    rtiow_host_write_ppm_levels, main.cu:364-379): same bytes as the writers that take the T framebuffer, text and binary, on frames
    small (one thread) and large (16 formatting threads, each writing its own range of the file), incl. every level 0..255."""
    rng = np.random.default_rng(11)
    for dt in (np.float32, np.float64):
        for shape in ((3, 5, 3), (301, 257, 3), (540, 960, 3)):
            img = rng.uniform(-0.2, 1.2, shape).astype(dt)
            flat = img.reshape(-1)
            flat[:min(256, flat.size)] = ((np.arange(256) + 0.5) / 256.0)[:min(256, flat.size)]
            lev, nans = native.levels(img)
            assert nans == 0 and lev.dtype == np.uint8 and lev.shape == img.shape
            assert np.array_equal(lev, np.trunc(dt(256) * np.clip(img, dt(0), dt(0.999))).astype(np.uint8))
            for binary in (False, True):
                a, b = str(tmp_path / "t.ppm"), str(tmp_path / "l.ppm")
                native.write_ppm(a, img, binary=binary)
                native.write_ppm_levels(b, lev, binary=binary)
                assert open(a, "rb").read() == open(b, "rb").read()
    img = np.zeros((2, 2, 3), np.float32); img[1, 0, 2] = np.nan
    assert native.levels(img)[1] == 1                      # a NaN channel is reported: the caller takes the float writer
    with pytest.raises(native.RtiowError):
        native.write_ppm_levels(str(tmp_path / "no_such_dir" / "x.ppm"), np.zeros((1, 1, 3), np.uint8))


def test_shard_rows_partition_and_place_rows(native):
    for H, n, strip in [(1080, 8, 8), (1080, 3, 8), (192, 2, 8), (50, 4, 8), (7, 3, 2), (5, 8, 8), (1080, 1, 8)]:
        seen = []
        full = np.zeros((H, 4, 3), np.float32)
        want = np.arange(H * 4 * 3, dtype=np.float32).reshape(H, 4, 3)
        for r in range(n):
            rows = native.shard_rows(H, r, n, strip)
            assert np.all(np.diff(rows) > 0)
            assert all((row // strip) % n == r for row in rows)
            seen += list(rows)
            native.place_rows(full, np.ascontiguousarray(want[rows]), r, n, strip)
        assert sorted(seen) == list(range(H))
        assert np.array_equal(full, want)
    with pytest.raises(native.RtiowError):
        native.shard_rows(10, 3, 3, 8)


def _exe(native, name="global-float-hip-raytrace"):
    return os.path.join(os.path.dirname(native.lib_paths()["hip"]), "..", "bin", name)


def test_cli_help_and_missing_scene_id(native):
    # main.cu:62-73: --help -> usage on stdout, exit 0; no --scene_id -> message on stderr, usage on stdout, exit 1
    for name in ("global-float-hip-raytrace", "global-double-hip-raytrace"):
        r = subprocess.run([_exe(native, name), "--help"], capture_output=True, text=True)
        assert r.returncode == 0 and "--scene_id arg" in r.stdout and "(default: 320)" in r.stdout and "(default: 25)" in r.stdout
        r = subprocess.run([_exe(native, name), "-h"], capture_output=True, text=True)
        assert r.returncode == 0 and "Print usage" in r.stdout
        r = subprocess.run([_exe(native, name), "--width", "64"], capture_output=True, text=True)
        assert r.returncode == 1 and r.stderr == "Error: --scene_id is required.\n" and "Usage:" in r.stdout


def test_cli_unknown_option_aborts_like_uncaught_cxxopts(native):
    r = subprocess.run([_exe(native), "--scene_id", "1", "--bogus", "3"], capture_output=True, text=True)
    assert r.returncode == -6 and "does not exist" in r.stderr and r.stdout == ""
    r = subprocess.run([_exe(native), "--scene_id", "abc"], capture_output=True, text=True)
    assert r.returncode == -6 and r.stdout == ""


def test_committed_jump_constant_equals_the_matrix_power(native):
    """xorwow_jump67.inc (A^(2^67), used to skip 67 of the 98 squarings per process) against the same 32
    jump matrices derived from the one-step matrix A: host arithmetic of librtiow_hip.so, no GPU needed."""
    import ctypes
    lib = ctypes.CDLL(native.lib_paths()["hip_debug"])
    lib.rtiow_debug_jump_matrices.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t, ctypes.c_int]
    n = 32 * 160 * 5
    fast, scratch = (ctypes.c_uint32 * n)(), (ctypes.c_uint32 * n)()
    assert lib.rtiow_debug_jump_matrices(fast, n, 0) == 32
    assert lib.rtiow_debug_jump_matrices(scratch, n, 1) == 32
    assert bytes(fast) == bytes(scratch)
    assert len(set(bytes(fast)[k * 3200:(k + 1) * 3200] for k in range(32))) == 32     # 32 different matrices
    assert lib.rtiow_debug_jump_matrices(fast, n - 1, 0) < 0                            # buffer too small


def test_cli_extension_flags_reject_bad_values(native):
    # --schedule / --ppm_format / --scene_source are additions; a bad value fails like a bad integer does
    for flag, bad in (("--schedule", "bogus"), ("--ppm_format", "p9"), ("--scene_source", "texture")):
        r = subprocess.run([_exe(native), "--scene_id", "1", flag, bad], capture_output=True, text=True)
        assert r.returncode == -6 and "failed to parse" in r.stderr and r.stdout == ""


def test_cli_device_failure_keeps_stdout_empty(native):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    # main.cu:14-21: runtime failure -> message on stderr, exit(code), nothing on stdout (empty CSV cell)
    r = subprocess.run([_exe(native), "--scene_id=3", "--width=16", "--height=8"], capture_output=True, text=True)
    assert r.returncode != 0 and r.stdout == "" and "HIP_SAFE_CALL" in r.stderr


def test_bench_refuses_to_run_without_a_gpu():
    """No CPU fallback anywhere: bench.py says so and exits non-zero when no GPU is visible."""
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from tests.conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout) and not r.stdout.strip().startswith("{")


def test_renderer_group_documents_the_rccl_banner_on_stdout(native):
    """the library no longer redirects fd 1 around ncclCommInitAll, so a caller whose stdout is data has to know that RCCL prints its
    version banner there -- RendererGroup says so and offers quiet_stdout."""
    import inspect
    doc = native.RendererGroup.__doc__
    assert "RCCL" in doc and "stdout" in doc.lower() and "quiet_stdout" in doc
    assert "quiet_stdout" in inspect.signature(native.RendererGroup.__init__).parameters
