"""CPU sanitizer run (the reference carries latent UB on this path: an uninitialised
sphere slot, hittable.h:34, and int(256*x) of a NaN in its writer, main.cu:374): the oracle, the
host library with both PPM writers, and the harness tools, built with AddressSanitizer + UBSan
(`make -C oracle asan`) and run on small inputs and on the reference's CSV fixtures.  Any finding
aborts the process (-fno-sanitize-recover), which fails the test."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT

ASAN = os.path.join(ROOT, "oracle", "_asan")
GOLD = os.path.join(ROOT, "tests", "golden", "csv")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=66", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.fixture(scope="module")
def asan_bin():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return ASAN


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, env=ENV, **kw)
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-3000:]
    return r


def test_oracle_host_library_and_writers_under_asan_ubsan(asan_bin, tmp_path):
    r = _run([os.path.join(asan_bin, "sanitize_main"), str(tmp_path)])
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    assert "0 self-check failure(s)" in r.stdout
    # the NaN channel the driver plants is written the way the reference's x86 build prints it
    text = open(os.path.join(str(tmp_path), "global_float_scene3_24x14_3samples_12bounces_8threadsPerBlockRow.ppm")).read().split("\n")
    assert text[:3] == ["P3", "24 14", "255"] and text[3] == "0 255 -2147483648"


def test_tools_under_asan_ubsan(asan_bin, native, tmp_path):
    rng = np.random.default_rng(5)
    a = rng.random((9, 13, 3)).astype(np.float32)
    b = np.clip(a + rng.normal(0, 0.05, a.shape).astype(np.float32), 0, 1)
    pa, pb = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    native.write_ppm(pa, a); native.write_ppm(pb, b)
    for tool in ("ppm_diff", "scaled_ppm_diff"):
        out = str(tmp_path / (tool + ".ppm"))
        r = _run([os.path.join(asan_bin, tool), pa, pb, out])
        assert r.returncode == 0 and os.path.getsize(out) > 0, r.stderr[-500:]
        # error paths: a missing file, a truncated file, mismatched sizes
        assert _run([os.path.join(asan_bin, tool), pa, str(tmp_path / "missing.ppm"), out]).returncode != 0
        trunc = tmp_path / "trunc.ppm"
        trunc.write_text(open(pa).read()[:60])
        assert _run([os.path.join(asan_bin, tool), pa, str(trunc), out]).returncode != 0
        small = str(tmp_path / "small.ppm")
        native.write_ppm(small, a[:4])
        assert _run([os.path.join(asan_bin, tool), pa, small, out]).returncode != 0
    for src in ("250427_gpu_global_float_timing_100sample.csv", "gpu_global_float_timing.csv"):
        out = str(tmp_path / "avg.csv")
        r = _run([os.path.join(asan_bin, "csv_avg"), os.path.join(GOLD, src), out])
        assert r.returncode == 0
        want = "250427_avg_gpu_global_float_timing_100sample.csv" if src.startswith("2504") else "avg_gpu_global_float_timing.csv"
        assert open(out, "rb").read() == open(os.path.join(GOLD, want), "rb").read()
    assert _run([os.path.join(asan_bin, "csv_avg"), str(tmp_path / "nope.csv"), str(tmp_path / "o.csv")]).returncode != 0
    bad = tmp_path / "bad.csv"
    bad.write_text("scene_id,width\n1,2,3,4,5,6,7,abc,\n,,,,\n")
    _run([os.path.join(asan_bin, "csv_avg"), str(bad), str(tmp_path / "o.csv")])       # any exit code, but no sanitizer finding
