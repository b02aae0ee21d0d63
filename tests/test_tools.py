"""Harness tools of SURVEY §8(f): ppm_diff / scaled_ppm_diff (same CLI and P3 output as the
reference's src/ppm_diff tools, pinned against those tools built into oracle/_ref when present)
and csv_avg (process.py's pandas groupby-mean, pinned by the reference's own in->out CSV pairs)."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "csv")
REF = os.path.join(ROOT, "oracle", "_ref")


def _bin(native, name):
    return os.path.join(os.path.dirname(native.lib_paths()["hip"]), "..", "bin", name)


@pytest.mark.parametrize("src,want", [
    ("250427_gpu_global_float_timing_100sample.csv", "250427_avg_gpu_global_float_timing_100sample.csv"),
    ("250427_gpu_global_double_timing.csv", "250427_avg_gpu_global_double_timing.csv"),
    ("gpu_global_float_timing.csv", "avg_gpu_global_float_timing.csv"),      # has the empty --threads 32 rows
])
def test_csv_avg_reproduces_reference_pairs(native, tmp_path, src, want):
    out = str(tmp_path / "avg.csv")
    r = subprocess.run([_bin(native, "csv_avg"), os.path.join(GOLD, src), out], capture_output=True, text=True)
    assert r.returncode == 0 and "Averaged data saved to" in r.stdout
    assert open(out, "rb").read() == open(os.path.join(GOLD, want), "rb").read()


def test_csv_avg_matches_pandas_on_synthetic_rows(native, tmp_path):
    pd = pytest.importorskip("pandas")
    rng = np.random.default_rng(3)
    rows = ["scene_id,width,height,samples,bounces,threads,run,render_only_time_ms,end_to_end_time_ms"]
    for t in (16, 4, 8):
        for (w, h) in ((640, 384), (320, 192)):
            for run in range(1, 6):
                if t == 16 and w == 640:
                    rows.append("3,%d,%d,100,50,%d,%d," % (w, h, t, run))          # failed launch: empty cells
                else:
                    rows.append("3,%d,%d,100,50,%d,%d,%15.8f,%15.8f" % (w, h, t, run, rng.uniform(1, 500), rng.uniform(500, 900)))
    src = tmp_path / "in.csv"
    src.write_text("\n".join(rows) + "\n")
    out = tmp_path / "out.csv"
    assert subprocess.run([_bin(native, "csv_avg"), str(src), str(out)], capture_output=True).returncode == 0
    df = pd.read_csv(src)
    want = df.groupby(["scene_id", "width", "height", "samples", "bounces", "threads"])[["render_only_time_ms", "end_to_end_time_ms"]].mean().reset_index()
    want = want.rename(columns={"render_only_time_ms": "avg_render_only_time_ms", "end_to_end_time_ms": "avg_end_to_end_time_ms"})
    want_path = tmp_path / "want.csv"
    want.to_csv(want_path, index=False)
    assert out.read_bytes() == want_path.read_bytes()
    assert subprocess.run([_bin(native, "csv_avg"), str(tmp_path / "missing.csv"), str(out)], capture_output=True).returncode == 1


def _write_images(tmp_path, native):
    rng = np.random.default_rng(11)
    a = rng.random((7, 9, 3)).astype(np.float32)
    b = np.clip(a + rng.normal(0, 0.05, a.shape).astype(np.float32), 0, 1)
    pa, pb = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    native.write_ppm(pa, a)
    native.write_ppm(pb, b)
    # a P6 copy of b with header comments, to exercise the binary reader and comment skipping
    vb = np.array(open(pb, "rb").read().split()[4:], np.int64).astype(np.uint8)
    p6 = str(tmp_path / "b6.ppm")
    open(p6, "wb").write(b"P6\n# made by a test\n9 7\n# another comment\n255\n" + vb.tobytes())
    va = np.array(open(pa, "rb").read().split()[4:], np.int64)
    return pa, pb, p6, va, vb.astype(np.int64)


def test_ppm_diff_output_statistics_and_gate(native, tmp_path):
    pa, pb, p6, va, vb = _write_images(tmp_path, native)
    out = str(tmp_path / "d.ppm")
    r = subprocess.run([_bin(native, "ppm_diff"), pa, pb, out], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("Successfully wrote difference image to " + out)
    text = open(out).read()
    assert text.startswith("P3\n9 7\n255\n")
    body = text.split("\n")[3:-1]
    assert all(len(line.split()) == 12 for line in body[:-1]) and 1 <= len(body[-1].split()) <= 12
    got = np.array(text.split()[4:], np.int64)
    want = np.abs(va - vb)
    assert np.array_equal(got, want)
    assert "mean %.6f" % want.mean() in r.stdout and "max %d" % want.max() in r.stdout
    # P6 input gives the same difference image
    out6 = str(tmp_path / "d6.ppm")
    assert subprocess.run([_bin(native, "ppm_diff"), pa, p6, out6], capture_output=True).returncode == 0
    assert open(out6).read() == text
    # the tolerance gate: identical images pass any bound, different ones fail a zero bound
    assert subprocess.run([_bin(native, "ppm_diff"), pa, pa, out, "--max-abs", "0", "--max-mean", "0"], capture_output=True).returncode == 0
    assert subprocess.run([_bin(native, "ppm_diff"), pa, pb, out, "--max-abs", "0"], capture_output=True).returncode == 2
    # usage / missing file / size mismatch -> 1, like the reference
    assert subprocess.run([_bin(native, "ppm_diff"), pa, pb], capture_output=True).returncode == 1
    assert subprocess.run([_bin(native, "ppm_diff"), pa, str(tmp_path / "nope.ppm"), out], capture_output=True).returncode == 1
    small = str(tmp_path / "s.ppm")
    native.write_ppm(small, np.zeros((2, 2, 3), np.float32))
    assert subprocess.run([_bin(native, "ppm_diff"), pa, small, out], capture_output=True).returncode == 1


def test_ppm_tools_match_reference_tools_byte_for_byte(native, tmp_path):
    if not os.path.exists(os.path.join(REF, "ref_ppm_diff")):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    pa, pb, p6, _, _ = _write_images(tmp_path, native)
    for tool in ("ppm_diff", "scaled_ppm_diff"):
        for second in (pb, p6, pa):
            mine, ref = str(tmp_path / "mine.ppm"), str(tmp_path / "ref.ppm")
            r1 = subprocess.run([_bin(native, tool), pa, second, mine], capture_output=True, text=True)
            r2 = subprocess.run([os.path.join(REF, "ref_" + tool), pa, second, ref], capture_output=True, text=True)
            assert r1.returncode == r2.returncode == 0
            assert open(mine, "rb").read() == open(ref, "rb").read(), (tool, second)
            # the reference's stdout lines are a prefix set of ours (ppm_diff adds a statistics line)
            assert r2.stdout.replace(ref, mine).strip().splitlines()[-1] in r1.stdout


def test_benchmark_script_writes_reference_csv_schema(native):
    text = open(os.path.join(ROOT, "tools", "hip_benchmark.sh")).read()
    assert "scene_id,width,height,samples,bounces,threads,run,render_only_time_ms,end_to_end_time_ms" in text
    assert subprocess.run(["bash", "-n", os.path.join(ROOT, "tools", "hip_benchmark.sh")]).returncode == 0
