"""Register budget of the render kernels, read from the compiler's own metadata (no GPU needed: hipcc
cross-compiles gfx950 here).  DESIGN.md §4.5 quotes these figures; this test is what keeps them from drifting.

Why it matters: the path loop is bound by vector issue.  An SGPR the allocator cannot keep is spilled to a lane of
a VGPR (v_writelane / v_readlane pairs inside the loop are vector instructions), scratch would add memory traffic to
a loop that has none, and the VGPR count sets the waves per SIMD (MI355X_MICROARCH.md, Register files: <= 96
allocated registers -> 5 waves, <= 128 -> 4).  The launch parameters a kernel needs rarely -- camera, cold
bookkeeping, grid description, screening table -- are therefore re-read from the kernarg segment where they are
used (device/params.h: cam_of, cold_of, grid_of, screen_of) instead of living in SGPRs for the whole loop.
"""
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT


@pytest.fixture(scope="module")
def kernel_metadata(tmp_path_factory, native):
    from raytracingincuda_amd import build as b
    out = str(tmp_path_factory.mktemp("isa") / "rtiow_hip.s")
    flags = [f for f in b.HIP_FLAGS if f not in ("-shared",)]
    subprocess.run([b._hipcc()] + flags + ["-S", "--cuda-device-only", "-o", out, os.path.join(b.CSRC, "rtiow_hip.hip")],
                   check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    meta = {}
    pat = re.compile(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n"
                     r"(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)")
    found = list(pat.finditer(text))
    names = subprocess.run(["c++filt"], input="\n".join(m.group(1) for m in found), capture_output=True, text=True, check=True).stdout.splitlines()
    for m, name in zip(found, names):
        meta[name] = {"scratch": int(m.group(2)), "sgpr": int(m.group(3)), "sgpr_spill": int(m.group(4)), "vgpr": int(m.group(5)), "vgpr_spill": int(m.group(6))}
    return meta, text


def _find(meta, *parts):
    hits = [k for k in meta if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, hits)
    return meta[hits[0]]


def test_every_render_kernel_is_free_of_scratch_and_vgpr_spills(kernel_metadata):
    meta, _ = kernel_metadata
    render = {k: v for k, v in meta.items() if "render_" in k}
    assert len(render) >= 36                      # {static, persistent, prepass} x {LDS, scalar} x {count, plain} x {f32, f64} + solo
    for k, v in render.items():
        assert v["scratch"] == 0 and v["vgpr_spill"] == 0, (k, v)


def test_main_launch_register_budget(kernel_metadata):
    meta, _ = kernel_metadata
    # the kernels bench.py times: LDS scene source (template argument 0), no counting
    f32 = _find(meta, "render_persistent_kernel<float, 0, false, false>")
    assert f32["sgpr_spill"] <= 6 and f32["vgpr"] <= 96, f32        # five waves per SIMD
    # the same kernel with the bounded rejection loop (full frames: launch_render takes it at >= 4 pools per resident wave)
    f32b = _find(meta, "render_persistent_kernel<float, 0, false, true>")
    assert f32b["sgpr_spill"] <= 6 and f32b["vgpr"] <= 96, f32b
    pre = _find(meta, "render_prepass_kernel<float, 0, false, false>")
    assert pre["sgpr_spill"] <= 6 and pre["vgpr"] <= 96, pre
    solo = _find(meta, "render_solo_kernel<float, 0>")
    assert solo["sgpr_spill"] <= 6 and solo["vgpr"] <= 96, solo
    static = _find(meta, "render_kernel<float, 0, false>")
    assert static["sgpr_spill"] == 0 and static["vgpr"] <= 96, static
    f64 = _find(meta, "render_persistent_kernel<double, 0, false, false>")
    assert f64["sgpr_spill"] <= 16 and f64["vgpr"] <= 128, f64      # four waves per SIMD
    assert _find(meta, "render_solo_kernel<double, 0>")["vgpr"] <= 128


def test_design_md_quotes_the_committed_kernel(kernel_metadata):
    """DESIGN.md states the register figures of the fp32 main launch; they must be the compiler's."""
    meta, _ = kernel_metadata
    f32 = _find(meta, "render_persistent_kernel<float, 0, false, false>")
    f64 = _find(meta, "render_persistent_kernel<double, 0, false, false>")
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"render_persistent_kernel<float>`?: (\d+) VGPRs, (\d+) SGPR spills.*?render_persistent_kernel<double>`?: (\d+) VGPRs, (\d+) SGPR spills", text, re.S)
    assert m, "DESIGN.md §4.5 must carry the line `render_persistent_kernel<float>: N VGPRs, M SGPR spills ... render_persistent_kernel<double>: ...`"
    assert (int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4))) == (f32["vgpr"], f32["sgpr_spill"], f64["vgpr"], f64["sgpr_spill"])


def test_fp64_issue_fraction_uses_the_kernels_own_double_precision_share(kernel_metadata):
    """bench.py charges the fp64 main kernel's double-precision instructions 4 issue cycles and the others 2 (roofline.frac);
    FP64_KERNEL_DP_SHARE must be the share in the compiler's output for that kernel."""
    import bench
    _, text = kernel_metadata
    sym = [l for l in text.splitlines() if l.startswith("_ZN") and "render_persistent_kernelIdLi0ELb0ELb0E" in l and l.rstrip().split()[0].endswith(":")]
    assert len(sym) == 1, sym
    start = text.index("\n" + sym[0].split()[0])
    body = text[start:text.index(".Lfunc_end", start)]
    valu = [l.split()[0] for l in body.splitlines() if l.startswith("\tv_")]
    dp = [op for op in valu if re.search(r"_f64|_[bui]64", op)]
    share = len(dp) / len(valu)
    assert abs(share - bench.FP64_KERNEL_DP_SHARE) <= 0.04, (share, len(valu))
