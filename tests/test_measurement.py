"""bench.py's roofline figure cannot go stale (CPU tests of the measurement plumbing).

`roofline.frac` is EXECUTED work -- the vector-issue fraction of the main launch, from rocprofv3 --pmc passes -- and
every counter record carries the build id of the library it was measured on (SHA-256 of the HIP sources + flags,
compiled into the library: rtiow_build_id()).  bench.py uses a record only when that id is the loaded library's,
and prints null otherwise.  The passes themselves need a GPU (tests/test_gpu_parity.py runs bench.py with its
live passes); everything around them is checked here.
"""
import argparse
import json
import os
import sys

import pytest

from tests.conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_build_id_is_the_hash_of_sources_and_flags(native):
    from raytracingincuda_amd import build as b
    assert native.build_id() == b.hip_build_id()            # the library that is loaded was built from the sources in the tree
    assert len(native.build_id()) == 64 and b.hip_build_id(["-DANYTHING"]) != b.hip_build_id()


def test_counter_tree_is_reduced_per_launch_class(tmp_path):
    import pmc_passes
    d = tmp_path / "host" / "1234"
    d.mkdir(parents=True)
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"]
    def row(disp, kernel, counter, value):
        rows.append('%d,%d,1,1,1,1,1,1,"%s",256,0,0,88,0,106,%s,%s,0,1' % (disp, disp, kernel, counter, value))
    main = "void (anonymous namespace)::render_persistent_kernel<float, 0, false, true>((anonymous namespace)::RenderParams<float>)"
    pre = "void (anonymous namespace)::render_prepass_kernel<float, 0, false, true>((anonymous namespace)::RenderParams<float>)"
    for disp, v in ((1, 100.0), (5, 300.0)):                 # two dispatches of the main launch; one of them reported in two rows (dimension instances)
        row(disp, main, "SQ_INSTS_VALU", v)
    row(5, main, "SQ_INSTS_VALU", 50.0)
    row(2, pre, "SQ_INSTS_VALU", 7.0)
    row(3, "void (anonymous namespace)::cost_scan_kernel(unsigned int const*, unsigned int*, unsigned int*)", "SQ_INSTS_VALU", 1.0)
    row(4, "void (anonymous namespace)::place_pixels_kernel<float>(float const*, int const*, float*, int)", "WRITE_SIZE", 24300.0)
    row(9, "some_other_kernel", "SQ_INSTS_VALU", 1e9)
    (d / "pmc_counter_collection.csv").write_text("\n".join(rows) + "\n")
    means, counts = pmc_passes.parse_counter_tree(str(tmp_path))
    assert means["main"]["SQ_INSTS_VALU"] == (100.0 + 350.0) / 2 and counts["main"]["SQ_INSTS_VALU"] == 2
    assert means["prepass"]["SQ_INSTS_VALU"] == 7.0 and means["sort"]["SQ_INSTS_VALU"] == 1.0 and means["place"]["WRITE_SIZE"] == 24300.0
    assert set(means) == {"main", "prepass", "sort", "place"}


def test_issue_fraction_arithmetic():
    import pmc_passes
    # 1024 SIMDs x 2.4e9 cycles/s x 10 ms = 2.4576e10 SIMD-cycles; a wave64 instruction takes 2 of them
    d = pmc_passes.derive({"SQ_INSTS_VALU": 6.144e9, "SQ_THREAD_CYCLES_VALU": 32.0 * 5e9, "SQ_ACTIVE_INST_VALU": 5e9, "GRBM_GUI_ACTIVE": 8 * 2.0e7}, launch_ms=10.0)
    assert d["valu_issue_frac"] == pytest.approx(0.5) and d["active_lane_frac"] == pytest.approx(0.5)
    assert d["simd_cycles_per_valu_inst"] == pytest.approx(2.0e7 * 1024 / 6.144e9) and d["valu_issue_frac_at_profiled_clock"] == pytest.approx(2 * 6.144e9 / (2.0e7 * 1024))
    assert "valu_issue_frac" not in pmc_passes.derive({"SQ_INSTS_VALU": 1.0})       # no launch time, no rate


def _args(**kw):
    base = dict(scene_id=3, width=1920, height=1080, samples=100, bounces=50, precision=32, schedule="sorted", scene_source="grid", threads=0)
    base.update(kw)
    return argparse.Namespace(**base)


def test_a_record_of_another_build_is_never_used(tmp_path, monkeypatch):
    import pmc_passes
    import bench
    key = pmc_passes.config_key(3, 1920, 1080, 100, 50, 32)
    path = tmp_path / "pmc_records.json"
    rec = {"build_id": "a" * 64, "key": key, "counters": {"main": {"SQ_INSTS_VALU": 7.0e9}}, "derived_main": {}}
    path.write_text(json.dumps({key: rec}))
    assert pmc_passes.load_record(str(path), key, "a" * 64)["build_id"] == "a" * 64
    assert pmc_passes.load_record(str(path), key, "b" * 64) is None
    assert pmc_passes.load_record(str(path), pmc_passes.config_key(1, 1920, 1080, 100, 50, 32), "a" * 64) is None
    assert pmc_passes.load_record(str(tmp_path / "absent.json"), key, "a" * 64) is None
    monkeypatch.setattr(bench, "PMC_RECORDS", str(path))
    got, note = bench.pmc_committed(_args(), "b" * 64)
    assert got is None and "no record" in note
    got, note = bench.pmc_committed(_args(), "a" * 64)
    assert got["build_id"] == "a" * 64 and "matches" in note


def test_roofline_frac_is_executed_work_or_null():
    import bench
    st = {"num_spheres": 125, "primary_rays": 1920 * 1080 * 100, "prepass_samples": 3, "phases": 2, "solo_waves": 0}
    main_ms = [10.0, 10.0]
    # no counters: frac is null, the comparable-work figure stays
    rf = bench.roofline_object(_args(), st, 468_000_000, main_ms, None, "--pmc off", 1)
    assert rf["frac"] is None and rf["achieved"] is None and rf["traffic"] is None and rf["issued"] is None
    assert rf["algorithmic_frac"] == pytest.approx(rf["algorithmic_TFLOPs"] / 157.3, rel=1e-3) and rf["bound"] == "valu" and rf["unit"] == "TFLOP/s"
    # counters of this build: frac = 2 SIMD-cycles per wave-instruction / the launch's SIMD-cycles at 2.4 GHz
    pmc = {"build_id": "x", "counters": {"main": {"SQ_INSTS_VALU": 6.144e9, "SQ_INSTS_SALU": 2.0e9}, "prepass": {"SQ_INSTS_VALU": 3.0e8}},
           "traffic_main": {"hbm_bytes": 2.0e8, "fetch_bytes_raw": 8.0e7, "write_bytes": 2.7e7}}
    rf = bench.roofline_object(_args(), st, 468_000_000, main_ms, pmc, "test", 1)
    assert rf["frac"] == pytest.approx(0.5, abs=1e-4) and rf["achieved"] == pytest.approx(0.5 * 157.3, rel=1e-3) and rf["peak"] == 157.3
    assert rf["issued"]["valu_issue_frac"] == pytest.approx(rf["frac"], abs=1e-4) and rf["traffic"] == 2.0e8 and rf["write_bytes"] == 2.7e7
    assert rf["practical_peak"]["value"] == bench.PRACTICAL_FMA_TFLOPS[32] and rf["practical_peak"]["frac_of_practical"] == pytest.approx(0.5 * 157.3 / bench.PRACTICAL_FMA_TFLOPS[32], rel=1e-3)
    assert rf["issued"]["cycles_per_inst_charged"] == 2.0
    # fp64: the kernel's double-precision instructions (FP64_KERNEL_DP_SHARE of them) are charged 4 cycles, the others 2 (the
    # text said "fp64: 4" while every instruction was charged 2); achieved = frac x the fp64 peak, and the text says what is computed
    rf64 = bench.roofline_object(_args(precision=64), st, 468_000_000, main_ms, pmc, "test", 1)
    cpi = 2.0 + 2.0 * bench.FP64_KERNEL_DP_SHARE
    assert rf64["frac"] == pytest.approx(0.5 * cpi / 2.0, abs=1e-4) and rf64["peak"] == 78.6
    assert rf64["achieved"] == pytest.approx(rf64["frac"] * 78.6, rel=1e-3) and rf64["issued"]["cycles_per_inst_charged"] == pytest.approx(cpi, abs=1e-3)
    assert rf64["issued"]["valu_issue_frac"] == pytest.approx(0.5, abs=1e-4)          # the 2-cycle figure stays beside it
    assert "%.2f cycles per instruction" % cpi in rf64["achieved_is"] and "charged 4" in rf64["achieved_is"]
    assert bench.issue_fraction(6.144e9, 10.0, 32) == (pytest.approx(0.5), 2.0)
    # more than one rank: no counter figure at all
    assert bench.roofline_object(_args(), st, 468_000_000, main_ms, pmc, "test", 2)["frac"] is None


def test_failing_live_passes_do_not_take_the_bench_line_down(monkeypatch):
    """No rocprofv3, no permission to profile, a pass that dies: bench.py notes why and goes on (frac null or the committed record)."""
    import bench
    import pmc_passes
    def boom(*a, **k):
        raise RuntimeError("rocprofv3 not found")
    monkeypatch.setattr(pmc_passes, "collect", boom)
    rec, note = bench.pmc_live(_args())
    assert rec is None and "live passes failed" in note and "rocprofv3 not found" in note
    monkeypatch.setenv("ROCPROFILER_REGISTER_FORCE_LOAD", "1")
    assert bench.under_a_profiler()                     # bench.py below rocprofv3 itself: no nested passes


def test_committed_records_name_their_build(native):
    """profiles/pmc_records.json: every record carries a build id; those of the current tree (if any) make
    `bench.py --pmc committed` print a figure, the others make it print null -- which this test reports, not fails."""
    path = os.path.join(ROOT, "profiles", "pmc_records.json")
    if not os.path.exists(path):
        pytest.skip("no committed counter records")
    recs = json.load(open(path))
    assert recs and all(len(r.get("build_id", "")) >= 8 and r["counters"].get("main", {}).get("SQ_INSTS_VALU", 0) > 0 for r in recs.values())
    current = [k for k, r in recs.items() if r["build_id"] == native.build_id()]
    print("records of the current build:", current or "none (bench.py --pmc committed prints null; the default live passes do not need them)")


def test_scaling_table_reads_driver_records(tmp_path):
    """scripts/scaling_table.py: SCALE_rNN.json (contract lines nested anywhere, or raw lines) -> ms, speed-up, efficiency, against floor_ms."""
    import json, subprocess, sys
    from tests.conftest import ROOT
    lines = [{"n_gpus": n, "ms_per_step": ms, "value": 1920 * 1080 * 100 / ms / 1e3,
              "scaling_detail": {"floor_ms": 4.0, "kernel_ms_per_rank": [ms - 0.2] * n, "gather_ms": 0.1, "gather_transport": "rccl"}}
             for n, ms in ((1, 12.0), (2, 8.0), (4, 6.0), (8, 5.0))]
    nested, raw = tmp_path / "scale.json", tmp_path / "lines.txt"
    nested.write_text(json.dumps({"runs": [{"n": l["n_gpus"], "parsed": l, "stdout": "noise"} for l in lines]}))
    raw.write_text("\n".join(json.dumps(l) for l in lines))
    for f in (nested, raw):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "scaling_table.py"), str(f)], capture_output=True, text=True, check=True).stdout
        rows = [json.loads(l) for l in out.splitlines()]
        assert [r["n_gpus"] for r in rows] == [1, 2, 4, 8]
        assert [r["speedup"] for r in rows] == [1.0, 1.5, 2.0, 2.4] and [r["efficiency"] for r in rows] == [1.0, 0.75, 0.5, 0.3]
        assert rows[3]["ms_over_floor"] == 1.25 and rows[0]["speedup_limit_by_floor"] == 3.0
