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
    assert "practical_peak" not in rf            # a "fraction" of a self-declared ceiling that read 1.046 is gone
    assert rf["issued"]["cycles_per_inst_charged"] == 2.0 and rf["issue_saturation"] is None      # no SQ_ACTIVE_INST_VALU in this record
    # the saturation evidence comes from the counters alone: 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU SIMD cycles per instruction, and that busy time
    # over all SIMD cycles of the profiled launch
    pmc_s = {"build_id": "x", "counters": {"main": {"SQ_INSTS_VALU": 6.144e9, "SQ_ACTIVE_INST_VALU": 6.0e9, "GRBM_GUI_ACTIVE": 8 * 2.4e7}}}
    sat = bench.roofline_object(_args(), st, 468_000_000, main_ms, pmc_s, "test", 1)["issue_saturation"]
    assert sat["simd_cycles_busy_per_valu_inst"] == pytest.approx(4 * 6.0e9 / 6.144e9, abs=1e-3) and sat["simd_valu_busy_share"] == pytest.approx(4 * 6.0e9 / (2.4e7 * 1024), abs=1e-4)
    # fp64 WITHOUT executed double-precision counts in the record: the static ISA share stands in, and the line says so
    rf64 = bench.roofline_object(_args(precision=64), st, 468_000_000, main_ms, pmc, "test", 1)
    cpi = 2.0 + 2.0 * bench.FP64_KERNEL_DP_SHARE
    assert rf64["frac"] == pytest.approx(0.5 * cpi / 2.0, abs=1e-4) and rf64["peak"] == 78.6 and rf64["dp_cycles_from"] == "static"
    assert rf64["achieved"] == pytest.approx(rf64["frac"] * 78.6, rel=1e-3) and rf64["issued"]["cycles_per_inst_charged"] == pytest.approx(cpi, abs=1e-3)
    assert rf64["issued"]["valu_issue_frac"] == pytest.approx(0.5, abs=1e-4)          # the 2-cycle figure stays beside it
    assert "static ISA share" in rf64["achieved_is"] and "charged 4" in rf64["achieved_is"]
    # fp64 WITH the "f64" pass: what executed is charged -- add / mul / fma 4 cycles, transcendentals 8, the rest 2
    main64 = dict(pmc["counters"]["main"], SQ_INSTS_VALU_ADD_F64=0.5e9, SQ_INSTS_VALU_MUL_F64=0.5e9, SQ_INSTS_VALU_FMA_F64=2.0e9, SQ_INSTS_VALU_TRANS_F64=0.072e9)
    pmc64 = dict(pmc, counters={"main": main64})
    rf64 = bench.roofline_object(_args(precision=64), st, 468_000_000, main_ms, pmc64, "test", 1)
    cycles = 2.0 * (6.144e9 - 3.0e9 - 0.072e9) + 4.0 * 3.0e9 + 8.0 * 0.072e9
    assert rf64["dp_cycles_from"] == "executed" and rf64["issued"]["cycles_per_inst_charged"] == pytest.approx(cycles / 6.144e9, abs=1e-3)
    assert rf64["frac"] == pytest.approx(cycles / (1024 * 2.4e9 * 10e-3), abs=1e-4) and rf64["issued"]["dp_share_executed"] == pytest.approx(3.072e9 / 6.144e9, abs=1e-4)
    assert "SQ_INSTS_VALU_*_F64" in rf64["achieved_is"] and "static" not in rf64["achieved_is"]
    assert bench.issue_fraction(6.144e9, 10.0, 32) == (pytest.approx(0.5), 2.0)
    assert bench.issue_fraction(6.144e9, 10.0, 64, dp=(3.0e9, 0.072e9))[1] == pytest.approx(cycles / 6.144e9)
    # more than one rank and no per-rank records: no counter figure at all
    assert bench.roofline_object(_args(), st, 468_000_000, main_ms, pmc, "test", 2)["frac"] is None


def test_n_gt_1_line_rates_every_rank_against_its_shard_record(tmp_path, monkeypatch):
    """VERDICT r04 missing #3: the N > 1 contract line carries roofline.frac -- each rank's measured main-launch time against the committed
    SQ_INSTS_VALU of its shard (taken on one GPU, keyed by shard and build id); the headline is the slowest rank's; a record of another
    build gives null for that rank, never an old number."""
    import bench
    import pmc_passes
    args = _args(strip_rows=8, pmc="auto")
    keys = [pmc_passes.config_key(3, 1920, 1080, 100, 50, 32, shard=(k, 2, 8)) for k in range(2)]
    assert keys[0].endswith("_r0of2x8") and keys[1].endswith("_r1of2x8") and keys[0] != pmc_passes.config_key(3, 1920, 1080, 100, 50, 32)
    assert pmc_passes.config_key(3, 1920, 1080, 100, 50, 32, shard="0,1,8") == pmc_passes.config_key(3, 1920, 1080, 100, 50, 32)     # one rank = the whole frame
    assert "--shard" in pmc_passes.one_render_args({"scene_id": 3, "width": 1920, "height": 1080, "samples": 100, "bounces": 50, "precision": 32, "shard": "1,2,8"}, 2)
    path = tmp_path / "pmc_records.json"
    path.write_text(json.dumps({keys[0]: {"build_id": "a" * 64, "key": keys[0], "counters": {"main": {"SQ_INSTS_VALU": 3.0e9, "SQ_THREAD_CYCLES_VALU": 16.0 * 3e9, "SQ_ACTIVE_INST_VALU": 3e9}}},
                                keys[1]: {"build_id": "a" * 64, "key": keys[1], "counters": {"main": {"SQ_INSTS_VALU": 3.6e9}}}}))
    monkeypatch.setattr(bench, "PMC_RECORDS", str(path))
    recs, note = bench.shard_records(args, 2, "a" * 64)
    assert all(r is not None for r in recs) and "2 of 2 ranks" in note
    st = {"num_spheres": 125, "primary_rays": 1920 * 1080 * 100 // 2, "prepass_samples": 3, "phases": 2, "solo_waves": 512}
    ranks = [{"pmc": recs[0], "main_ms": 6.0}, {"pmc": recs[1], "main_ms": 8.0}]
    rf = bench.roofline_object(args, st, 234_000_000, [6.0, 6.0], None, note, 2, ranks)
    f0, f1 = 2 * 3.0e9 / (1024 * 2.4e9 * 6e-3), 2 * 3.6e9 / (1024 * 2.4e9 * 8e-3)
    assert [e["frac"] for e in rf["frac_per_rank"]] == [pytest.approx(f0, abs=1e-4), pytest.approx(f1, abs=1e-4)]
    assert rf["frac_is_rank"] == 1 and rf["frac"] == pytest.approx(f1, abs=1e-4) and rf["frac_max_over_ranks"] == pytest.approx(max(f0, f1), abs=1e-4)
    assert rf["frac_per_rank"][0]["active_lane_frac"] == pytest.approx(0.25) and rf["kernel"].startswith("render_solo_kernel")
    # another build is loaded: null everywhere, and the note says why
    recs, note = bench.shard_records(args, 2, "b" * 64)
    assert recs == [None, None] and "no per-shard record" in note
    rf = bench.roofline_object(args, st, 234_000_000, [6.0, 6.0], None, note, 2, [{"pmc": None, "main_ms": 6.0}, {"pmc": None, "main_ms": 8.0}])
    assert rf["frac"] is None and [e["frac"] for e in rf["frac_per_rank"]] == [None, None] and rf["frac_is_rank"] == 1
    assert bench.shard_records(_args(strip_rows=8, pmc="off"), 2, "a" * 64) == ([None, None], "--pmc off")


def test_headline_frac_is_not_null_when_a_record_of_the_loaded_build_is_committed(native, monkeypatch):
    """the GPU test of the contract line accepts frac == null (live passes may be refused on a box).  This pins the committed-record
    leg on the CPU: whenever profiles/pmc_records.json holds the headline's record for the build in the tree, `--pmc committed` yields a number."""
    import bench
    import pmc_passes
    rec, note = bench.pmc_committed(_args(), native.build_id())
    if rec is None:
        pytest.skip("no committed headline record of the current build (%s)" % note)
    st = {"num_spheres": 125, "primary_rays": 1920 * 1080 * 100, "prepass_samples": 3, "phases": 2, "solo_waves": 0}
    rf = bench.roofline_object(_args(), st, 468_000_000, [10.3], rec, note, 1)
    assert rf["frac"] is not None and 0.2 < rf["frac"] < 1.0 and rf["build_id"] == native.build_id() and rf["traffic"]


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
