#!/usr/bin/env python3
"""rocprofv3 --pmc passes over ONE render configuration, reduced to a small record.

The render path is bound by vector-ALU issue, so the roofline figure of a bench line is the
fraction of the SIMDs' issue slots the main launch fills -- SQ_INSTS_VALU against the launch time -- and HBM
traffic is evidence that the path is nowhere near the memory roof.  Both come from hardware counters:

  pass "sq"     SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES
                SQ_WAIT_INST_ANY SQ_WAIT_ANY (the 8 SQ slots of gfx950) + GRBM_GUI_ACTIVE (its own block)
  pass "fetch"  FETCH_SIZE      } separate passes, as MI355X_MICROARCH.md (HBM, rocprofv3 PMC slots) prescribes:
  pass "write"  WRITE_SIZE      } the two do not fit the TCC's 4 slots together
  pass "f64"    SQ_INSTS_VALU + SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 (fp64 kernels): the double-precision instructions that EXECUTED,
                which bench.py charges 4 issue cycles (transcendentals 8) instead of 2

A shard of a multi-GPU frame is a configuration of its own: cfg["shard"] = "rank,nranks,strip_rows" renders only that rank's row strips
(one_render.py --shard) and the record's key ends in _r<rank>of<nranks>x<strip_rows>.  SQ_INSTS_VALU of a shard does not depend on which
device renders it, so `bench.py --gpus N` rates every rank's main launch against records taken on ONE GPU (scripts/pmc_shard_records.py).

Each pass profiles `python3 scripts/one_render.py <config>` (the program itself after `--`, nothing that re-execs).
The record carries `build_id` = rtiow_build_id() of the library that rendered (SHA-256 of its sources and flags):
bench.py uses a committed record only when that id equals the id of the library it has loaded, and prints null
otherwise -- a counter figure can never outlive the kernel it was measured on.

    python3 scripts/pmc_passes.py --out profiles/r03/r03_pmc_records.json [--scene_id 3 --width 1920 ...] [--passes sq,fetch,write]

bench.py imports collect() for its live leg (a child process per pass, started before bench.py touches the GPU).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PASSES = {
    "sq": ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES",
           "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"],
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "f64": ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"],
}
F64_COUNTERS = ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")
SCHEDULES = {"static": 0, "persistent": 1, "sorted": 2}
SOURCES = {"lds": 0, "scalar": 1, "lds_exact": 2, "grid": 3}
N_SIMD = 256 * 4                      # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32
NOMINAL_GHZ = 2.4


def config_key(scene_id, width, height, samples, bounces, precision, schedule="sorted", scene_source="grid", shard=None):
    """`shard` = (rank, nranks, strip_rows) or "rank,nranks,strip_rows" for one rank's row strips of the frame."""
    key = "s%d_%dx%d_%dspp_%db_f%d" % (scene_id, width, height, samples, bounces, precision)
    if (schedule, scene_source) != ("sorted", "grid"):
        key += "_%s_%s" % (schedule, scene_source)
    if shard:
        rank, nranks, strip = (int(x) for x in (shard.split(",") if isinstance(shard, str) else shard))
        if nranks > 1:
            key += "_r%dof%dx%d" % (rank, nranks, strip)
    return key


def kernel_class(name):
    """Which launch of a render step a kernel name belongs to (device/render_kernels.h, device/cost_sort.h)."""
    if "render_prepass_kernel" in name:
        return "prepass"
    if "render_persistent_kernel" in name or "render_solo_kernel" in name or "render_kernel" in name:
        return "main"
    if "cost_" in name:
        return "sort"
    if "place_pixels_kernel" in name:
        return "place"
    return None


def parse_counter_tree(directory):
    """{class: {counter: mean per dispatch}} over every *_counter_collection.csv below `directory`.  A dispatch's
    value of a counter is the SUM of its rows (rocprofv3 may print one row per dimension instance)."""
    per_dispatch = {}
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            cls = kernel_class(r["Kernel_Name"])
            if cls is None:
                continue
            key = (cls, r["Counter_Name"], os.path.basename(f), r.get("Dispatch_Id", r.get("Correlation_Id", "")))
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(r["Counter_Value"])
    out, counts = {}, {}
    for (cls, counter, _, _), v in per_dispatch.items():
        out.setdefault(cls, {}).setdefault(counter, []).append(v)
    for cls in out:
        for counter, vals in out[cls].items():
            counts.setdefault(cls, {})[counter] = len(vals)
            out[cls][counter] = sum(vals) / len(vals)
    return out, counts


def one_render_args(cfg, reps):
    return ["--scene", str(cfg["scene_id"]), "--w", str(cfg["width"]), "--h", str(cfg["height"]), "--s", str(cfg["samples"]),
            "--b", str(cfg["bounces"]), "--prec", str(cfg["precision"]), "--sched", str(SCHEDULES[cfg.get("schedule", "sorted")]),
            "--source", str(SOURCES[cfg.get("scene_source", "grid")]), "--threads", str(cfg.get("threads", 0)), "--reps", str(reps), "--build-id"] + (
                ["--shard", str(cfg["shard"])] if cfg.get("shard") else [])


def run_pass(cfg, counters, reps, timeout, keep_dir=None, log=None):
    """One rocprofv3 --pmc child over scripts/one_render.py.  Returns (per-class means, dispatch counts, build_id, ms list)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        raise RuntimeError("rocprofv3 not found")
    work = keep_dir or tempfile.mkdtemp(prefix="rtiow_pmc_", dir="/tmp")
    os.makedirs(work, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [rocprof, "--pmc"] + counters + ["-d", work, "-o", "pmc", "--output-format", "csv", "--",
                                           sys.executable, os.path.join(ROOT, "scripts", "one_render.py")] + one_render_args(cfg, reps)
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    if log is not None:
        log.append({"cmd": " ".join(cmd[:len(counters) + 2]) + " ... one_render.py", "rc": r.returncode, "seconds": round(time.perf_counter() - t0, 2)})
    if r.returncode != 0:
        raise RuntimeError("rocprofv3 pass failed (rc %d): %s" % (r.returncode, (r.stderr or r.stdout)[-400:]))
    build_id, ms = None, None
    for line in r.stdout.splitlines():
        if line.startswith("build_id "):
            build_id = line.split()[1]
        if line.startswith("render_ms "):
            ms = [float(x) for x in line.split()[1:]]
    means, counts = parse_counter_tree(work)
    if keep_dir is None:
        shutil.rmtree(work, ignore_errors=True)
    return means, counts, build_id, ms


def derive(main, launch_ms=None):
    """Issue figures of the main launch from its counters (per launch).  `launch_ms` = the launch time to rate them
    against (the un-profiled HIP-event time when the caller has one; profiled passes run at a lower clock)."""
    d = {}
    insts = main.get("SQ_INSTS_VALU")
    if insts:
        d["valu_wave_insts_per_launch"] = insts
        if main.get("SQ_THREAD_CYCLES_VALU") and main.get("SQ_ACTIVE_INST_VALU"):
            # lanes enabled per vector instruction / 64.  Calibrated on a kernel that runs every instruction with all 64
            # lanes (bin/valu_peak under the same counters: SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = 64.00,
            # profiles/r03/r03_lane_counter_calibration.json)
            d["active_lane_frac"] = main["SQ_THREAD_CYCLES_VALU"] / (main["SQ_ACTIVE_INST_VALU"] * 64.0)
        if main.get("GRBM_GUI_ACTIVE"):
            cycles = main["GRBM_GUI_ACTIVE"] / 8.0                 # rocprofv3 sums the 8 XCDs
            d["launch_cycles_profiled"] = cycles
            d["simd_cycles_per_valu_inst"] = cycles * N_SIMD / insts
            d["valu_issue_frac_at_profiled_clock"] = 2.0 * insts / (cycles * N_SIMD)
        if launch_ms:
            d["valu_issue_frac"] = 2.0 * insts / (N_SIMD * NOMINAL_GHZ * 1e9 * launch_ms * 1e-3)
    dp = dp_instruction_counts(main)
    if dp is not None:
        d["dp_insts_per_launch"], d["dp_trans_insts_per_launch"] = dp
        if insts:
            d["dp_share_executed"] = (dp[0] + dp[1]) / insts
    return d


def dp_instruction_counts(main):
    """(add + mul + fma, transcendental) double-precision wave-instructions of a launch from the "f64" pass, or None without it."""
    if not all(k in main for k in F64_COUNTERS):
        return None
    return (main["SQ_INSTS_VALU_ADD_F64"] + main["SQ_INSTS_VALU_MUL_F64"] + main["SQ_INSTS_VALU_FMA_F64"], main["SQ_INSTS_VALU_TRANS_F64"])


def collect(cfg, passes=("sq", "fetch", "write"), reps=2, timeout=420, keep_root=None):
    """Run the passes; returns the record (see module docstring).  Raises on the first failing pass."""
    rec = {"config": dict(cfg), "key": config_key(cfg["scene_id"], cfg["width"], cfg["height"], cfg["samples"], cfg["bounces"], cfg["precision"],
                                                  cfg.get("schedule", "sorted"), cfg.get("scene_source", "grid"), cfg.get("shard")),
           "counters": {}, "dispatches": {}, "passes": [], "collected_unix": int(time.time())}
    for name in passes:
        keep = os.path.join(keep_root, name) if keep_root else None
        means, counts, build_id, ms = run_pass(cfg, PASSES[name], reps, timeout, keep, rec["passes"])
        rec["passes"][-1]["pass"] = name
        if not means.get("main"):
            raise RuntimeError("pass %s: no counters for the main launch" % name)
        if rec.get("build_id") not in (None, build_id):
            raise RuntimeError("passes ran on different library builds")
        rec["build_id"] = build_id
        if name == "sq":
            rec["profiled_render_ms"] = ms
        for cls, c in means.items():
            rec["counters"].setdefault(cls, {}).update(c)
            rec["dispatches"].setdefault(cls, {}).update(counts[cls])
    main = rec["counters"].get("main", {})
    rec["derived_main"] = derive(main)
    if "FETCH_SIZE" in main and "WRITE_SIZE" in main:
        # rocprofv3 prints both in KB.  FETCH_SIZE counts 128-byte reads of a wide streaming load as 64 (guide, HBM):
        # doubled = an upper bound; this path's reads are 16-byte records gathered in cost order, so both are kept.
        rec["traffic_main"] = {"fetch_bytes_raw": main["FETCH_SIZE"] * 1024.0, "write_bytes": main["WRITE_SIZE"] * 1024.0,
                               "hbm_bytes": (2.0 * main["FETCH_SIZE"] + main["WRITE_SIZE"]) * 1024.0,
                               "note": "separate --pmc passes; hbm_bytes = 2 x FETCH_SIZE (gfx950 correction, upper bound for gathers) + WRITE_SIZE"}
        step = {k: sum(rec["counters"].get(c, {}).get(k, 0.0) for c in rec["counters"]) for k in ("FETCH_SIZE", "WRITE_SIZE")}
        rec["traffic_step"] = {"fetch_bytes_raw": step["FETCH_SIZE"] * 1024.0, "write_bytes": step["WRITE_SIZE"] * 1024.0}
    return rec


def load_record(path, key, build_id):
    """The committed record for `key`, or None when there is none or it was measured on another build."""
    if not os.path.exists(path):
        return None
    rec = json.load(open(path)).get(key)
    if not rec or rec.get("build_id") != build_id:
        return None
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True, help="JSON file of records keyed by configuration (updated in place)")
    ap.add_argument("--scene_id", type=int, default=3); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--samples", type=int, default=100); ap.add_argument("--bounces", type=int, default=50); ap.add_argument("--precision", type=int, default=32)
    ap.add_argument("--schedule", default="sorted"); ap.add_argument("--scene_source", default="grid"); ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--passes", default="sq,fetch,write"); ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--keep", default=None, help="keep the raw rocprofv3 trees under this directory")
    ap.add_argument("--shard", default="", help="rank,nranks,strip_rows: only that rank's row strips of the frame")
    a = ap.parse_args()
    cfg = {k: getattr(a, k) for k in ("scene_id", "width", "height", "samples", "bounces", "precision", "schedule", "scene_source", "threads")}
    if a.shard:
        cfg["shard"] = a.shard
    rec = collect(cfg, tuple(a.passes.split(",")), a.reps, keep_root=a.keep)
    data = json.load(open(a.out)) if os.path.exists(a.out) else {}
    data[rec["key"]] = rec
    json.dump(data, open(a.out, "w"), indent=1, sort_keys=True)
    print(json.dumps({"key": rec["key"], "build_id": rec["build_id"], "derived_main": rec["derived_main"], "traffic_main": rec.get("traffic_main")}))


if __name__ == "__main__":
    main()
