#!/usr/bin/env python3
"""bench.py -- headline benchmark of the HIP render path.

Metric (BASELINE.json): Mrays/s = W x H x samples / render time, on
scene 3 (125 spheres, the reference's default: branch), 1920x1080, 100 spp, 50 bounces, fp32.
One "step" = one full render of the frame (for N GPUs: every rank renders its interleaved
row strips, then ONE exchange of the strips to rank 0 -- the exchange is inside the step).

    python bench.py [--gpus N] [--steps K] [--warmup W]                                  (a)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...        (b)

(a) without a launcher: N = 1 is the plain single-handle path; N > 1 drives the N GPUs of the node from THIS
    process through the library's own group (rtiow_group_*, csrc/rtiow_group.hip: one handle, stream and launch
    per device, then ncclSend/ncclRecv to device 0 over xGMI -- RCCL, loaded by the library -- or peer copies).
    `--devices a,b,..` lists the device of every rank; a device may repeat (ranks then share it: how the N-rank
    path is exercised on a one-GPU box).
(b) under torch.distributed.run (RANK/WORLD_SIZE/MASTER_* in the environment): one process per GPU, backend
    "nccl" (= RCCL), the strips gathered with torch.distributed.  Distributed even at WORLD_SIZE=1: process group,
    gather, all_reduce and barriers all execute.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      dominant kernel = the main launch of the render (render_persistent_kernel; the sorted schedule's
                prepass of 3 of the 100 samples is a separate, named launch) against the VECTOR-ALU ISSUE peak:
                `achieved` = SQ_INSTS_VALU wave-instructions of that launch x 64 lanes x 2 flop (one FMA issue slot
                each) / its mean HIP-event time; `peak` 157.3 TFLOP/s = one wave64 instruction per 2 cycles per
                SIMD-32; `frac` = achieved / peak = the share of the chip's vector issue slots the launch fills --
                EXECUTED work.  The counters come from rocprofv3 --pmc passes run by this very bench.py (child
                processes over the same library, before this process touches the GPU; `--pmc live`, default at
                N = 1) or from the committed record under profiles/ -- used only when its build id equals
                rtiow_build_id() of the loaded library (SHA-256 of sources + flags); otherwise null, never a stale
                number.  `algorithmic_TFLOPs` / `algorithmic_frac` = the reference's own arithmetic served per
                second: segments x (23 flop x every sphere + 120) + rays x 60 -- comparable work,
                not executed work (the grid walk tests a few spheres per segment), so it may exceed the peak.
                `traffic` = HBM bytes of the launch from the FETCH_SIZE / WRITE_SIZE passes (separate passes, gfx950
                x2 fetch correction as an upper bound), against `algorithmic_hbm_bytes_per_launch`.
                N > 1: every rank's main launch is rated the same way -- its own launch time, measured live on its device, against
                the SQ_INSTS_VALU of ITS shard from the committed per-shard records (profiles/pmc_records.json, keys ..._r<k>of<N>x<strip>,
                taken on one GPU by scripts/pmc_shard_records.py; a shard executes the same instructions on any device).  `frac` is
                then the fraction of the rank whose main launch is the longest (the one that bounds the step), `frac_per_rank` lists
                all; null for a rank without a record of the loaded build.
  cpu_baseline  the reference's serial tracer (oracle/_ref, built from the reference's own sources) or, if absent,
                the oracle's serial port, timed on this host (1 thread) on a bounded sample of the same workload;
                "config1" inside it is BASELINE.json configs[0] in full
  scaling_detail  what bounds strong scaling of this path: a pixel's samples are ONE sequential RNG
                chain, so no rank finishes before prepass + its longest chain x the latency of a lone ray's trip:
                "floor_ms"; per-rank kernel_ms and the time of the exchange
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import numpy as np  # noqa: E402

VALU_FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz = one wave64 FMA per 2 cycles per SIMD
VALU_FP64_PEAK_TFLOPS = 78.6      # v_fma_f64: half that rate (one wave64 instruction per 4 cycles)
HBM_PEAK_GBS = 8000.0
# What the part gives a stream of plain (un-packed) independent FMAs, measured with bin/valu_peak (profiles/archive/r01_valu_peak.json): v_fma_f32 80.85 TFLOP/s
# at 4 waves per SIMD, 84.65 at 8 (v_pk_fma_f32: 125-128); v_fma_f64 61.57 at 4 waves.  The render kernels run 5 (fp32) / 4 (fp64) waves per SIMD.
PRACTICAL_FMA_TFLOPS = {32: 82.0, 64: 61.57}
# fp64 kernels: double-precision add / mul / fma wave-instructions are charged 4 SIMD-32 cycles, double-precision transcendentals (v_rcp_f64, v_rsq_f64,
# v_sqrt_f64) 8, everything else -- generator steps, integer and fp32 grid walk, moves -- 2.  Which instructions those are comes from EXECUTED counts
# (the "f64" pass of scripts/pmc_passes.py: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64).  Only when a record lacks that pass: the share of v_*_f64 in the
# kernel's static ISA (tests/test_kernel_resources.py keeps it within +-0.04 of the compiler's), flagged "static" in the line.
FP64_KERNEL_DP_SHARE = 0.69
DP_CYCLES, DP_TRANS_CYCLES = 4.0, 8.0
# Configurations rendered after the headline's timed region, from the same loaded library, so that the round's claims about them are on the
# driver's clock too (N = 1 only): name -> (scene, W, H, spp, bounces, precision)
EXTRA_CONFIGS = [("fp64_headline", (3, 1920, 1080, 100, 50, 64)), ("scene1_487_spheres_1080p", (1, 1920, 1080, 100, 50, 32)),
                 ("baseline_config2_scene1_320x192_10spp_25b", (1, 320, 192, 10, 25, 32))]
PMC_RECORDS = os.environ.get("RTIOW_PMC_RECORDS") or os.path.join(ROOT, "profiles", "pmc_records.json")    # the committed counter records (the override is for tests)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene_id", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--samples", type=int, default=100)
    ap.add_argument("--bounces", type=int, default=50)
    ap.add_argument("--precision", type=int, default=32, choices=(32, 64))
    ap.add_argument("--threads", type=int, default=0, help="reference --threads (block T x T); 0 = library tiling")
    ap.add_argument("--scene_source", default="grid", choices=("grid", "lds", "scalar", "lds_exact"))
    ap.add_argument("--schedule", default="sorted", choices=("sorted", "persistent", "static"))
    ap.add_argument("--strip_rows", type=int, default=0, help="rows per interleaved strip; 0 = 8 for N <= 2, 2 for N >= 4 (profiles/archive/r01_strip_rows_sweep.txt)")
    ap.add_argument("--devices", default="", help="without a launcher: the device of every rank, e.g. 0,1,2,3 (default 0..N-1); a device may repeat")
    ap.add_argument("--gather", default="auto", choices=("auto", "rccl", "peer", "host"), help="transport of the in-library group (without a launcher, N > 1); auto falls back rccl -> peer -> host at gather time")
    ap.add_argument("--pmc", default="auto", choices=("auto", "live", "committed", "off"),
                    help="where roofline.frac's counters come from: rocprofv3 --pmc passes run now (live), the committed record if it matches "
                         "the loaded build (committed), neither (off); auto = live at N = 1, falling back to committed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the extra_configs object (fp64 headline, 487-sphere scene, BASELINE config [1]; N = 1 only)")
    ap.add_argument("--no-scaling-probe", action="store_true",
                    help="skip the 1-pixel lone-ray probe (it launches the main kernel by the same name: keeps rocprofv3 --stats averages clean)")
    return ap.parse_args()


def host_cpu():
    """Model name and logical core count of the host the CPU baseline runs on."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "host_logical_cores": os.cpu_count()}


def cpu_baseline(args):
    """Reference serial tracer on a bounded sample OF THE BENCHMARK FRAME ITSELF: every 16th row of the W x H view (same scene, camera, spp and
    bounces: 1/16 of the frame's rays, rows spread over sky, spheres and ground), 1 thread.  ~10 s of CPU work.  Without oracle/_ref: the
    oracle's serial port on the view at 1/4 linear resolution (the same number of rays)."""
    W, H, S, B = args.width, args.height, args.samples, args.bounces
    step = 16
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_serial_driver")
    if os.path.exists(drv) and H >= step:
        rows = len(range(step // 2, H, step))
        rays = W * rows * S
        t0 = time.perf_counter()
        r = subprocess.run([drv, str(args.scene_id), str(W), str(H), str(S), str(B), str(step)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        if r.returncode == 0 and r.stdout.startswith(b"P3"):
            sample = "scene %d, every %dth row of the %dx%d benchmark view itself (%d rows = 1/%d of its rays), %d spp, depth %d, serial fp64" % (
                args.scene_id, step, W, H, rows, step, S, B)
            return dict({"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference",
                         "sample": sample + "; reference src/InOneWeekend headers built by oracle/Makefile (oracle/ref_serial_driver.cc, row mode)", "seconds": dt}, **host_cpu())
    W, H = max(args.width // 4, 16), max(args.height // 4, 9)
    sample = "scene %d, %dx%d (the %dx%d view at 1/4 linear resolution), %d spp, depth %d, serial fp64" % (
        args.scene_id, W, H, args.width, args.height, S, B)
    rays = W * H * S
    from tests.oracle_lib import Oracle   # the oracle is only ever the checker / CPU baseline
    orc = Oracle()
    t0 = time.perf_counter()
    orc.render_serial(args.scene_id, W, H, S, B)
    dt = time.perf_counter() - t0
    return dict({"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port", "sample": sample + "; oracle serial port", "seconds": dt}, **host_cpu())


def cpu_baseline_config1():
    """BASELINE.json configs[0] in full: the reference's serial tracer, scene 1, 320x192, 10 spp,
    depth 25, one thread ("C1 in full")."""
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_serial_driver")
    cfg = (1, 320, 192, 10, 25)
    rays = cfg[1] * cfg[2] * cfg[3]
    what = "BASELINE.json configs[0]: serial CPU, scene 1, 320x192, 10 spp, depth 25, full frame"
    if os.path.exists(drv):
        t0 = time.perf_counter()
        r = subprocess.run([drv] + [str(x) for x in cfg], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        if r.returncode == 0 and r.stdout.startswith(b"P3"):
            return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference", "sample": what, "seconds": dt}
    from tests.oracle_lib import Oracle
    t0 = time.perf_counter()
    Oracle().render_serial(*cfg)
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port", "sample": what + "; oracle serial port", "seconds": dt}


def pmc_config(args):
    return {"scene_id": args.scene_id, "width": args.width, "height": args.height, "samples": args.samples, "bounces": args.bounces,
            "precision": args.precision, "schedule": args.schedule, "scene_source": args.scene_source, "threads": args.threads}


def under_a_profiler():
    """bench.py itself running below rocprofv3 (scripts/refresh_profiles.sh): no nested profiler children."""
    return any(k.startswith("ROCPROFILER_") or k.startswith("ROCP_") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")


def live_passes(precision, traffic=True):
    return ("sq",) + (("fetch", "write") if traffic else ()) + (("f64",) if precision == 64 else ())


def pmc_live(args):
    """The rocprofv3 --pmc passes of this configuration, run NOW as child processes over the library this bench is
    about to load (must be called before this process initialises the GPU).  Returns (record | None, note)."""
    import pmc_passes
    try:
        rec = pmc_passes.collect(pmc_config(args), passes=live_passes(args.precision), reps=2, timeout=420)
    except Exception as e:                       # no rocprofv3, no permission, a failing pass: the bench line still prints
        return None, "live passes failed: %s" % str(e)[:300]
    return rec, "rocprofv3 --pmc child processes of this bench.py run (scripts/pmc_passes.py), %s" % ", ".join(
        "%s %.1f s" % (p.get("pass") or p["cmd"].split("--pmc ")[1].split()[0], p["seconds"]) for p in rec["passes"])


def pmc_committed(args, build_id):
    import pmc_passes
    key = pmc_passes.config_key(args.scene_id, args.width, args.height, args.samples, args.bounces, args.precision, args.schedule, args.scene_source)
    rec = pmc_passes.load_record(PMC_RECORDS, key, build_id)
    if rec is None:
        return None, "no record for %s taken on build %s... in %s" % (key, build_id[:12], os.path.relpath(PMC_RECORDS, ROOT))
    return rec, "committed record %s[%s], build id matches the loaded library" % (os.path.relpath(PMC_RECORDS, ROOT), key)


def shard_records(args, world, build_id):
    """N > 1: the committed counter record of every rank's shard (scripts/pmc_shard_records.py), or None where there is none for the loaded build."""
    import pmc_passes
    if args.pmc == "off":
        return [None] * world, "--pmc off"
    recs = []
    for k in range(world):
        key = pmc_passes.config_key(args.scene_id, args.width, args.height, args.samples, args.bounces, args.precision, args.schedule, args.scene_source,
                                    shard=(k, world, args.strip_rows))
        recs.append(pmc_passes.load_record(PMC_RECORDS, key, build_id))
    have = sum(r is not None for r in recs)
    note = ("per-shard records %s[..._r<k>of%dx%d] of the loaded build (SQ_INSTS_VALU of a shard, taken on one GPU: scripts/pmc_shard_records.py) for %d of %d ranks; "
            "each rank's launch time is its own, measured in this run" % (os.path.relpath(PMC_RECORDS, ROOT), world, args.strip_rows, have, world))
    if have == 0:
        note = "no per-shard record of build %s... in %s (N > 1 uses committed records: scripts/pmc_shard_records.py)" % (build_id[:12], os.path.relpath(PMC_RECORDS, ROOT))
    return recs, note


def lone_ray_trip_us(rt, device_index, prec, scene, args):
    """Latency of one path segment of a ray that has a wave to itself (what bounds the end of every
    shard): a 1-pixel frame of this scene, a few hundred samples, HIP-event time /
    segments.  The pixel is the frame's centre pixel region seen through the same camera maths."""
    S = 400
    with rt.Renderer(device_index, prec) as r:
        r.set_camera(rt.camera(prec, 1, 1, S, args.bounces))
        r.set_scene(scene)
        r.init_rng(1227)
        segs = r.count_segments(0)
        best = min(r.render(0) for _ in range(5))
    return (best * 1e3 / segs if segs else None), segs


def issue_fraction(valu_insts, launch_ms, precision, cus=256, clock_mhz=2400, dp=None):
    """(frac, cycles_per_inst): share of the SIMD-32 issue cycles the launch's vector instructions take.  fp32: 2 cycles per wave64 instruction.
    fp64: `dp` = (double-precision add/mul/fma, double-precision transcendental) wave-instructions that EXECUTED (pmc_passes.dp_instruction_counts)
    are charged 4 and 8 cycles, the others 2; without those counts the static ISA share FP64_KERNEL_DP_SHARE stands in for them.
    The device's compute units and nominal clock come from the library (rtiow_stats.num_cus / clock_mhz = hipDeviceProp_t), not from a constant."""
    if precision == 32:
        cpi = 2.0
    elif dp is not None:
        cpi = (2.0 * (valu_insts - dp[0] - dp[1]) + DP_CYCLES * dp[0] + DP_TRANS_CYCLES * dp[1]) / valu_insts
    else:
        cpi = 2.0 * (1.0 - FP64_KERNEL_DP_SHARE) + DP_CYCLES * FP64_KERNEL_DP_SHARE
    return cpi * valu_insts / (cus * 4 * clock_mhz * 1e6 * launch_ms * 1e-3), cpi


def dp_counts(rec, valu_insts):
    """The executed double-precision counts of a record, scaled to `valu_insts` (the f64 pass counts its own SQ_INSTS_VALU: runs differ by a few 1e-4)."""
    import pmc_passes
    main = rec["counters"].get("main", {})
    dp = pmc_passes.dp_instruction_counts(main)
    if dp is None or not main.get("SQ_INSTS_VALU"):
        return None
    return dp


def device_of(st):
    """Compute units and nominal clock of the device the handle runs on; MI355X figures when an older library does not report them."""
    return {"cus": int(st.get("num_cus") or 256), "clock_mhz": int(st.get("clock_mhz") or 2400)}


def run_extra_configs(rt, device_index, args, extra_pmc):
    """fp64 headline, the 487-sphere scene at 1080p and BASELINE config [1], each rendered a few times by the library this bench has loaded
    (after the headline's timed region).  `frac` / `active_lane_frac` from this run's own sq pass (extra_pmc) or the committed record of THIS build."""
    import pmc_passes
    out = []
    for name, (scene_id, W, H, S, B, prec) in EXTRA_CONFIGS:
        with rt.Renderer(device_index, prec) as r:
            r.set_camera(rt.camera(prec, W, H, S, B))
            r.set_scene(rt.build_scene(scene_id, prec))
            r.set_scene_source(getattr(rt, SOURCES[args.scene_source]))
            r.set_schedule(getattr(rt, SCHEDULES[args.schedule]))
            r.init_rng(1227)
            for _ in range(2):
                r.render(0)
            r.synchronize()
            steps = 8 if W * H >= 1000000 else 20
            main_ms = []
            t0 = time.perf_counter()
            for _ in range(steps):
                r.render(0, sync=True)
                main_ms.append(r.stats()["main_ms"])
            r.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            st = r.stats()
        rec = extra_pmc.get(name)
        note = "this run's rocprofv3 --pmc sq pass" if rec else None
        if rec is None:
            key = pmc_passes.config_key(scene_id, W, H, S, B, prec, args.schedule, args.scene_source)
            rec = pmc_passes.load_record(PMC_RECORDS, key, rt.build_id())
            note = "committed record of this build" if rec else None
        frac = lanes = dp_basis = None
        if rec and rec.get("build_id") == rt.build_id():
            main = rec["counters"].get("main", {})
            if main.get("SQ_INSTS_VALU"):
                frac = round(issue_fraction(main["SQ_INSTS_VALU"], float(np.mean(main_ms)), prec, dp=dp_counts(rec, main["SQ_INSTS_VALU"]) if prec == 64 else None, **device_of(st))[0], 4)
                if prec == 64:
                    dp_basis = "executed (SQ_INSTS_VALU_*_F64)" if dp_counts(rec, main["SQ_INSTS_VALU"]) else "static ISA share"
            d = pmc_passes.derive(main)
            lanes = round(d["active_lane_frac"], 4) if "active_lane_frac" in d else None
        out.append({"name": name, "workload": "scene %d (%d spheres), %dx%d, %d spp, %d bounces, fp%d" % (scene_id, st["num_spheres"], W, H, S, B, prec),
                    "steps": steps, "ms_per_step": round(ms, 4), "value": round(float(W) * H * S / (ms * 1e-3) / 1e6, 3), "unit": "Mrays/s",
                    "main_launch_ms": round(float(np.mean(main_ms)), 4), "prepass_ms": round(float(st["prepass_ms"]), 4),
                    "frac": frac, "active_lane_frac": lanes, "counters_from": note, **({"dp_cycles_from": dp_basis} if prec == 64 else {})})
    return out


SOURCES = {"grid": "SCENE_GRID", "lds": "SCENE_LDS", "scalar": "SCENE_SCALAR", "lds_exact": "SCENE_LDS_EXACT"}
SCHEDULES = {"sorted": "SCHED_SORTED", "persistent": "SCHED_PERSISTENT", "static": "SCHED_STATIC"}


def rank_fractions(args, ranks, dev):
    """N > 1: every rank's main launch against the issue peak of ITS device -- `ranks` = [{"pmc": record of that rank's shard (this build) or None,
    "main_ms": mean HIP-event time of its main launch, measured on its own device}].  Returns (per-rank list, index of the rank whose main launch
    is the longest = the one that bounds the step)."""
    import pmc_passes
    out = []
    for k, rk in enumerate(ranks):
        rec, mms = rk.get("pmc"), rk.get("main_ms")
        entry = {"rank": k, "main_launch_ms": round(mms, 4) if mms else None, "frac": None, "valu_wave_insts": None, "active_lane_frac": None}
        main = rec["counters"].get("main", {}) if rec else {}
        if main.get("SQ_INSTS_VALU") and mms:
            insts = main["SQ_INSTS_VALU"]
            frac, _ = issue_fraction(insts, mms, args.precision, dp=dp_counts(rec, insts) if args.precision == 64 else None, **dev)
            d = pmc_passes.derive(main)
            entry.update({"frac": round(frac, 4), "valu_wave_insts": insts, "active_lane_frac": round(d["active_lane_frac"], 4) if "active_lane_frac" in d else None})
        out.append(entry)
    timed = [e for e in out if e["main_launch_ms"]]
    slowest = max(timed, key=lambda e: e["main_launch_ms"])["rank"] if timed else 0
    return out, slowest


def roofline_object(args, st, segments_main_rank0, main_ms, pmc, pmc_note, world, ranks=None):
    """The dominant kernel (the main launch: rank 0's at N = 1, the slowest rank's at N > 1) against the vector-issue peak; see the module docstring."""
    prec, S = args.precision, args.samples
    nspheres = st["num_spheres"]
    my_rays = float(st["primary_rays"])
    per_seg = 23.0 * nspheres + 120.0
    peak = VALU_FP32_PEAK_TFLOPS if prec == 32 else VALU_FP64_PEAK_TFLOPS
    rays_main = my_rays * (S - st["prepass_samples"]) / S
    flops = float(segments_main_rank0) * per_seg + rays_main * 60.0       # the main launch alone
    mms = float(np.mean(main_ms))
    algorithmic = flops / (mms * 1e-3) / 1e12
    # algorithmic HBM bytes of the main launch, per pixel: the framebuffer write (3 T) + its starting state:
    # the 48/64-byte hand-over record and a 4-byte order entry (sorted schedule) or the 24-byte RNG state
    my_pixels = my_rays / S
    state_bytes = ((48 if prec == 32 else 64) + 4) if st["phases"] == 2 else 24
    fb_bytes = my_pixels * (3 * (4 if prec == 32 else 8) + state_bytes)
    achieved = frac = traffic = fetch_b = write_b = None
    issued = saturation = per_rank = None
    dp_basis = None
    dev = device_of(st)
    dominant_rank = 0
    if world > 1 and ranks:
        per_rank, dominant_rank = rank_fractions(args, ranks, dev)
        pmc = ranks[dominant_rank].get("pmc")             # the record the headline fraction is computed from
        mms_dom = ranks[dominant_rank].get("main_ms") or mms
    else:
        mms_dom = mms
        if world > 1:
            pmc = None
    if pmc is not None:
        import pmc_passes
        main = pmc["counters"].get("main", {})
        d = pmc_passes.derive(main, launch_ms=mms_dom)
        if d.get("valu_wave_insts_per_launch"):
            # frac = share of the SIMD-32 issue slots filled (2 cycles per wave64 instruction, 1024 SIMDs, 2.4 GHz); achieved = that share of
            # the peak, i.e. one FMA issue slot = 64 lanes x 2 flop (fp64 FMA slots are half as many per second: the fp64 peak above)
            dp = dp_counts(pmc, d["valu_wave_insts_per_launch"]) if prec == 64 else None
            dp_basis = None if prec == 32 else ("executed" if dp is not None else "static")
            frac, cycles_per_inst = issue_fraction(d["valu_wave_insts_per_launch"], mms_dom, prec, dp=dp, **dev)
            # the peaks are MI355X's (256 CUs at 2.4 GHz): on any other part the fraction still holds, the TFLOP/s figures are scaled with it
            peak = peak * dev["cus"] * dev["clock_mhz"] / (256.0 * 2400.0)
            achieved = frac * peak
            issued = {"valu_wave_insts_per_launch": d["valu_wave_insts_per_launch"],
                      "valu_issue_frac": round(d["valu_issue_frac"], 4),    # every instruction at 2 cycles (the fp32 figure; a lower bound in fp64)
                      "cycles_per_inst_charged": round(cycles_per_inst, 3),
                      "dp_insts_per_launch": d.get("dp_insts_per_launch"), "dp_trans_insts_per_launch": d.get("dp_trans_insts_per_launch"),
                      "dp_share_executed": round(d["dp_share_executed"], 4) if "dp_share_executed" in d else None,
                      "simd_cycles_per_valu_inst_profiled": round(d["simd_cycles_per_valu_inst"], 3) if "simd_cycles_per_valu_inst" in d else None,
                      "valu_issue_frac_at_profiled_clock": round(d["valu_issue_frac_at_profiled_clock"], 4) if "valu_issue_frac_at_profiled_clock" in d else None,
                      "active_lane_frac": round(d["active_lane_frac"], 4) if "active_lane_frac" in d else None,
                      "salu_insts_per_launch": main.get("SQ_INSTS_SALU"),
                      "wave_cycles_split": {k: main.get(k) for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA") if main.get(k)},   # quad-cycles of resident waves: waiting (s_waitcnt), waiting for an issue slot, issuing vector / scalar instructions
                      "prepass_valu_wave_insts": pmc["counters"].get("prepass", {}).get("SQ_INSTS_VALU")}
            if main.get("SQ_ACTIVE_INST_VALU"):
                # How busy the vector pipes are, from the counters alone (no peak, no clock): a resident wave is "issuing a vector instruction"
                # for SQ_ACTIVE_INST_VALU quad-cycles; per instruction that is how long one occupies its SIMD's issue stage.
                busy = 4.0 * main["SQ_ACTIVE_INST_VALU"] / d["valu_wave_insts_per_launch"]
                saturation = {"simd_cycles_busy_per_valu_inst": round(busy, 3),
                              "simd_valu_busy_share": round(4.0 * main["SQ_ACTIVE_INST_VALU"] / (d["launch_cycles_profiled"] * pmc_passes.N_SIMD), 4) if d.get("launch_cycles_profiled") else None,
                              "what": "4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = SIMD cycles a vector instruction of this launch occupies its issue stage (a stream of independent "
                                      "v_fma_f32 measures 3.9 at five waves per SIMD, v_pk_* 5.9, v_fma_f64 4.6: bin/valu_cost), and the same busy time as a share of all SIMD cycles "
                                      "of the profiled launch (GRBM_GUI_ACTIVE; waves overlap in the stage, so it can pass 1): at ~1 the SIMDs have no issue cycles left -- fewer "
                                      "wave-instructions or more lanes per instruction are the only levers"}
        t = pmc.get("traffic_main")
        if t:
            traffic, fetch_b, write_b = t["hbm_bytes"], t["fetch_bytes_raw"], t["write_bytes"]
    kernel = {"static": "render_kernel", "persistent": "render_persistent_kernel", "sorted": "render_solo_kernel" if st["solo_waves"] else "render_persistent_kernel"}[args.schedule]
    fp64_text = ""
    if prec == 64:
        fp64_text = ("; in this fp64 kernel the double-precision add / mul / fma instructions that executed (SQ_INSTS_VALU_*_F64, the run's own \"f64\" pass) are charged 4 and its "
                     "double-precision transcendentals 8" if dp_basis != "static" else
                     "; in this fp64 kernel %.0f %% of them are double-precision instructions charged 4 (static ISA share, tests/test_kernel_resources.py: no executed counts in the record)"
                     % (100 * FP64_KERNEL_DP_SHARE))
    out = {"bound": "valu", "achieved": round(achieved, 3) if achieved is not None else None, "peak": peak, "unit": "TFLOP/s",
            "frac": round(frac, 4) if frac is not None else None, "traffic": traffic,
            "achieved_is": "EXECUTED vector issue: frac = SIMD-32 issue cycles of the main launch's SQ_INSTS_VALU wave-instructions / (1024 SIMDs x 2.4 GHz x launch time), "
                           "a wave64 instruction charged 2 cycles" + fp64_text +
                           "; achieved = frac x peak (the FMA slots of this precision: 64 lanes x 2 flop per 2 (fp32) / 4 (fp64) cycles per SIMD). null: no counters for THIS build",
            "issue_saturation": saturation,
            "counters_from": pmc_note, "build_id": pmc.get("build_id") if pmc else None, "issued": issued,
            "device": dev,    # what frac is rated against: 4 SIMDs per compute unit at the nominal clock, both from hipDeviceProp_t through the library
            "algorithmic_TFLOPs": round(algorithmic, 3), "algorithmic_frac": round(algorithmic / peak, 4),
            "algorithmic_is": "the reference's own sphere loop served per second: 23 flop x every sphere x every segment + 120 per segment + 60 per ray "
                              "(SURVEY.md 8d); the grid walk finds the same hits testing a few spheres per segment, so this is comparable work, not executed work",
            "fetch_bytes": fetch_b, "write_bytes": write_b,   # raw FETCH_SIZE / WRITE_SIZE of the main launch
            "kernel": "%s<%s>" % (kernel, "float" if prec == 32 else "double"),
            "launch_ms_mean": round(mms, 4), "launch_ms_min": round(float(np.min(main_ms)), 4),
            "algorithmic_flops_per_launch": flops, "segments_in_launch": int(segments_main_rank0),
            "samples_in_launch": int(S - st["prepass_samples"]),
            "algorithmic_hbm_bytes_per_launch": fb_bytes,
            "hbm_achieved_GBps": round((traffic if traffic else fb_bytes) / (mms * 1e-3) / 1e9, 3), "hbm_peak_GBps": HBM_PEAK_GBS}
    if prec == 64:
        out["dp_cycles_from"] = dp_basis
    if per_rank is not None:
        out["frac_per_rank"] = per_rank
        out["frac_is_rank"] = dominant_rank        # the rank whose main launch is the longest; algorithmic_* and launch_ms_* stay rank 0's
        fr = [e["frac"] for e in per_rank if e["frac"] is not None]
        out["frac_max_over_ranks"] = max(fr) if fr else None
        out["frac_mean_over_ranks"] = round(float(np.mean(fr)), 4) if fr else None
    return out


def emit(json_fd, args, ctx):
    """Rank 0: the contract line."""
    W, H, S, B, prec = args.width, args.height, args.samples, args.bounces, args.precision
    st = ctx["stats0"]
    rays = float(W) * H * S
    ms_per_step = ctx["elapsed"] / args.steps * 1e3
    value = rays / (ms_per_step * 1e-3) / 1e6
    kms = float(np.mean(ctx["kernel_ms"]))
    per_seg = 23.0 * st["num_spheres"] + 120.0
    flops_step = float(ctx["segments0"]) * per_seg + float(st["primary_rays"]) * 60.0
    peak = VALU_FP32_PEAK_TFLOPS if prec == 32 else VALU_FP64_PEAK_TFLOPS
    world = ctx["world"]
    line = {
        "metric": "Mrays/s (= W x H x samples / render time)",
        "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32" if prec == 32 else "f64", "data": "synthetic",
        "config": {"workload": "scene %d (%d spheres), %dx%d, %d spp, %d bounces, XORWOW seed 1227" % (args.scene_id, st["num_spheres"], W, H, S, B),
                   "scene_id": args.scene_id, "spheres": st["num_spheres"], "width": W, "height": H, "samples": S, "bounces": B,
                   "threads": args.threads, "scene_source": args.scene_source, "schedule": args.schedule,
                   "sharding": ctx["sharding"], "backend": ctx["backend"], "host": ctx["host"]},
        "kernel_ms_mean": round(kms, 4), "kernel_ms_min": round(float(np.min(ctx["kernel_ms"])), 4),
        "kernel_ms_mean_max_over_ranks": round(ctx["kernel_mean_max"], 4),
        "segments_per_ray": round(ctx["segments_total"] / rays, 4),
        "roofline": roofline_object(args, st, ctx["segments_main0"], ctx["main_ms"], ctx["pmc"], ctx["pmc_note"], world, ctx.get("ranks")),
        "step": {"launches": "prepass (%d spp) + cost sort + main" % st["prepass_samples"] if st["phases"] == 2 else "main",
                 "kernel_ms_mean": round(kms, 4), "prepass_ms": round(float(st["prepass_ms"]), 4),
                 "scene_prepare_ms": round(float(st["scene_prepare_ms"]), 3),   # host: screening table + grid plan, once per scene, before the first render's start event
                 "place_ms": round(float(st["place_ms"]), 4), "staged_stores": int(st["staged_stores"]),   # sorted schedule: pixels stored in slot order, place_pixels_kernel writes the image in whole lines
                 "solo_waves": int(st["solo_waves"]),   # > 0: a partly filled GPU (shard, small frame), render_solo_kernel
                 "algorithmic_flops": flops_step, "algorithmic_TFLOPs": round(flops_step / (kms * 1e-3) / 1e12, 3),
                 "algorithmic_frac": round(flops_step / (kms * 1e-3) / 1e12 / peak, 4)},
    }
    trip_us = ctx["trip_us"]
    line["scaling_detail"] = {
        "floor_ms": round(ctx["floor_ms"], 4), "floor_is": "prepass_ms + longest per-pixel chain of the main launch x the trip latency of a lone ray on an idle GPU (1-pixel probe), max over ranks: "
                                                            "what a rank reaches if its longest chain runs undisturbed from the first trip; shards run it at 1.6-2x that "
                                                            "latency, two heavy pixels per solo wave beside the loaded SIMDs (DESIGN.md sections 4.3, 5)",
        "efficiency_vs_floor": round(ctx["floor_ms"] / ms_per_step, 4) if ctx["floor_ms"] and ms_per_step else None,   # floor_ms / ms_per_step: 1 = the step is as short as its longest chain allows
        "longest_chain_segments": ctx["chain_max"], "lone_ray_trip_us": round(trip_us, 4) if trip_us else None,
        "lone_ray_probe": "1x1 frame, 400 spp, %d segments" % ctx["trip_segments"],
        "kernel_ms_per_rank": ctx["kernel_ms_per_rank"], "gather_ms_per_rank": ctx["gather_ms_per_rank"],
        "gather_ms": round(ctx["gather_ms"], 4) if ctx["gather_ms"] is not None else None,
        "gather_transport": ctx.get("gather_transport"),
        "gather_bytes_total": int(W) * H * 3 * (4 if prec == 32 else 8) if world > 1 or ctx["backend"] else 0}
    if ctx.get("extra_configs") is not None:
        line["extra_configs"] = ctx["extra_configs"]
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args)
        line["cpu_baseline"]["config1"] = cpu_baseline_config1()
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(line) + "\n").encode())


def run_group(args, json_fd):
    """N > 1 without a launcher: ONE process, the library's own group over the node's GPUs (include/rtiow.h)."""
    import raytracingincuda_amd as rt
    N = args.gpus
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(N))
    if len(devices) != N:
        raise SystemExit("--devices lists %d devices for --gpus %d" % (len(devices), N))
    prec = args.precision
    W, H, S, B = args.width, args.height, args.samples, args.bounces
    scene = rt.build_scene(args.scene_id, prec)
    cam = rt.camera(prec, W, H, S, B)
    gather = {"auto": rt.GATHER_AUTO, "rccl": rt.GATHER_RCCL, "peer": rt.GATHER_PEER, "host": rt.GATHER_HOST}[args.gather]
    g = rt.RendererGroup(N, prec, args.strip_rows, gather, devices)      # raises when the node has fewer GPUs or the HIP library is missing
    g.set_camera(cam)
    g.set_scene(scene)
    g.set_scene_source(getattr(rt, SOURCES[args.scene_source]))
    g.set_schedule(getattr(rt, SCHEDULES[args.schedule]))
    g.init_rng(1227)                                    # untimed, like main.cu:326-330
    members = [g.member(k) for k in range(N)]
    segments = [m.count_segments(args.threads) for m in members]          # untimed; also a first warm launch per device
    chains = [int(m.stats()["max_chain_main"]) for m in members]
    segments_main0 = int(members[0].stats()["segments_main"])
    trip_us, trip_segments = (None, 0) if args.no_scaling_probe else lone_ray_trip_us(rt, devices[0], prec, scene, args)

    for _ in range(args.warmup):
        g.render(args.threads)
        g.gather()
    for m in members:
        m.synchronize()
    kernel_ms, main_ms, gather_ms, per_rank, main_per_rank = [], [], [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kernel_ms.append(g.render(args.threads))         # every device's launches are enqueued, then every stop event awaited: max over devices
        g.gather()                                       # the one exchange + de-interleave on device 0 (blocks on its stop event)
        gs = g.stats()
        per_rank.append(gs["kernel_ms"])
        gather_ms.append(gs["gather_ms"])
        main_per_rank.append([m.stats()["main_ms"] for m in members])
        main_ms.append(main_per_rank[-1][0])
    for m in members:
        m.synchronize()
    elapsed = time.perf_counter() - t0
    gs = g.stats()
    sts = [m.stats() for m in members]
    floor = max(float(sts[k]["prepass_ms"]) + chains[k] * (trip_us or 0.0) * 1e-3 for k in range(N))
    recs, pmc_note = shard_records(args, N, rt.build_id())
    main_mean = np.mean(np.array(main_per_rank), axis=0)
    ranks = [{"pmc": recs[k], "main_ms": float(main_mean[k])} for k in range(N)]
    ctx = {"world": N, "elapsed": elapsed, "kernel_ms": kernel_ms, "main_ms": main_ms, "stats0": sts[0], "segments0": segments[0],
           "segments_main0": segments_main0, "segments_total": float(sum(segments)), "kernel_mean_max": float(np.mean(kernel_ms)),
           "kernel_ms_per_rank": [round(float(x), 4) for x in np.mean(np.array(per_rank), axis=0)],
           "gather_ms_per_rank": None, "gather_ms": float(np.mean(gather_ms)),
           "gather_transport": {rt.GATHER_RCCL: "rccl %d (ncclSend/ncclRecv to device 0)" % gs["rccl_version"], rt.GATHER_PEER: "peer copies" + (": " + gs["transport_note"] if gs["transport_note"] else ""),
                               rt.GATHER_HOST: "host-staged copies" + (": " + gs["transport_note"] if gs["transport_note"] else "")}.get(gs["gather_mode"], "none"),
           "floor_ms": floor, "chain_max": max(chains), "trip_us": trip_us, "trip_segments": trip_segments,
           "sharding": "interleaved %d-row strips over %d ranks, one exchange to rank 0 inside the step" % (args.strip_rows, N),
           "backend": "rtiow_group (in-library RCCL / peer copies)", "host": "one process, devices %s" % ",".join(str(d) for d in devices),
           "pmc": None, "pmc_note": pmc_note, "ranks": ranks}
    emit(json_fd, args, ctx)
    g.close()


def run_ranks(args, json_fd, world, rank, local_rank, distributed, pmc, pmc_note, extra_pmc=None):
    """N = 1, or one process per GPU under torch.distributed.run."""
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # RTIOW_BENCH_BACKEND=gloo is a REHEARSAL mode for boxes with fewer GPUs than ranks (ranks
    # share devices, the gather bounces through the host); the real multi-GPU run uses RCCL.
    backend = os.environ.get("RTIOW_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    if not distributed and args.devices:
        device_index = int(args.devices.split(",")[0])
    torch.cuda.set_device(device_index)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    import raytracingincuda_amd as rt
    from raytracingincuda_amd.distributed import StripGather

    prec = args.precision
    tdtype = torch.float32 if prec == 32 else torch.float64
    W, H, S, B = args.width, args.height, args.samples, args.bounces
    scene = rt.build_scene(args.scene_id, prec)
    cam = rt.camera(prec, W, H, S, B)

    r = rt.Renderer(device_index, prec)
    stream = torch.cuda.current_stream()
    r.set_stream(stream.cuda_stream)
    r.set_camera(cam)
    r.set_scene(scene)
    r.set_scene_source(getattr(rt, SOURCES[args.scene_source]))
    r.set_schedule(getattr(rt, SCHEDULES[args.schedule]))
    r.set_shard(rank, world, args.strip_rows)
    gather = StripGather(W, H, rank, world, args.strip_rows, tdtype, "cuda:%d" % device_index, stage_via_cpu=(backend != "nccl"),
                         always_collective=distributed)
    view = gather.local_view()
    r.bind_framebuffer(view.data_ptr(), view.numel() * view.element_size())
    r.init_rng(1227)                                   # untimed, like main.cu:326-330
    segments = r.count_segments(args.threads)          # untimed; also a first warm launch
    segments_main0 = int(r.stats()["segments_main"])
    chain_main = int(r.stats()["max_chain_main"])      # this rank's longest per-pixel chain in the main launch
    trip_us, trip_segments = (None, 0) if args.no_scaling_probe else lone_ray_trip_us(rt, device_index, prec, scene, args)

    rank_recs = None
    if world > 1:
        pmc = None
        rank_recs, pmc_note = shard_records(args, world, rt.build_id())     # every rank reads the (small) committed file; rank 0 uses them
    elif args.pmc == "off":
        pmc, pmc_note = None, "--pmc off"
    elif pmc is None and args.pmc in ("auto", "committed"):      # no live passes (asked not to, or they failed): the committed record, if it is of THIS build
        rec, note = pmc_committed(args, rt.build_id())
        pmc, pmc_note = rec, (pmc_note + "; " if pmc_note else "") + note
    if pmc is not None and pmc.get("build_id") != rt.build_id():
        pmc, pmc_note = None, "counters were taken on build %s..., the loaded library is %s..." % (str(pmc.get("build_id"))[:12], rt.build_id()[:12])

    main_ms = []
    gather_events = []

    def step(timed):
        ms = r.render(args.threads, sync=timed)        # HIP events on the launch stream
        if timed:
            main_ms.append(r.stats()["main_ms"])        # the main launch alone (events around it)
        if distributed:
            if timed:                                   # events around the collective, on the stream it is enqueued on
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                gather.gather()
                e1.record(stream)
                gather_events.append((e0, e1))
            else:
                gather.gather()
        return ms

    for _ in range(args.warmup):
        step(False)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kernel_ms.append(step(True))
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gather_ms = float(np.mean([a.elapsed_time(b) for a, b in gather_events])) if gather_events else None

    st_local = r.stats()
    floor_local = float(st_local["prepass_ms"]) + chain_main * (trip_us or 0.0) * 1e-3
    if distributed:
        t = torch.tensor([elapsed, float(np.mean(kernel_ms)), float(segments), gather_ms or 0.0, floor_local, float(chain_main), float(np.mean(main_ms)) if main_ms else 0.0],
                         dtype=torch.float64, device="cuda")
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        per_rank = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(per_rank, t)
        elapsed, kernel_mean_max, segments_total = float(tmax[0]), float(tmax[1]), float(tsum[2])
        gather_ms_max, floor_ms, chain_max = float(tmax[3]), float(tmax[4]), int(tmax[5])
        kernel_ms_per_rank = [round(float(x[1]), 4) for x in per_rank]
        gather_ms_per_rank = [round(float(x[3]), 4) for x in per_rank]
        main_ms_per_rank = [float(x[6]) for x in per_rank]
    else:
        main_ms_per_rank = None
        kernel_mean_max, segments_total = float(np.mean(kernel_ms)), float(segments)
        gather_ms_max, floor_ms, chain_max = None, floor_local, chain_main
        kernel_ms_per_rank, gather_ms_per_rank = [round(float(np.mean(kernel_ms)), 4)], None

    extras = None
    if world == 1 and not distributed and not args.no_extra_configs:
        extras = run_extra_configs(rt, device_index, args, extra_pmc or {})       # after the timed region, same library
    if rank == 0:
        ctx = {"extra_configs": extras, "world": world, "elapsed": elapsed, "kernel_ms": kernel_ms, "main_ms": main_ms, "stats0": r.stats(), "segments0": segments,
               "segments_main0": segments_main0, "segments_total": segments_total, "kernel_mean_max": kernel_mean_max,
               "kernel_ms_per_rank": kernel_ms_per_rank, "gather_ms_per_rank": gather_ms_per_rank, "gather_ms": gather_ms_max,
               "gather_transport": ("torch.distributed gather, backend %s" % backend) if distributed else None,
               "floor_ms": floor_ms, "chain_max": chain_max, "trip_us": trip_us, "trip_segments": trip_segments,
               "sharding": "interleaved %d-row strips, gather to rank 0 inside the step" % args.strip_rows if distributed else "none",
               "backend": backend if distributed else None,
               "host": "one process per GPU (torch.distributed.run)" if distributed else "one process, device %d" % device_index,
               "pmc": pmc, "pmc_note": pmc_note,
               "ranks": [{"pmc": rank_recs[k], "main_ms": main_ms_per_rank[k]} for k in range(world)] if rank_recs is not None and main_ms_per_rank else None}
        emit(json_fd, args, ctx)
    r.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    # The driver reads ONE JSON line from stdout; libraries are chatty there (RCCL prints a version
    # banner on stdout when its first communicator is created).  Everything but that line goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torch.distributed.run the job is distributed whatever its size (see the docstring)
    distributed = world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ)
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.strip_rows <= 0:
        args.strip_rows = 8 if max(world, args.gpus) <= 2 else 2
    if args.gpus > 1 and not distributed:
        return run_group(args, json_fd)                 # no launcher: the in-library group drives the N GPUs from this process
    # counter passes of THIS run: child processes, started before this process initialises the GPU
    pmc, pmc_note, extra_pmc = None, None, {}
    if world == 1 and args.pmc in ("auto", "live"):
        if under_a_profiler():
            pmc_note = "bench.py itself runs below a profiler: no nested passes"
        else:
            pmc, pmc_note = pmc_live(args)
            if pmc is not None and not distributed and not args.no_extra_configs:      # one sq pass per extra configuration, same rule (child processes, now)
                import pmc_passes
                for name, (scene_id, W, H, S, B, prec) in EXTRA_CONFIGS:
                    cfg = dict(pmc_config(args), scene_id=scene_id, width=W, height=H, samples=S, bounces=B, precision=prec, threads=0)
                    try:
                        extra_pmc[name] = pmc_passes.collect(cfg, passes=live_passes(prec, traffic=False), reps=2, timeout=240)
                    except Exception:
                        pass
    run_ranks(args, json_fd, world, rank, local_rank, distributed, pmc, pmc_note, extra_pmc)


if __name__ == "__main__":
    main()
