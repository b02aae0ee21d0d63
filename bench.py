#!/usr/bin/env python3
"""bench.py -- headline benchmark of the HIP render path.

Metric (BASELINE.json): Mrays/s = W x H x samples / render time, on
scene 3 (125 spheres, the reference's default: branch), 1920x1080, 100 spp, 50 bounces, fp32.
One "step" = one full render of the frame (for N GPUs: every rank renders its interleaved
row strips, then one RCCL gather of the strips to rank 0 -- the gather is inside the step).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      dominant kernel = the main launch of the render (render_persistent_kernel; the
                sorted schedule's prepass of 3 of the 100 samples is a separate, named launch)
                against the VALU peak: algorithmic flops of that launch = segments*(23*N+120) +
                rays*60 (segments counted per launch on the device by
                rtiow_count_segments) / its mean HIP-event time over the timed steps.  The same
                figures for the whole step (all launches) are under "step".
  cpu_baseline  the reference's serial tracer (oracle/_ref, built from the reference's own
                sources) or, if absent, the oracle's serial port, timed on this host (1 thread)
                on a bounded sample of the same workload; "config1" inside it is BASELINE.json
                configs[0] in full (serial CPU, scene 1, 320x192, 10 spp, 25 bounces)
  scaling       what bounds strong scaling of this path: a pixel's samples are ONE
                sequential RNG chain, so no rank finishes before prepass + its longest chain x
                the latency of a lone ray's trip: "floor_ms" (measured here: the longest chain is
                counted on the device, the trip latency is timed on a 1-pixel frame); per-rank
                kernel_ms and gather_ms (events around the collective) when the run is distributed

Launched by torch.distributed.run (RANK/WORLD_SIZE in the environment) the run is distributed
even at WORLD_SIZE=1: process group "nccl" (= RCCL), the strip gather, the all_reduce of the
timings and the barriers all execute, so the N>1 code path can be exercised on a one-GPU box.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

VALU_FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz
VALU_FP64_PEAK_TFLOPS = 78.6
HBM_PEAK_GBS = 8000.0


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
    ap.add_argument("--strip_rows", type=int, default=0, help="rows per interleaved strip; 0 = 8 for N <= 2, 2 for N >= 4 (profiles/r01_strip_rows_sweep.txt)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    """Reference serial tracer on a bounded sample: the same scene, camera, spp and bounces
    on a 1/16-area frame (W/4 x H/4), 1 thread.  ~10-30 s of CPU work."""
    W, H = max(args.width // 4, 16), max(args.height // 4, 9)
    S, B = args.samples, args.bounces
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_serial_driver")
    sample = "scene %d, %dx%d (the %dx%d view at 1/4 linear resolution), %d spp, depth %d, serial fp64" % (
        args.scene_id, W, H, args.width, args.height, S, B)
    rays = W * H * S
    if os.path.exists(drv):
        t0 = time.perf_counter()
        r = subprocess.run([drv, str(args.scene_id), str(W), str(H), str(S), str(B)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        if r.returncode == 0 and r.stdout.startswith(b"P3"):
            return dict({"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference",
                         "sample": sample + "; reference src/InOneWeekend headers built by oracle/Makefile", "seconds": dt}, **host_cpu())
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


def pmc_traffic(args):
    """HBM bytes of the main render launch from the committed rocprofv3 --pmc passes (profiles/traffic.json):
    (total with the gfx950 x2 FETCH correction, fetch bytes raw, write bytes), or (None, None, None)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, None, None
    key = "s%d_%dx%d_%dspp_%db_f%d" % (args.scene_id, args.width, args.height, args.samples, args.bounces, args.precision)
    e = json.load(open(path)).get(key, {})
    f, w = e.get("main_launch_FETCH_SIZE_KB"), e.get("main_launch_WRITE_SIZE_KB")
    return (e.get("hbm_bytes_main_launch", e.get("hbm_bytes_per_launch")), f * 1024.0 if f is not None else None, w * 1024.0 if w is not None else None)


def pmc_issue(args):
    """Vector-instruction issue of the main launch from the committed rocprofv3 --pmc passes (the newest
    profiles/r*_pmc_sq_final.json; headline configuration only): wave-instructions per launch and SIMD cycles per
    instruction.  A wave64 VALU instruction occupies a SIMD-32 for at least 2 cycles, so 2 / cycles-per-
    instruction is the fraction of the vector issue peak the launch reaches."""
    import glob
    if (args.scene_id, args.width, args.height, args.samples, args.bounces, args.precision, args.schedule, args.scene_source) != (3, 1920, 1080, 100, 50, 32, "sorted", "grid"):
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq_final.json")))
    if not files:
        return None
    d = json.load(open(files[-1])).get("counters", {}).get("derived_main")
    if not d:
        return None
    return {"valu_wave_insts_per_launch": d["valu_insts_per_launch"], "simd_cycles_per_valu_inst": round(d["simd_cycles_per_valu_inst"], 3),
            "valu_issue_frac": round(2.0 / d["simd_cycles_per_valu_inst"], 4), "source": os.path.relpath(files[-1], ROOT)}


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
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # RTIOW_BENCH_BACKEND=gloo is a REHEARSAL mode for boxes with fewer GPUs than ranks (ranks
    # share devices, the gather bounces through the host); the real multi-GPU run uses RCCL.
    backend = os.environ.get("RTIOW_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    if args.strip_rows <= 0:
        args.strip_rows = 8 if world <= 2 else 2
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
    r.set_scene_source({"grid": rt.SCENE_GRID, "lds": rt.SCENE_LDS, "scalar": rt.SCENE_SCALAR, "lds_exact": rt.SCENE_LDS_EXACT}[args.scene_source])
    r.set_schedule({"sorted": rt.SCHED_SORTED, "persistent": rt.SCHED_PERSISTENT, "static": rt.SCHED_STATIC}[args.schedule])
    r.set_shard(rank, world, args.strip_rows)
    gather = StripGather(W, H, rank, world, args.strip_rows, tdtype, "cuda:%d" % device_index, stage_via_cpu=(backend != "nccl"),
                         always_collective=distributed)
    view = gather.local_view()
    r.bind_framebuffer(view.data_ptr(), view.numel() * view.element_size())
    r.init_rng(1227)                                   # untimed, like main.cu:326-330
    segments = r.count_segments(args.threads)          # untimed; also a first warm launch
    nspheres = r.stats()["num_spheres"]

    chain_main = int(r.stats()["max_chain_main"])      # this rank's longest per-pixel chain in the main launch
    trip_us, trip_segments = (None, 0) if args.no_scaling_probe else lone_ray_trip_us(rt, device_index, prec, scene, args)

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
        t = torch.tensor([elapsed, float(np.mean(kernel_ms)), float(segments), gather_ms or 0.0, floor_local, float(chain_main)], dtype=torch.float64, device="cuda")
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        per_rank = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(per_rank, t)
        elapsed, kernel_mean_max, segments_total = float(tmax[0]), float(tmax[1]), float(tsum[2])
        gather_ms_max, floor_ms, chain_max = float(tmax[3]), float(tmax[4]), int(tmax[5])
        kernel_ms_per_rank = [round(float(x[1]), 4) for x in per_rank]
        gather_ms_per_rank = [round(float(x[3]), 4) for x in per_rank]
    else:
        kernel_mean_max, segments_total = float(np.mean(kernel_ms)), float(segments)
        gather_ms_max, floor_ms, chain_max = None, floor_local, chain_main
        kernel_ms_per_rank, gather_ms_per_rank = [round(float(np.mean(kernel_ms)), 4)], None

    if rank == 0:
        rays = float(W) * H * S
        ms_per_step = elapsed / args.steps * 1e3
        value = rays / (ms_per_step * 1e-3) / 1e6
        # roofline of the dominant kernel (rank 0's launches; for N>1 this rank's shard)
        st = r.stats()
        my_rays = float(st["primary_rays"])
        my_segments = float(segments)
        per_seg = 23.0 * nspheres + 120.0
        flops_step = my_segments * per_seg + my_rays * 60.0
        kms = float(np.mean(kernel_ms))
        peak = VALU_FP32_PEAK_TFLOPS if prec == 32 else VALU_FP64_PEAK_TFLOPS
        rays_main = my_rays * (S - st["prepass_samples"]) / S
        flops = float(st["segments_main"]) * per_seg + rays_main * 60.0       # the main launch alone
        mms = float(np.mean(main_ms))
        achieved = flops / (mms * 1e-3) / 1e12
        # algorithmic HBM bytes of the main launch, per pixel: the framebuffer write (3 T) + its starting state:
        # the 48/64-byte hand-over record and a 4-byte order entry (sorted schedule) or the 24-byte RNG state
        my_pixels = my_rays / S
        state_bytes = ((48 if prec == 32 else 64) + 4) if st["phases"] == 2 else 24
        fb_bytes = my_pixels * (3 * (4 if prec == 32 else 8) + state_bytes)
        traffic, fetch_b, write_b = pmc_traffic(args) if world == 1 and args.schedule == "sorted" else (None, None, None)
        line = {
            "metric": "Mrays/s (= W x H x samples / render time)",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32" if prec == 32 else "f64", "data": "synthetic",
            "config": {"workload": "scene %d (%d spheres), %dx%d, %d spp, %d bounces, XORWOW seed 1227" % (args.scene_id, nspheres, W, H, S, B),
                       "scene_id": args.scene_id, "spheres": nspheres, "width": W, "height": H, "samples": S, "bounces": B,
                       "threads": args.threads, "scene_source": args.scene_source, "schedule": args.schedule,
                       "sharding": "interleaved %d-row strips, gather to rank 0 inside the step" % args.strip_rows if distributed else "none",
                       "backend": backend if distributed else None},
            "kernel_ms_mean": round(kms, 4), "kernel_ms_min": round(float(np.min(kernel_ms)), 4),
            "kernel_ms_mean_max_over_ranks": round(kernel_mean_max, 4),
            "segments_per_ray": round(segments_total / rays, 4),
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,   # the PMC passes: one GPU, full frame, default schedule
                         "fetch_bytes": fetch_b, "write_bytes": write_b,   # the same passes, raw FETCH_SIZE / WRITE_SIZE of the main launch (the sorted schedule stores pixels in cost order: writes exceed the 12 B/pixel framebuffer)
                         "kernel": "%s<%s>" % ("render_kernel" if args.schedule == "static" else "render_persistent_kernel", "float" if prec == 32 else "double"),
                         "launch_ms_mean": round(mms, 4), "launch_ms_min": round(float(np.min(main_ms)), 4),
                         "algorithmic_flops_per_launch": flops, "segments_in_launch": int(st["segments_main"]),
                         "samples_in_launch": int(S - st["prepass_samples"]),
                         "algorithmic_hbm_bytes_per_launch": fb_bytes,
                         "hbm_achieved_GBps": round(fb_bytes / (mms * 1e-3) / 1e9, 3), "hbm_peak_GBps": HBM_PEAK_GBS,
                         "achieved_is": "ALGORITHMIC flops per second: the reference's own sphere loop, 23 flop x every sphere x every segment (SURVEY.md 8d). "
                                        "The grid walk finds the same hits testing a few spheres per segment, so this figure measures useful work against the "
                                        "reference's algorithm, not executed arithmetic, and can exceed the peak; 'issued' is what the hardware executed",
                         "issued": pmc_issue(args) if world == 1 else None},
            "step": {"launches": "prepass (%d spp) + cost sort + main" % st["prepass_samples"] if st["phases"] == 2 else "main",
                     "kernel_ms_mean": round(kms, 4), "prepass_ms": round(float(st["prepass_ms"]), 4),
                     "solo_waves": int(st["solo_waves"]),   # > 0: a partly filled GPU (shard, small frame), render_solo_kernel
                     "algorithmic_flops": flops_step, "achieved_TFLOPs": round(flops_step / (kms * 1e-3) / 1e12, 3),
                     "frac_of_peak": round(flops_step / (kms * 1e-3) / 1e12 / peak, 4)},
        }
        line["scaling_detail"] = {
            "floor_ms": round(floor_ms, 4), "floor_is": "prepass_ms + longest per-pixel chain of the main launch x the trip latency of a lone ray on an idle GPU (1-pixel probe), max over ranks: "
                                                                "what a rank reaches if its longest chain runs undisturbed from the first trip; shards run it at 1.6-2x that "
                                                                "latency, two heavy pixels per solo wave beside the loaded SIMDs (DESIGN.md sections 4.3, 5)",
            "longest_chain_segments": chain_max, "lone_ray_trip_us": round(trip_us, 4) if trip_us else None,
            "lone_ray_probe": "1x1 frame, 400 spp, %d segments" % trip_segments,
            "kernel_ms_per_rank": kernel_ms_per_rank, "gather_ms_per_rank": gather_ms_per_rank,
            "gather_ms": round(gather_ms_max, 4) if gather_ms_max is not None else None,
            "gather_bytes_total": int(W) * H * 3 * (4 if prec == 32 else 8) if distributed else 0}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
            line["cpu_baseline"]["config1"] = cpu_baseline_config1()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    r.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
