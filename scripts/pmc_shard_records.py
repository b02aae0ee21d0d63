#!/usr/bin/env python3
"""Counter records of every rank's shard of the headline frame for N = 2 / 4 / 8, taken on ONE GPU.

`bench.py --gpus N` prints roofline.frac on its N > 1 line from these: rank k's main launch executes the same vector
instructions whichever device renders it (the hand-out inside a launch is dynamic, so SQ_INSTS_VALU of a shard varies
by a few tenths of a percent between runs, not with the device), and its launch time is measured live on its own device.
One "sq" pass per shard (14 shards, ~4 s each), keyed s3_1920x1080_100spp_50b_f32_r<k>of<N>x<strip_rows>, strip rows as
bench.py picks them (8 for N <= 2, 2 above).

    python3 scripts/pmc_shard_records.py --out profiles/pmc_records.json [--ns 2,4,8] [--precision 32] [bench-style config flags]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_passes as pp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--ns", default="2,4,8")
    ap.add_argument("--scene_id", type=int, default=3); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--samples", type=int, default=100); ap.add_argument("--bounces", type=int, default=50); ap.add_argument("--precision", type=int, default=32)
    a = ap.parse_args()
    data = json.load(open(a.out)) if os.path.exists(a.out) else {}
    for n in (int(x) for x in a.ns.split(",")):
        strip = 8 if n <= 2 else 2
        for rank in range(n):
            cfg = {"scene_id": a.scene_id, "width": a.width, "height": a.height, "samples": a.samples, "bounces": a.bounces, "precision": a.precision,
                   "schedule": "sorted", "scene_source": "grid", "threads": 0, "shard": "%d,%d,%d" % (rank, n, strip)}
            rec = pp.collect(cfg, passes=("sq",) if a.precision == 32 else ("sq", "f64"), reps=2, timeout=300)
            data[rec["key"]] = rec
            json.dump(data, open(a.out, "w"), indent=1, sort_keys=True)
            print(json.dumps({"key": rec["key"], "build_id": rec["build_id"][:12], "SQ_INSTS_VALU": rec["counters"]["main"].get("SQ_INSTS_VALU"),
                              "profiled_render_ms": rec.get("profiled_render_ms")}), flush=True)


if __name__ == "__main__":
    main()
