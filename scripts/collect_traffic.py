"""Turns two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of scripts/one_render.py into
profiles/traffic.json (HBM bytes per render launch), as MI355X_MICROARCH.md §HBM prescribes:
separate passes, KB units, and FETCH_SIZE doubled on gfx950 for wide coalesced reads (our reads
are dword gathers, so both the raw and the doubled figure are kept; the doubled one is an upper
bound).  Usage: collect_traffic.py <fetch_dir> <write_dir> <key> [out.json]"""
import csv, glob, json, os, sys

def mean_counter(d, name):
    """Bytes of ONE render = sum over its kernels (phase A, sort, phase B) of the mean per dispatch."""
    per_kernel = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if ("render" in k or "cost_" in k) and r["Counter_Name"] == name:
                per_kernel.setdefault(k, []).append(float(r["Counter_Value"]))      # the full name: one entry per kernel
    return sum(sum(v) / len(v) for v in per_kernel.values()) if per_kernel else None

def main_launch_counter(d, name):
    """Mean over dispatches of the MAIN launch (render_persistent_kernel / render_kernel) alone."""
    v = []
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if ("render_persistent_kernel" in k or "render_kernel" in k) and r["Counter_Name"] == name:
                v.append(float(r["Counter_Value"]))
    return sum(v) / len(v) if v else None

fetch_dir, write_dir, key = sys.argv[1:4]
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
fetch_kb, write_kb = mean_counter(fetch_dir, "FETCH_SIZE"), mean_counter(write_dir, "WRITE_SIZE")
data = json.load(open(out)) if os.path.exists(out) else {}
mf, mw = main_launch_counter(fetch_dir, "FETCH_SIZE"), main_launch_counter(write_dir, "WRITE_SIZE")
data[key] = {"FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
             "main_launch_FETCH_SIZE_KB": mf, "main_launch_WRITE_SIZE_KB": mw,
             "hbm_bytes_main_launch": (2.0 * mf + mw) * 1024.0 if mf is not None and mw is not None else None,
             "hbm_bytes_per_launch_raw": (fetch_kb + write_kb) * 1024.0,
             "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
             "note": "rocprofv3 --pmc, separate passes; FETCH_SIZE x2 (gfx950 correction, upper bound for non-streaming reads)"}
json.dump(data, open(out, "w"), indent=1)
print(json.dumps(data[key]))
