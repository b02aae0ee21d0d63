#!/usr/bin/env python3
"""SCALE_rNN.json (the driver's N = 1, 2, 4, 8 runs of bench.py, one contract line each) -> a table of ms per frame, speed-up,
efficiency, and the same against each run's own `scaling_detail.floor_ms` (prepass + longest per-pixel chain x lone-trip latency: what
strong scaling of ONE frame cannot go below).  Accepts the driver's JSON (a list / dict holding the parsed lines
under "parsed" or "runs"), or a file of raw contract lines.

    python3 scripts/scaling_table.py SCALE_r04.json [--markdown]
"""
import json
import sys


def contract_lines(obj):
    """Every dict that looks like a bench.py contract line, anywhere inside `obj`."""
    out = []
    if isinstance(obj, dict):
        if "n_gpus" in obj and "ms_per_step" in obj and "value" in obj:
            out.append(obj)
        for v in obj.values():
            out += contract_lines(v)
    elif isinstance(obj, list):
        for v in obj:
            out += contract_lines(v)
    elif isinstance(obj, str) and obj.lstrip().startswith("{") and '"n_gpus"' in obj:
        try:
            out += contract_lines(json.loads(obj))
        except ValueError:
            pass
    return out


def load(path):
    text = open(path).read()
    try:
        return contract_lines(json.loads(text))
    except ValueError:
        return contract_lines([l for l in text.splitlines() if l.strip()])


def table(lines):
    by_n = {}
    for l in lines:
        by_n[int(l["n_gpus"])] = l                      # the last line per N wins
    if 1 not in by_n:
        raise SystemExit("no N = 1 line: speed-up has no base")
    t1 = float(by_n[1]["ms_per_step"])
    rows = []
    for n in sorted(by_n):
        l = by_n[n]
        ms = float(l["ms_per_step"])
        sd = l.get("scaling_detail") or {}
        floor = sd.get("floor_ms")
        rows.append({"n_gpus": n, "ms_per_step": round(ms, 3), "value_Mrays_s": l["value"], "speedup": round(t1 / ms, 3), "efficiency": round(t1 / ms / n, 3),
                     "floor_ms": floor, "ms_over_floor": round(ms / floor, 2) if floor else None,
                     "speedup_limit_by_floor": round(t1 / floor, 2) if floor else None,
                     "kernel_ms_per_rank": sd.get("kernel_ms_per_rank"), "gather_ms": sd.get("gather_ms"), "gather_transport": sd.get("gather_transport")})
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if not args:
        raise SystemExit(__doc__)
    rows = table(load(args[0]))
    if "--markdown" in sys.argv:
        print("| GPUs | ms per frame | Mrays/s | speed-up | efficiency | floor ms | ms / floor | slowest rank's kernel ms | exchange ms | transport |")
        print("|---|---|---|---|---|---|---|---|---|---|")
        for r in rows:
            k = r["kernel_ms_per_rank"]
            print("| %d | %.3f | %.0f | %.2f | %.2f | %s | %s | %s | %s | %s |" % (
                r["n_gpus"], r["ms_per_step"], r["value_Mrays_s"], r["speedup"], r["efficiency"], r["floor_ms"], r["ms_over_floor"],
                ("%.3f" % max(k)) if k else "-", r["gather_ms"], r["gather_transport"] or "-"))
    else:
        for r in rows:
            print(json.dumps(r))


if __name__ == "__main__":
    main()
