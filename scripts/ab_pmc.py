#!/usr/bin/env python3
"""Hardware counters of the MAIN launch for two or more builds of librtiow_hip.so (lib/ab/*.so, build.build_variant):
one `rocprofv3 --pmc` child per library and counter set over scripts/one_render.py (the program itself after `--`).
Counts do not depend on the box or its clock; times do -- for those use scripts/ab_libs.py on the same box.

    python3 scripts/ab_pmc.py libA.so libB.so [--sets sq,lds] [-- --scene 1 --w 1920 ...]   (one_render.py arguments)
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_passes as pp

SETS = {
    "sq": pp.PASSES["sq"],
    "lds": ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_SENDMSG", "SQ_BUSY_CYCLES"],
}

def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--"); extra = args[k + 1:]; args = args[:k]
    sets = ["sq"]
    if "--sets" in args:
        k = args.index("--sets"); sets = args[k + 1].split(","); del args[k:k + 2]
    cfg = {"scene_id": 3, "width": 1920, "height": 1080, "samples": 100, "bounces": 50, "precision": 32}
    names = {"--scene": "scene_id", "--w": "width", "--h": "height", "--s": "samples", "--b": "bounces", "--prec": "precision"}
    for k in range(0, len(extra), 2):
        cfg[names[extra[k]]] = int(extra[k + 1])
    for lib in args:
        os.environ["RTIOW_HIP_LIBRARY"] = os.path.abspath(lib)
        rec = {"lib": os.path.basename(lib), "config": cfg}
        for s in sets:
            means, counts, build_id, ms = pp.run_pass(cfg, SETS[s], 2, 420)
            for cls in ("main", "prepass"):
                rec.setdefault(cls, {}).update({k: round(v) for k, v in means.get(cls, {}).items()})
            rec["build_id"] = build_id
            if s == "sq":
                rec["profiled_render_ms"] = ms
        m = rec["main"]
        if m.get("SQ_ACTIVE_INST_VALU"):
            rec["main_lanes_per_valu_inst"] = round(m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"], 2)
        if m.get("SQ_WAVE_CYCLES"):
            wc = m["SQ_WAVE_CYCLES"]
            rec["main_wave_cycle_split"] = {"wait_any": round(m.get("SQ_WAIT_ANY", 0) / wc, 3), "wait_inst_any": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                                            "active_valu": round(m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3), "active_sca": round(m.get("SQ_ACTIVE_INST_SCA", 0) / wc, 3)}
        print(json.dumps(rec), flush=True)

if __name__ == "__main__":
    main()
