#!/usr/bin/env python3
"""Dynamic instruction mix of ONE trip of a lone ray (the critical-path audit of the chain-latency regime).

A 1 x 1 frame is rendered with S1 and with S2 samples under `rocprofv3 --pmc`; every other wave of the launch only runs the kernel's prologue in
both, so the DIFFERENCE of the launch's counters divided by the difference of its path segments is what one trip of the lone lane executes:
vector / scalar / LDS / scalar-memory / branch instructions per trip.  With the trip's measured time (scripts/lone_trip_audit.py: HIP events at the
effective clock the wave itself stamped) that gives cycles per instruction on the critical path -- and the floor: a lone wave issues one instruction
per ~4 cycles (MI355X_MICROARCH.md: v_fma_f32 one wave alone 4 cycles, dependent ~4-6.6), so a trip cannot be shorter than ~4 x its instructions.
Usage: lone_trip_counters.py [scene_id [precision]]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pmc_passes as pp

scene = int(sys.argv[1]) if len(sys.argv) > 1 else 3
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 32
SETS = {"sq": ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_THREAD_CYCLES_VALU"],
        "mem": ["SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU"]}

def segments(S):
    import raytracingincuda_amd as rt
    with rt.Renderer(0, prec) as r:
        r.set_camera(rt.camera(prec, 1, 1, S, 50)); r.set_scene(rt.build_scene(scene, prec)); r.init_rng(1227)
        return r.count_segments(0)

def counters(S):
    cfg = {"scene_id": scene, "width": 1, "height": 1, "samples": S, "bounces": 50, "precision": prec, "schedule": "sorted", "scene_source": "grid", "threads": 0}
    out = {}
    for name, ctrs in SETS.items():
        means, _, _, _ = pp.run_pass(cfg, ctrs, 2, 240)
        out.update(means.get("main", {}))
    return out

S1, S2 = 100, 900
c1, c2 = counters(S1), counters(S2)          # the profiled children first: this process has not touched the GPU yet
n1, n2 = segments(S1), segments(S2)
d = {k: (c2[k] - c1[k]) / (n2 - n1) for k in c1 if k in c2}
out = {"config": "scene %d, 1x1 frame, fp%d: counters(%d spp) - counters(%d spp) over %d - %d path segments" % (scene, prec, S2, S1, n2, n1),
       "per_trip": {"vector_insts": round(d["SQ_INSTS_VALU"], 1), "scalar_insts": round(d["SQ_INSTS_SALU"], 1), "lds_insts": round(d["SQ_INSTS_LDS"], 1),
                    "scalar_memory_insts": round(d["SQ_INSTS_SMEM"], 1), "branch_insts": round(d["SQ_INSTS_BRANCH"], 1),
                    "vector_memory_insts": round(d.get("SQ_INSTS_VMEM_RD", 0) + d.get("SQ_INSTS_VMEM_WR", 0), 2),
                    "lanes_per_vector_inst": round(d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"], 1) if d.get("SQ_ACTIVE_INST_VALU") else None,
                    "wave_cycles_quad": round(d["SQ_WAVE_CYCLES"], 1), "wait_any_quad": round(d["SQ_WAIT_ANY"], 1), "wait_inst_any_quad": round(d["SQ_WAIT_INST_ANY"], 1),
                    "active_valu_quad": round(d["SQ_ACTIVE_INST_VALU"], 1), "active_scalar_quad": round(d["SQ_ACTIVE_INST_SCA"], 1), "wait_lds_quad": round(d.get("SQ_WAIT_INST_LDS", 0), 1)},
       "note": "wave_cycles etc. are in quad-cycles (x 4 = cycles) and include the cycles of the lone wave only per extra trip (the other waves have exited)"}
t = out["per_trip"]
t["all_insts"] = round(t["vector_insts"] + t["scalar_insts"] + t["lds_insts"] + t["scalar_memory_insts"] + t["branch_insts"] + t["vector_memory_insts"], 1)
print(json.dumps(out, indent=1))
