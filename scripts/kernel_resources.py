"""Register / scratch / code-size figures of the render kernels for a set of -D defines, from the compiler's metadata
(no GPU: hipcc cross-compiles gfx950).  Usage: kernel_resources.py [-DNAME=VALUE ...] [--filter substring] [--keep out.s]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from raytracingincuda_amd import build as b

def metadata(defines=(), keep=None):
    out = keep or os.path.join(tempfile.mkdtemp(prefix="isa"), "rtiow_hip.s")
    flags = [f for f in b.HIP_FLAGS if f != "-shared"]
    subprocess.run([b._hipcc()] + flags + list(defines) + ["-S", "--cuda-device-only", "-o", out, os.path.join(b.CSRC, "rtiow_hip.hip")], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    pat = re.compile(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n"
                     r"(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)")
    found = list(pat.finditer(text))
    names = subprocess.run(["c++filt"], input="\n".join(m.group(1) for m in found), capture_output=True, text=True, check=True).stdout.splitlines()
    meta = {}
    for m, name in zip(found, names):
        meta[name] = {"scratch": int(m.group(2)), "sgpr": int(m.group(3)), "sgpr_spill": int(m.group(4)), "vgpr": int(m.group(5)), "vgpr_spill": int(m.group(6))}
    # static instruction counts per kernel body (lines between the symbol's label and its s_endpgm)
    for m, name in zip(found, names):
        sym = m.group(1)
        k = text.find("\n" + sym + ":")
        if k < 0: continue
        e = text.find(".Lfunc_end", k)
        body = text[k:e]
        ins = [l.strip() for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        meta[name]["insts"] = len(ins)
        meta[name]["valu"] = sum(1 for l in ins if l.startswith("v_"))
        meta[name]["salu"] = sum(1 for l in ins if l.startswith("s_"))
        meta[name]["lds"] = sum(1 for l in ins if l.startswith("ds_"))
    return meta

if __name__ == "__main__":
    args = sys.argv[1:]
    filt = "render_"
    keep = None
    if "--filter" in args:
        k = args.index("--filter"); filt = args[k + 1]; del args[k:k + 2]
    if "--keep" in args:
        k = args.index("--keep"); keep = args[k + 1]; del args[k:k + 2]
    for name, v in sorted(metadata(args, keep).items()):
        if filt in name:
            print(name, v)
