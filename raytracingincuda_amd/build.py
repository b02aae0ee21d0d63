"""Build every native artefact of the package in-tree.

  lib/librtiow_hip.so    HIP kernels + device C-ABI (include/rtiow.h), hipcc --offload-arch=gfx950
  lib/librtiow_host.so   host-side scene/camera/PPM C interface (include/rtiow_host.h), g++
  bin/global-float-hip-raytrace, bin/global-double-hip-raytrace   drop-in executables
  bin/ppm_diff, bin/scaled_ppm_diff, bin/csv_avg                  harness tools (when present)

hipcc cross-compiles for gfx950 without a GPU.  Run as `python -m raytracingincuda_amd.build`.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "lib")
BIN = os.path.join(PKG, "bin")

# Floating-point contract of the render path: no implicit contraction, IEEE
# correctly rounded fp32 divide/sqrt, denormals kept.
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
    "-fno-fast-math", "-I" + INC,
]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-I" + INC]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build librtiow_hip.so for gfx950)")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, verbose):
    if verbose:
        print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


# ---- device-assembly post-pass for the render kernels -----------------------------------------
# On gfx950 `v_cndmask_b32` in its 32-bit (VOP2) encoding, which reads VCC implicitly, issues at
# ~18 cycles per wave-instruction whenever VCC does not come straight from the preceding compare
# (the 2nd/3rd select of one comparison, masks built by scalar instructions); the 64-bit (VOP3)
# encoding of the SAME instruction with VCC as an explicit operand issues at the normal 5 cycles
# (bin/valu_cost: v_cndmask_b32 18.1, v_cndmask_e64_vcc 5.6, cmp_then_3_cndmask 6.8 per
# instruction at 5 waves per SIMD; profiles/r02_valu_cost_cndmask.json).  The compiler picks
# the short encoding, so the device assembly is re-encoded before it is assembled: same
# instructions, same operands, 4 bytes longer each; headline frame 12.92 -> 12.78 ms
# (profiles/r02_ab_cndmask_e64.jsonl), images bit-identical (the GPU suite runs on this build).
_CNDMASK_E32 = r"v_cndmask_b32_e32 (v[0-9]+), ([^,]+), (v[0-9]+), vcc$"


def _llvm_tool(name):
    for d in (os.environ.get("RTIOW_LLVM_BIN"), "/opt/rocm/lib/llvm/bin", "/opt/rocm/llvm/bin"):
        if d and os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    raise RuntimeError(name + " not found")


def _build_hip_so_reencoded(out, srcs, extra, verbose):
    """librtiow_hip.so with the render kernels' device code taken through assembly text."""
    import re
    import tempfile
    main_src, other_srcs = srcs[0], srcs[1:]
    compile_flags = [f for f in HIP_FLAGS if f != "-shared"] + list(extra)
    with tempfile.TemporaryDirectory(prefix="rtiow_build_") as tmp:
        asm, asm2 = os.path.join(tmp, "dev.s"), os.path.join(tmp, "dev_e64.s")
        _run([_hipcc()] + compile_flags + ["--cuda-device-only", "-S", "-o", asm, main_src], verbose)
        text = open(asm).read()
        text, n = re.subn(_CNDMASK_E32, r"v_cndmask_b32_e64 \1, \2, \3, vcc", text, flags=re.M)
        if n == 0 or re.search(r"v_cndmask_b32_e32 .*vcc$", text, flags=re.M):
            raise RuntimeError("cndmask re-encoding did not cover the device assembly (%d rewritten)" % n)
        open(asm2, "w").write(text)
        obj, hsaco, fatbin = os.path.join(tmp, "dev.o"), os.path.join(tmp, "dev.hsaco"), os.path.join(tmp, "dev.hipfb")
        _run([_llvm_tool("clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm2, "-o", obj], verbose)
        _run([_llvm_tool("ld.lld"), "-shared", obj, "-o", hsaco], verbose)
        _run([_llvm_tool("clang-offload-bundler"), "-type=o", "-bundle-align=4096",
              "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
              "-input=/dev/null", "-input=" + hsaco, "-output=" + fatbin], verbose)
        objs = [os.path.join(tmp, "host.o")]
        _run([_hipcc()] + compile_flags + ["-Wno-unused-command-line-argument", "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary",
                                            "-Xclang", fatbin, "-c", main_src, "-o", objs[0]], verbose)
        for i, src in enumerate(other_srcs):
            objs.append(os.path.join(tmp, "other%d.o" % i))
            _run([_hipcc()] + compile_flags + ["-c", src, "-o", objs[-1]], verbose)
        _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"], verbose)
    if verbose:
        print("re-encoded %d v_cndmask_b32 (VOP2 -> VOP3) in the device code of %s" % (n, os.path.basename(out)), flush=True)


def build_hip_so(out, srcs, extra=(), verbose=True):
    """The HIP library: through the assembly post-pass above; if any step of that pipeline is not
    available, the plain one-step hipcc build of the same sources (same kernels, VOP2 selects)."""
    if os.environ.get("RTIOW_PLAIN_HIPCC_BUILD") != "1":
        try:
            _build_hip_so_reencoded(out, srcs, extra, verbose)
            return
        except Exception as e:      # noqa: BLE001 -- any tool failure: say so and build directly
            print("raytracingincuda_amd.build: assembly post-pass failed (%s); building %s directly with hipcc"
                  % (e, os.path.basename(out)), file=sys.stderr, flush=True)
    _run([_hipcc()] + HIP_FLAGS + list(extra) + ["-o", out] + list(srcs) + ["-ldl"], verbose)


def build_stats(verbose=True):
    """Optional: the HIP library with the execution-profile counters compiled in
    (lib/librtiow_hip_stats.so, -DRTIOW_PATH_STATS; used by scripts/path_stats_probe.py only)."""
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "librtiow_hip_stats.so")
    build_hip_so(out, [os.path.join(CSRC, "rtiow_hip.hip"), os.path.join(CSRC, "rtiow_group.hip")], ["-DRTIOW_PATH_STATS"], verbose)
    return out


def build(force=False, verbose=True):
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(BIN, exist_ok=True)
    headers = [os.path.join(INC, "rtiow.h"), os.path.join(INC, "rtiow_host.h")]
    me = os.path.abspath(__file__)

    hip_srcs = [os.path.join(CSRC, "rtiow_hip.hip"), os.path.join(CSRC, "rtiow_group.hip")]
    hip_so = os.path.join(LIB, "librtiow_hip.so")
    if force or _newer(hip_so, hip_srcs + [me, os.path.join(CSRC, "xorwow_jump67.inc")] + headers):
        # librccl is NOT linked: rtiow_group.hip dlopens it on first use (-ldl for old glibc)
        build_hip_so(hip_so, hip_srcs, (), verbose)

    host_src = os.path.join(CSRC, "host", "rtiow_host.cpp")
    host_so = os.path.join(LIB, "librtiow_host.so")
    if force or _newer(host_so, [host_src, me] + headers):
        _run(["g++"] + HOST_FLAGS + ["-shared", "-o", host_so, host_src], verbose)

    main_src = os.path.join(CSRC, "host", "main.cpp")
    for prec, name in ((32, "global-float-hip-raytrace"), (64, "global-double-hip-raytrace")):
        exe = os.path.join(BIN, name)
        if force or _newer(exe, [main_src, hip_so, host_so, me] + headers):
            _run(["g++"] + HOST_FLAGS + ["-DRTIOW_PRECISION=%d" % prec, "-o", exe, main_src,
                  "-L" + LIB, "-lrtiow_hip", "-lrtiow_host", "-Wl,-rpath,$ORIGIN/../lib",
                  "-Wl,-rpath-link," + LIB, "-Wl,-rpath-link,/opt/rocm/lib"], verbose)

    tools = os.path.join(CSRC, "tools")
    if os.path.isdir(tools):
        for src in sorted(os.listdir(tools)):
            if not src.endswith(".cpp"):
                continue
            exe = os.path.join(BIN, src[:-4])
            path = os.path.join(tools, src)
            if force or _newer(exe, [path, me]):
                _run(["g++", "-O2", "-std=c++17", "-o", exe, path], verbose)
    gpu_tools = os.path.join(CSRC, "tools_gpu")
    if os.path.isdir(gpu_tools):
        for src in sorted(os.listdir(gpu_tools)):
            if not src.endswith(".hip"):
                continue
            exe = os.path.join(BIN, src[:-4])
            path = os.path.join(gpu_tools, src)
            if force or _newer(exe, [path, me]):
                _run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-o", exe, path], verbose)
    return {"hip": hip_so, "host": host_so, "bin": BIN}


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--stats" in sys.argv:
        build_stats()
