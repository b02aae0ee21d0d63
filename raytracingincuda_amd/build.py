"""Build every native artefact of the package in-tree.

  lib/librtiow_hip.so    HIP kernels + device C-ABI (include/rtiow.h), hipcc --offload-arch=gfx950
  lib/librtiow_hip_debug.so   the same sources with -DRTIOW_DEBUG_API: + the test hooks of include/rtiow_debug.h (tests/, study and profiling scripts only)
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
    "-fno-fast-math", "-fvisibility-inlines-hidden", "-I" + INC,
]
# the export list: rtiow_* only (csrc/librtiow_hip.map)
HIP_LINK = ["-Wl,--version-script=" + os.path.join(CSRC, "librtiow_hip.map"), "-ldl"]
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


def hip_includes():
    """The device / library headers rtiow_hip.hip is made of (csrc/device, csrc/library)."""
    out = []
    for sub in ("device", "library"):
        d = os.path.join(CSRC, sub)
        out += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".h", ".inc"))]
    return out


def hip_build_id(extra_flags=()):
    """SHA-256 over the HIP library's sources (in a fixed order), its public header and the compiler flags:
    what `rtiow_build_id()` of the built library returns.  PMC records under profiles/ carry the id of the build
    they were measured on; bench.py uses a record only when it matches the library that is loaded."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, "rtiow_hip.hip"), os.path.join(CSRC, "rtiow_group.hip")] + hip_includes() + [os.path.join(INC, "rtiow.h"), os.path.join(INC, "rtiow_debug.h")]
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    flags = [f for f in HIP_FLAGS if not f.startswith("-I")] + list(extra_flags)
    h.update(" ".join(flags).encode())
    h.update(open(os.path.join(CSRC, "librtiow_hip.map"), "rb").read())
    return h.hexdigest()


def _run(cmd, verbose):
    if verbose:
        print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build_stats(verbose=True):
    """Optional: the HIP library with the execution-profile counters compiled in
    (lib/librtiow_hip_stats.so, -DRTIOW_PATH_STATS; used by scripts/path_stats_probe.py only)."""
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "librtiow_hip_stats.so")
    _run([_hipcc()] + HIP_FLAGS + ["-DRTIOW_PATH_STATS", "-DRTIOW_DEBUG_API", '-DRTIOW_BUILD_ID="%s"' % hip_build_id(["-DRTIOW_PATH_STATS", "-DRTIOW_DEBUG_API"]), "-o", out, os.path.join(CSRC, "rtiow_hip.hip"), os.path.join(CSRC, "rtiow_group.hip")] + HIP_LINK, verbose)
    return out


def build_variant(name, defines=(), csrc=None, verbose=True):
    """A/B builds of the HIP library (lib/ab/<name>.so; scripts/ab_libs.py, scripts/pmc_passes.py via RTIOW_HIP_LIBRARY):
    the same flags as the product plus `defines`; `csrc` = another source tree (e.g. an older commit exported with git archive)."""
    out_dir = os.path.join(LIB, "ab")
    os.makedirs(out_dir, exist_ok=True)
    src = csrc or CSRC
    out = os.path.join(out_dir, name + ".so")
    flags = [f for f in HIP_FLAGS if not f.startswith("-I")] + ["-I" + (os.path.join(os.path.dirname(os.path.dirname(src)), "include") if csrc else INC)]
    _run([_hipcc()] + flags + list(defines) + ['-DRTIOW_BUILD_ID="variant:%s"' % name, "-o", out, os.path.join(src, "rtiow_hip.hip"), os.path.join(src, "rtiow_group.hip")] + HIP_LINK, verbose)
    return out


def build(force=False, verbose=True):
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(BIN, exist_ok=True)
    headers = [os.path.join(INC, "rtiow.h"), os.path.join(INC, "rtiow_host.h"), os.path.join(INC, "rtiow_debug.h")]
    me = os.path.abspath(__file__)

    hip_srcs = [os.path.join(CSRC, "rtiow_hip.hip"), os.path.join(CSRC, "rtiow_group.hip")]
    hip_so = os.path.join(LIB, "librtiow_hip.so")
    if force or _newer(hip_so, hip_srcs + hip_includes() + [me, os.path.join(CSRC, "librtiow_hip.map")] + headers):
        # librccl is NOT linked: rtiow_group.hip dlopens it on first use (-ldl for old glibc)
        _run([_hipcc()] + HIP_FLAGS + ['-DRTIOW_BUILD_ID="%s"' % hip_build_id(), "-o", hip_so] + hip_srcs + HIP_LINK, verbose)
    # the test build: the same kernels + the hooks of include/rtiow_debug.h (same flags, so the kernels are the product's; its build id names the extra define)
    dbg_so = os.path.join(LIB, "librtiow_hip_debug.so")
    if force or _newer(dbg_so, hip_srcs + hip_includes() + [me, os.path.join(CSRC, "librtiow_hip.map")] + headers):
        _run([_hipcc()] + HIP_FLAGS + ["-DRTIOW_DEBUG_API", '-DRTIOW_BUILD_ID="%s"' % hip_build_id(["-DRTIOW_DEBUG_API"]), "-o", dbg_so] + hip_srcs + HIP_LINK, verbose)

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
