"""ctypes mirror of the C-ABI (include/rtiow.h, include/rtiow_host.h).

The call sequence a user writes mirrors the phases of the reference's main()
(/root/reference/src/GlobalFloatCUDAInOneWeekend/main.cu:37-400):

    cam   = camera(32, width, height, samples, bounces)      # main.cu:100-124
    scene = build_scene(scene_id, 32)                        # main.cu:148-296
    r = Renderer(device=0, precision=32)                     # main.cu:81-92
    r.set_camera(cam); r.set_scene(scene)                    # main.cu:301-321
    r.init_rng(1227)                                         # main.cu:326-330
    ms  = r.render(threads=8)                                # main.cu:334-341
    rgb = r.read_framebuffer()                               # main.cu:373
    write_ppm(ppm_filename(32, scene_id, ...), rgb)          # main.cu:349-379

There is no CPU fallback: if librtiow_hip.so is missing or no GPU is visible the calls
raise.  (The CPU oracle lives in oracle/ and is test infrastructure only.)
"""
import ctypes
import os
import sys

import numpy as np

LAMBERTIAN, METAL, DIELECTRIC = 0, 1, 2
SCENE_LDS, SCENE_SCALAR, SCENE_LDS_EXACT, SCENE_GRID = 0, 1, 2, 3
SCHED_STATIC, SCHED_PERSISTENT, SCHED_SORTED = 0, 1, 2
GATHER_AUTO, GATHER_RCCL, GATHER_PEER, GATHER_HOST = 0, 1, 2, 3
GROUP_MAX_STATS = 16
ABI_VERSION = 6          # include/rtiow.h RTIOW_ABI_VERSION

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(_PKG, "lib")


class RtiowError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rtiow error %d: %s" % (code, message))
        self.code = code


class CameraF32(ctypes.Structure):
    _fields_ = [("img_width", ctypes.c_int32), ("img_height", ctypes.c_int32),
                ("samples_per_pixel", ctypes.c_int32), ("max_depth", ctypes.c_int32),
                ("pixel_samples_scale", ctypes.c_float),
                ("center", ctypes.c_float * 3), ("pixel00_loc", ctypes.c_float * 3),
                ("pixel_delta_u", ctypes.c_float * 3), ("pixel_delta_v", ctypes.c_float * 3),
                ("defocus_angle", ctypes.c_float),
                ("defocus_disk_u", ctypes.c_float * 3), ("defocus_disk_v", ctypes.c_float * 3)]


class CameraF64(ctypes.Structure):
    _fields_ = [("img_width", ctypes.c_int32), ("img_height", ctypes.c_int32),
                ("samples_per_pixel", ctypes.c_int32), ("max_depth", ctypes.c_int32),
                ("pixel_samples_scale", ctypes.c_double),
                ("center", ctypes.c_double * 3), ("pixel00_loc", ctypes.c_double * 3),
                ("pixel_delta_u", ctypes.c_double * 3), ("pixel_delta_v", ctypes.c_double * 3),
                ("defocus_angle", ctypes.c_double),
                ("defocus_disk_u", ctypes.c_double * 3), ("defocus_disk_v", ctypes.c_double * 3)]


class Stats(ctypes.Structure):
    _fields_ = [("rng_init_ms", ctypes.c_double), ("render_ms", ctypes.c_double),
                ("primary_rays", ctypes.c_uint64), ("local_rows", ctypes.c_int32),
                ("num_spheres", ctypes.c_int32), ("block_x", ctypes.c_int32), ("block_y", ctypes.c_int32),
                ("vgprs", ctypes.c_int32), ("sgprs", ctypes.c_int32), ("lds_bytes", ctypes.c_int32),
                ("scene_source", ctypes.c_int32),
                ("schedule", ctypes.c_int32), ("grid_blocks", ctypes.c_int32), ("phases", ctypes.c_int32),
                ("prepass_samples", ctypes.c_int32), ("prepass_ms", ctypes.c_double), ("main_ms", ctypes.c_double),
                ("segments_prepass", ctypes.c_uint64), ("segments_main", ctypes.c_uint64),
                ("max_chain_prepass", ctypes.c_uint64), ("max_chain_main", ctypes.c_uint64),
                ("grid_nx", ctypes.c_int32), ("grid_nz", ctypes.c_int32), ("grid_registered", ctypes.c_int32),
                ("grid_direct", ctypes.c_int32), ("grid_cell", ctypes.c_double),
                ("solo_waves", ctypes.c_int32), ("solo_lanes", ctypes.c_int32), ("scene_prepare_ms", ctypes.c_double),
                ("staged_stores", ctypes.c_int32), ("num_cus", ctypes.c_int32), ("place_ms", ctypes.c_double),
                ("clock_mhz", ctypes.c_int32), ("reserved0", ctypes.c_int32),
                ("main_clock_mhz", ctypes.c_double), ("prepass_clock_mhz", ctypes.c_double), ("main_wave0_ms", ctypes.c_double)]


class GroupStats(ctypes.Structure):
    _fields_ = [("ngpus", ctypes.c_int32), ("strip_rows", ctypes.c_int32), ("gather_mode", ctypes.c_int32),
                ("rccl_version", ctypes.c_int32), ("kernel_ms", ctypes.c_double * GROUP_MAX_STATS),
                ("kernel_ms_max", ctypes.c_double), ("gather_ms", ctypes.c_double), ("gather_bytes", ctypes.c_uint64),
                ("create_ms", ctypes.c_double), ("peer_links", ctypes.c_int32), ("reserved", ctypes.c_int32)]


# Every symbol include/rtiow.h declares (tests check that the built library exports them all).
HIP_SYMBOLS = [
    "rtiow_abi_version", "rtiow_build_id", "rtiow_create", "rtiow_destroy", "rtiow_last_error_string", "rtiow_set_stream",
    "rtiow_set_scene", "rtiow_set_camera", "rtiow_set_shard", "rtiow_local_rows", "rtiow_local_row_map",
    "rtiow_init_rng", "rtiow_render", "rtiow_count_segments", "rtiow_bind_framebuffer", "rtiow_framebuffer_device_ptr",
    "rtiow_read_framebuffer", "rtiow_read_levels", "rtiow_set_scene_source", "rtiow_set_schedule", "rtiow_get_stats", "rtiow_synchronize",
    "rtiow_render_async", "rtiow_render_wait", "rtiow_stream", "rtiow_device",
    "rtiow_group_create", "rtiow_group_create_error", "rtiow_group_destroy", "rtiow_group_last_error_string", "rtiow_group_size", "rtiow_group_member",
    "rtiow_group_set_scene", "rtiow_group_set_camera", "rtiow_group_set_scene_source", "rtiow_group_set_schedule",
    "rtiow_group_init_rng", "rtiow_group_render", "rtiow_group_gather", "rtiow_group_framebuffer_device_ptr",
    "rtiow_group_read_framebuffer", "rtiow_group_get_stats", "rtiow_group_transport_note",
]
# include/rtiow_debug.h: exported by lib/librtiow_hip_debug.so (the test build) only
DEBUG_SYMBOLS = [
    "rtiow_debug_read_rng", "rtiow_debug_read_costs", "rtiow_debug_timeline", "rtiow_debug_pixel_times", "rtiow_debug_ops", "rtiow_debug_jump_matrices", "rtiow_debug_grid_plan", "rtiow_debug_hit_world",
    "rtiow_debug_gather_schedule",
]
HOST_SYMBOLS = [
    "rtiow_host_scene_slots", "rtiow_host_build_scene", "rtiow_host_camera", "rtiow_host_ppm_filename",
    "rtiow_host_write_ppm", "rtiow_host_format_ppm", "rtiow_host_write_ppm_binary", "rtiow_host_write_ppm_levels", "rtiow_host_levels", "rtiow_host_shard_rows", "rtiow_host_place_rows",
]

_hip = None
_hip_debug = None
_host = None


def lib_paths():
    # RTIOW_HIP_LIBRARY: another build of the same C-ABI (A/B timing of kernel variants, the stats build)
    # "hip_debug": the test build (the same kernels + the hooks of include/rtiow_debug.h); RTIOW_HIP_DEBUG_LIBRARY: another build of it
    return {"hip": os.environ.get("RTIOW_HIP_LIBRARY") or os.path.join(_LIBDIR, "librtiow_hip.so"),
            "hip_debug": os.environ.get("RTIOW_HIP_DEBUG_LIBRARY") or os.path.join(_LIBDIR, "librtiow_hip_debug.so"),
            "host": os.path.join(_LIBDIR, "librtiow_host.so")}


def load_host_library():
    global _host
    if _host is None:
        path = lib_paths()["host"]
        if not os.path.exists(path):
            raise RtiowError(-100, "%s not built; run `python -m raytracingincuda_amd.build`" % path)
        lib = ctypes.CDLL(path)
        vp, i32p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)
        lib.rtiow_host_scene_slots.argtypes = [ctypes.c_int]
        lib.rtiow_host_build_scene.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, i32p, i32p]
        lib.rtiow_host_camera.argtypes = [ctypes.c_int] * 5 + [vp]
        lib.rtiow_host_ppm_filename.argtypes = [ctypes.c_int] * 7 + [ctypes.c_char_p, ctypes.c_size_t]
        lib.rtiow_host_write_ppm.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
        lib.rtiow_host_write_ppm_binary.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
        lib.rtiow_host_format_ppm.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_char_p,
                                              ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        lib.rtiow_host_shard_rows.argtypes = [ctypes.c_int] * 4 + [i32p]
        lib.rtiow_host_place_rows.argtypes = [ctypes.c_int] * 6 + [vp, vp]
        _host = lib
    return _host


def load_hip_library(debug=False):
    """Load librtiow_hip.so -- or, debug=True, the test build librtiow_hip_debug.so, which also exports the hooks of
    include/rtiow_debug.h.  Raises loudly when it is missing: there is no fallback path."""
    global _hip, _hip_debug
    if (_hip_debug if debug else _hip) is None:
        path = lib_paths()["hip_debug" if debug else "hip"]
        if not os.path.exists(path):
            raise RtiowError(-100, "%s not built; run `python -m raytracingincuda_amd.build`" % path)
        lib = ctypes.CDLL(path)
        vp, H = ctypes.c_void_p, ctypes.c_void_p
        i32p = ctypes.POINTER(ctypes.c_int32)
        lib.rtiow_abi_version.argtypes = []
        lib.rtiow_build_id.argtypes = []
        lib.rtiow_build_id.restype = ctypes.c_char_p
        lib.rtiow_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(H)]
        lib.rtiow_destroy.argtypes = [H]
        lib.rtiow_last_error_string.argtypes = [H]
        lib.rtiow_last_error_string.restype = ctypes.c_char_p
        lib.rtiow_set_stream.argtypes = [H, vp]
        lib.rtiow_set_scene.argtypes = [H, ctypes.c_int, vp, vp, vp, i32p, i32p]
        lib.rtiow_set_camera.argtypes = [H, vp]
        lib.rtiow_set_shard.argtypes = [H, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.rtiow_local_rows.argtypes = [H, ctypes.POINTER(ctypes.c_int)]
        lib.rtiow_local_row_map.argtypes = [H, i32p]
        lib.rtiow_init_rng.argtypes = [H, ctypes.c_uint64]
        lib.rtiow_render.argtypes = [H, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        lib.rtiow_count_segments.argtypes = [H, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
        lib.rtiow_bind_framebuffer.argtypes = [H, vp, ctypes.c_size_t]
        lib.rtiow_framebuffer_device_ptr.argtypes = [H, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
        lib.rtiow_read_framebuffer.argtypes = [H, vp, ctypes.c_size_t]
        lib.rtiow_set_scene_source.argtypes = [H, ctypes.c_int]
        lib.rtiow_set_schedule.argtypes = [H, ctypes.c_int, ctypes.c_int]
        lib.rtiow_get_stats.argtypes = [H, ctypes.POINTER(Stats)]
        lib.rtiow_synchronize.argtypes = [H]
        if debug:
            lib.rtiow_debug_read_rng.argtypes = [H, ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t]
            lib.rtiow_debug_read_costs.argtypes = [H, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t]
            lib.rtiow_debug_timeline.argtypes = [H, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
            lib.rtiow_debug_ops.argtypes = [H, ctypes.c_int, ctypes.c_size_t, vp, vp, vp, vp]
            lib.rtiow_debug_gather_schedule.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                        ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
        lib.rtiow_render_async.argtypes = [H, ctypes.c_int]
        lib.rtiow_render_wait.argtypes = [H, ctypes.POINTER(ctypes.c_float)]
        lib.rtiow_stream.argtypes = [H, ctypes.POINTER(vp)]
        lib.rtiow_device.argtypes = [H, ctypes.POINTER(ctypes.c_int)]
        G = ctypes.c_void_p
        lib.rtiow_group_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(G)]
        lib.rtiow_group_create_error.argtypes = []
        lib.rtiow_group_create_error.restype = ctypes.c_char_p
        lib.rtiow_group_destroy.argtypes = [G]
        lib.rtiow_group_last_error_string.argtypes = [G]
        lib.rtiow_group_last_error_string.restype = ctypes.c_char_p
        lib.rtiow_group_transport_note.argtypes = [G]
        lib.rtiow_group_transport_note.restype = ctypes.c_char_p
        lib.rtiow_group_size.argtypes = [G]
        lib.rtiow_group_member.argtypes = [G, ctypes.c_int, ctypes.POINTER(H)]
        lib.rtiow_group_set_scene.argtypes = [G, ctypes.c_int, vp, vp, vp, i32p, i32p]
        lib.rtiow_group_set_camera.argtypes = [G, vp]
        lib.rtiow_group_set_scene_source.argtypes = [G, ctypes.c_int]
        lib.rtiow_group_set_schedule.argtypes = [G, ctypes.c_int, ctypes.c_int]
        lib.rtiow_group_init_rng.argtypes = [G, ctypes.c_uint64]
        lib.rtiow_group_render.argtypes = [G, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        lib.rtiow_group_gather.argtypes = [G]
        lib.rtiow_group_framebuffer_device_ptr.argtypes = [G, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
        lib.rtiow_group_read_framebuffer.argtypes = [G, vp, ctypes.c_size_t]
        lib.rtiow_group_get_stats.argtypes = [G, ctypes.POINTER(GroupStats)]
        if debug:
            _hip_debug = lib
        else:
            _hip = lib
    return _hip_debug if debug else _hip


def build_id():
    """rtiow_build_id() of the loaded HIP library: SHA-256 of the sources + flags it was built from."""
    return load_hip_library().rtiow_build_id().decode()


def _dtype(precision):
    if precision == 32:
        return np.float32
    if precision == 64:
        return np.float64
    raise ValueError("precision must be 32 or 64")


def scene_slots(scene_id):
    return load_host_library().rtiow_host_scene_slots(int(scene_id))


def build_scene(scene_id, precision=32):
    """World creation, main.cu:148-296.  Returns a dict of numpy arrays in slot order
    (center_radius [n,4], albedo_fuzz [n,4], refraction_index [n], type [n], valid [n])."""
    lib = load_host_library()
    dt = _dtype(precision)
    n = lib.rtiow_host_scene_slots(int(scene_id))
    cr = np.zeros((n, 4), dt); af = np.zeros((n, 4), dt); ri = np.zeros(n, dt)
    ty = np.zeros(n, np.int32); va = np.zeros(n, np.int32)
    got = lib.rtiow_host_build_scene(int(scene_id), precision, cr.ctypes.data, af.ctypes.data, ri.ctypes.data,
                                     ty.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                     va.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    if got != n:
        raise RtiowError(got, "rtiow_host_build_scene failed")
    return {"scene_id": int(scene_id), "precision": precision, "center_radius": cr, "albedo_fuzz": af,
            "refraction_index": ri, "type": ty, "valid": va}


def compact_scene(scene):
    """Drop the slots the reference leaves default-constructed (main.cu:168)."""
    keep = scene["valid"] != 0
    out = dict(scene)
    for k in ("center_radius", "albedo_fuzz", "refraction_index", "type", "valid"):
        out[k] = np.ascontiguousarray(scene[k][keep])
    return out


def camera(precision, width, height, samples, bounces):
    """camera configuration + camera::initialize (main.cu:100-124, camera.h:33-68)."""
    lib = load_host_library()
    cam = CameraF64() if precision == 64 else CameraF32()
    _dtype(precision)
    rc = lib.rtiow_host_camera(precision, int(width), int(height), int(samples), int(bounces), ctypes.addressof(cam))
    if rc:
        raise RtiowError(rc, "rtiow_host_camera: bad arguments")
    return cam


def ppm_filename(precision, scene_id, width, height, samples, bounces, threads):
    buf = ctypes.create_string_buffer(256)
    rc = load_host_library().rtiow_host_ppm_filename(precision, scene_id, width, height, samples, bounces, threads, buf, 256)
    if rc:
        raise RtiowError(rc, "rtiow_host_ppm_filename failed")
    return buf.value.decode()


def _rgb_precision(rgb):
    if rgb.dtype == np.float32:
        return 32
    if rgb.dtype == np.float64:
        return 64
    raise ValueError("rgb must be float32 or float64")


def format_ppm(rgb):
    """P3 text of an [H, W, 3] image, main.cu:368-379."""
    rgb = np.ascontiguousarray(rgb)
    h, w = rgb.shape[0], rgb.shape[1]
    lib = load_host_library()
    n = ctypes.c_size_t(0)
    p = _rgb_precision(rgb)
    rc = lib.rtiow_host_format_ppm(p, w, h, rgb.ctypes.data, None, 0, ctypes.byref(n))
    if rc:
        raise RtiowError(rc, "rtiow_host_format_ppm failed")
    buf = ctypes.create_string_buffer(n.value)
    rc = lib.rtiow_host_format_ppm(p, w, h, rgb.ctypes.data, buf, n.value, ctypes.byref(n))
    if rc:
        raise RtiowError(rc, "rtiow_host_format_ppm failed")
    return buf.raw[:n.value]


def write_ppm(path, rgb, binary=False):
    """P3 text file exactly as main.cu:368-379 writes it; binary=True writes the P6 twin (same levels)."""
    rgb = np.ascontiguousarray(rgb)
    lib = load_host_library()
    fn = lib.rtiow_host_write_ppm_binary if binary else lib.rtiow_host_write_ppm
    rc = fn(os.fsencode(path), _rgb_precision(rgb), rgb.shape[1], rgb.shape[0], rgb.ctypes.data)
    if rc:
        raise RtiowError(rc, "Could not open file for writing: %s" % path)


def levels(rgb):
    """(levels, nan_channels): main.cu:367, 374-376 per channel on the host, uint8 [H, W, 3] (what rtiow_read_levels computes on the device)."""
    rgb = np.ascontiguousarray(rgb)
    lib = load_host_library()
    lib.rtiow_host_levels.restype = ctypes.c_longlong
    out = np.empty(rgb.shape, np.uint8)
    n = lib.rtiow_host_levels(_rgb_precision(rgb), rgb.shape[1], rgb.shape[0], ctypes.c_void_p(rgb.ctypes.data), ctypes.c_void_p(out.ctypes.data))
    if n < 0:
        raise RtiowError(int(n), "rtiow_host_levels: bad arguments")
    return out, int(n)


def write_ppm_levels(path, lev, binary=False):
    """The P3 / P6 file from levels (uint8 [H, W, 3]): threaded formatter, every thread writes its own range of the file."""
    lev = np.ascontiguousarray(lev, np.uint8)
    rc = load_host_library().rtiow_host_write_ppm_levels(os.fsencode(path), lev.shape[1], lev.shape[0], ctypes.c_void_p(lev.ctypes.data), 1 if binary else 0)
    if rc:
        raise RtiowError(rc, "Could not open file for writing: %s" % path)


def shard_rows(height, rank, nranks, strip_rows=8):
    """Global row indices rendered by `rank` (same rule as rtiow_set_shard)."""
    lib = load_host_library()
    n = lib.rtiow_host_shard_rows(height, rank, nranks, strip_rows, None)
    if n < 0:
        raise RtiowError(n, "rtiow_host_shard_rows: bad arguments")
    rows = np.zeros(n, np.int32)
    if n:
        lib.rtiow_host_shard_rows(height, rank, nranks, strip_rows, rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return rows


def place_rows(full_rgb, local_rgb, rank, nranks, strip_rows):
    """Scatter one shard's rows (rtiow_set_shard order) into the full [H, W, 3] image."""
    assert full_rgb.flags.c_contiguous and local_rgb.flags.c_contiguous and full_rgb.dtype == local_rgb.dtype
    rc = load_host_library().rtiow_host_place_rows(_rgb_precision(full_rgb), full_rgb.shape[1], full_rgb.shape[0],
                                                   rank, nranks, strip_rows, local_rgb.ctypes.data, full_rgb.ctypes.data)
    if rc:
        raise RtiowError(rc, "rtiow_host_place_rows: bad arguments")
    return full_rgb


class Renderer:
    """One GPU's render state behind the C-ABI handle (include/rtiow.h)."""

    def __init__(self, device=0, precision=32, debug=False):
        """debug=True: the handle lives in the test build of the library (librtiow_hip_debug.so: the same kernels), whose hooks
        (include/rtiow_debug.h) the debug_* methods call; the product library exports none of them."""
        self._lib = load_hip_library(debug)
        self._debug = bool(debug)
        self.precision = precision
        self.dtype = _dtype(precision)
        self._borrowed = False
        self._h = ctypes.c_void_p()
        rc = self._lib.rtiow_create(int(device), int(precision), ctypes.byref(self._h))
        if rc:
            self._h = None
            raise RtiowError(rc, "rtiow_create(device=%d) failed -- is a GPU visible?" % device)
        self.width = self.height = 0

    @classmethod
    def _borrow(cls, lib, handle, precision, width, height):
        """A view of a handle somebody else owns (a member of a RendererGroup): close() leaves it alone."""
        r = cls.__new__(cls)
        r._lib, r.precision, r.dtype, r._borrowed = lib, precision, _dtype(precision), True
        r._debug = hasattr(lib, "rtiow_debug_ops") and lib is _hip_debug
        r._h = handle
        r.width, r.height = width, height
        return r

    def _check(self, rc):
        if rc:
            raise RtiowError(rc, self._lib.rtiow_last_error_string(self._h).decode(errors="replace"))

    def close(self):
        if getattr(self, "_h", None):
            if not self._borrowed:
                self._lib.rtiow_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, hip_stream):
        self._check(self._lib.rtiow_set_stream(self._h, ctypes.c_void_p(int(hip_stream))))

    def set_scene(self, scene):
        dt = self.dtype
        cr = np.ascontiguousarray(scene["center_radius"], dt)
        af = np.ascontiguousarray(scene["albedo_fuzz"], dt)
        ri = np.ascontiguousarray(scene["refraction_index"], dt)
        ty = np.ascontiguousarray(scene["type"], np.int32)
        va = np.ascontiguousarray(scene["valid"], np.int32) if scene.get("valid") is not None else None
        i32p = ctypes.POINTER(ctypes.c_int32)
        self._check(self._lib.rtiow_set_scene(self._h, len(ty), cr.ctypes.data, af.ctypes.data, ri.ctypes.data,
                                              ty.ctypes.data_as(i32p), va.ctypes.data_as(i32p) if va is not None else None))

    def set_camera(self, cam):
        want = CameraF64 if self.precision == 64 else CameraF32
        if not isinstance(cam, want):
            raise TypeError("camera precision does not match the renderer")
        self._check(self._lib.rtiow_set_camera(self._h, ctypes.addressof(cam)))
        self.width, self.height = cam.img_width, cam.img_height

    def set_shard(self, rank, nranks, strip_rows=8):
        self._check(self._lib.rtiow_set_shard(self._h, rank, nranks, strip_rows))

    def set_scene_source(self, source):
        self._check(self._lib.rtiow_set_scene_source(self._h, source))

    def debug_hit_world(self, rays):
        """hit_world alone on rays [n, 6] = {origin, direction}: (t [n], sphere index [n])."""
        self._need_debug()
        dt = _dtype(self.precision)
        rays = np.ascontiguousarray(rays, dt)
        n = rays.shape[0]
        t = np.zeros(n, dt); idx = np.zeros(n, np.int32)
        self._lib.rtiow_debug_hit_world.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self._check(self._lib.rtiow_debug_hit_world(self._h, n, rays.ctypes.data, t.ctypes.data, idx.ctypes.data))
        return t, idx

    def set_schedule(self, schedule, waves_per_simd=0):
        self._check(self._lib.rtiow_set_schedule(self._h, schedule, waves_per_simd))

    @property
    def local_rows(self):
        n = ctypes.c_int(0)
        self._check(self._lib.rtiow_local_rows(self._h, ctypes.byref(n)))
        return n.value

    def local_row_map(self):
        rows = np.zeros(self.local_rows, np.int32)
        if len(rows):
            self._check(self._lib.rtiow_local_row_map(self._h, rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
        return rows

    def init_rng(self, seed=1227):
        self._check(self._lib.rtiow_init_rng(self._h, ctypes.c_uint64(seed)))

    def render(self, threads=8, sync=True):
        """render<<<>>> + sync; returns the HIP-event kernel time in ms (None when sync=False)."""
        if sync:
            ms = ctypes.c_float(0)
            self._check(self._lib.rtiow_render(self._h, int(threads), ctypes.byref(ms)))
            return ms.value
        self._check(self._lib.rtiow_render(self._h, int(threads), None))
        return None

    def count_segments(self, threads=8):
        """Untimed render that also counts path segments (hit_world calls)."""
        n = ctypes.c_uint64(0)
        self._check(self._lib.rtiow_count_segments(self._h, int(threads), ctypes.byref(n)))
        return n.value

    def bind_framebuffer(self, device_ptr, nbytes):
        self._check(self._lib.rtiow_bind_framebuffer(self._h, ctypes.c_void_p(int(device_ptr)), nbytes))

    def framebuffer_device_ptr(self):
        p = ctypes.c_void_p()
        n = ctypes.c_size_t(0)
        self._check(self._lib.rtiow_framebuffer_device_ptr(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def read_framebuffer(self):
        out = np.empty((self.local_rows, self.width, 3), self.dtype)
        if out.size:
            self._check(self._lib.rtiow_read_framebuffer(self._h, out.ctypes.data, out.nbytes))
        return out

    def read_levels(self):
        """(levels uint8 [rows, W, 3], nan_channels): the writer's quantisation done on the device (rtiow_read_levels)."""
        out = np.empty((self.local_rows, self.width, 3), np.uint8)
        nans = ctypes.c_uint64(0)
        self._check(self._lib.rtiow_read_levels(self._h, ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(out.nbytes), ctypes.byref(nans)))
        return out, int(nans.value)

    def _need_debug(self):
        if not getattr(self, "_debug", False):
            raise RtiowError(-101, "test hook: librtiow_hip.so does not export it -- construct Renderer(device, precision, debug=True) (librtiow_hip_debug.so)")

    def synchronize(self):
        self._check(self._lib.rtiow_synchronize(self._h))

    def stats(self):
        st = Stats()
        self._check(self._lib.rtiow_get_stats(self._h, ctypes.byref(st)))
        return {name: getattr(st, name) for name, _ in Stats._fields_}

    # -- test hooks
    def debug_read_rng(self):
        self._need_debug()
        n = self.local_rows * self.width
        out = np.zeros((n, 6), np.uint32)
        self._check(self._lib.rtiow_debug_read_rng(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), out.size))
        return out

    def debug_read_costs(self):
        """(own, smoothed) prepass cost maps of the last sorted render, each local_rows x width."""
        self._need_debug()
        own = np.zeros((self.local_rows, self.width), np.uint32)
        smoothed = np.zeros_like(own)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        self._check(self._lib.rtiow_debug_read_costs(self._h, own.ctypes.data_as(u32p), smoothed.ctypes.data_as(u32p), own.size))
        return own, smoothed

    def debug_timeline(self, threads=0):
        self._need_debug()
        out = np.zeros((16384, 8), np.uint64)
        n = ctypes.c_int(0)
        self._check(self._lib.rtiow_debug_timeline(self._h, int(threads), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), out.size, ctypes.byref(n)))
        return out[: n.value]

    def debug_pixel_times(self, threads=0):
        """uint32 [rows, W, 4]: per pixel {taken, finished (100 MHz ticks), segments, wave} in the last launch that rendered it."""
        self._need_debug()
        out = np.zeros((self.local_rows, self.width, 4), np.uint32)
        self._check(self._lib.rtiow_debug_pixel_times(self._h, int(threads), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.c_size_t(out.size)))
        return out

    def debug_ops(self, op, a, b=None, c=None):
        self._need_debug()
        a = np.ascontiguousarray(a, self.dtype)
        b = np.ascontiguousarray(b if b is not None else a, self.dtype)
        c = np.ascontiguousarray(c if c is not None else a, self.dtype)
        out = np.empty_like(a)
        self._check(self._lib.rtiow_debug_ops(self._h, op, a.size, a.ctypes.data, b.ctypes.data, c.ctypes.data, out.ctypes.data))
        return out


def debug_gather_schedule(devices, rows, width, precision=32, mode=GATHER_RCCL, fail_at=-1, fallback=False):
    """The exchange's schedule (rtiow_group_gather's own) run against a recorder: (list of 8-tuples, return code).
    Host only -- no GPU, no RCCL.  Record layout: csrc/rtiow_group.hip, rtiow_debug_gather_schedule.
    fallback=True: with RTIOW_GATHER_AUTO's chain (a transport that fails at gather time -> drain -> the next one, from the top);
    the last record is then (12, -1, transport that carried the image, ...)."""
    lib = load_hip_library(debug=True)
    n = len(devices)
    cap = 3 * (16 * n + 16) + 8
    if fallback:
        mode = int(mode) | 0x100
    rec = (ctypes.c_int64 * (8 * cap))()
    rc = ctypes.c_int(0)
    got = lib.rtiow_debug_gather_schedule(n, (ctypes.c_int * n)(*devices), (ctypes.c_int * n)(*rows), int(width), int(precision), int(mode), int(fail_at),
                                          rec, cap, ctypes.byref(rc))
    if got < 0:
        raise RtiowError(got, "rtiow_debug_gather_schedule: bad arguments")
    return [tuple(rec[8 * k: 8 * k + 8]) for k in range(got)], rc.value


class RendererGroup:
    """Several GPUs of one node driven from this process (rtiow_group_*, include/rtiow.h): interleaved
    row strips, one exchange to device 0 after the render (RCCL, or peer copies), de-interleaved there.
    `devices` may repeat a device: the ranks then share it (how the N-rank logic is tested on one GPU).

    STDOUT: RCCL prints a version banner ("RCCL version : ...") on the process's stdout when its first communicator is created, i.e.
    inside this constructor's rtiow_group_create on distinct devices.  The library does not touch file descriptors; a
    caller whose stdout is data passes quiet_stdout=True, which points fd 1 at stderr for the duration of the constructor (process-wide:
    only for callers with no other thread writing to stdout), or does the same around the call itself (bench.py, the executables)."""

    def __init__(self, ngpus=1, precision=32, strip_rows=8, gather=GATHER_AUTO, devices=None, quiet_stdout=False):
        saved = -1
        if quiet_stdout:
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
        try:
            self._create(ngpus, precision, strip_rows, gather, devices)
        finally:
            if saved >= 0:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)

    def _create(self, ngpus, precision, strip_rows, gather, devices):
        self._lib = load_hip_library()
        self.precision, self.dtype, self.ngpus = precision, _dtype(precision), int(ngpus)
        self._g = ctypes.c_void_p()
        devs = (ctypes.c_int * self.ngpus)(*[int(d) for d in devices]) if devices is not None else None
        if devices is not None and len(devices) != self.ngpus:
            raise ValueError("devices must list one device per rank")
        rc = self._lib.rtiow_group_create(self.ngpus, devs, int(precision), int(strip_rows), int(gather), ctypes.byref(self._g))
        if rc:
            self._g = None
            why = self._lib.rtiow_group_create_error().decode(errors="replace")
            raise RtiowError(rc, "rtiow_group_create(%d GPUs) failed: %s" % (self.ngpus, why or "are that many GPUs visible?"))
        self.width = self.height = 0

    def _check(self, rc):
        if rc:
            raise RtiowError(rc, self._lib.rtiow_group_last_error_string(self._g).decode(errors="replace"))

    def close(self):
        if getattr(self, "_g", None):
            self._lib.rtiow_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_scene(self, scene):
        dt = self.dtype
        cr = np.ascontiguousarray(scene["center_radius"], dt)
        af = np.ascontiguousarray(scene["albedo_fuzz"], dt)
        ri = np.ascontiguousarray(scene["refraction_index"], dt)
        ty = np.ascontiguousarray(scene["type"], np.int32)
        va = np.ascontiguousarray(scene["valid"], np.int32) if scene.get("valid") is not None else None
        i32p = ctypes.POINTER(ctypes.c_int32)
        self._check(self._lib.rtiow_group_set_scene(self._g, len(ty), cr.ctypes.data, af.ctypes.data, ri.ctypes.data,
                                                    ty.ctypes.data_as(i32p), va.ctypes.data_as(i32p) if va is not None else None))

    def set_camera(self, cam):
        want = CameraF64 if self.precision == 64 else CameraF32
        if not isinstance(cam, want):
            raise TypeError("camera precision does not match the group")
        self._check(self._lib.rtiow_group_set_camera(self._g, ctypes.addressof(cam)))
        self.width, self.height = cam.img_width, cam.img_height

    def set_scene_source(self, source):
        self._check(self._lib.rtiow_group_set_scene_source(self._g, source))

    def set_schedule(self, schedule, waves_per_simd=0):
        self._check(self._lib.rtiow_group_set_schedule(self._g, schedule, waves_per_simd))

    def init_rng(self, seed=1227):
        self._check(self._lib.rtiow_group_init_rng(self._g, ctypes.c_uint64(seed)))

    def render(self, threads=8):
        """All devices render their strips; returns the slowest device's HIP-event kernel time (ms)."""
        ms = ctypes.c_float(0)
        self._check(self._lib.rtiow_group_render(self._g, int(threads), ctypes.byref(ms)))
        return ms.value

    def gather(self):
        self._check(self._lib.rtiow_group_gather(self._g))

    def member(self, rank):
        """Rank `rank`'s own handle as a borrowed Renderer (per-device stats, count_segments, knobs)."""
        h = ctypes.c_void_p()
        self._check(self._lib.rtiow_group_member(self._g, int(rank), ctypes.byref(h)))
        return Renderer._borrow(self._lib, h, self.precision, self.width, self.height)

    def read_framebuffer(self):
        """Exchange + de-interleave on device 0, then the full [H, W, 3] image."""
        out = np.empty((self.height, self.width, 3), self.dtype)
        self._check(self._lib.rtiow_group_read_framebuffer(self._g, out.ctypes.data, out.nbytes))
        return out

    def stats(self):
        st = GroupStats()
        self._check(self._lib.rtiow_group_get_stats(self._g, ctypes.byref(st)))
        d = {name: getattr(st, name) for name, _ in GroupStats._fields_ if name != "kernel_ms"}
        d["kernel_ms"] = [st.kernel_ms[k] for k in range(min(st.ngpus, GROUP_MAX_STATS))]
        d["transport_note"] = self._lib.rtiow_group_transport_note(self._g).decode(errors="replace")
        return d
