"""ctypes loader for the CPU oracle (oracle/liboracle.so) -- test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")


def ensure_built():
    src = os.path.join(ORACLE_DIR, "rtiow_oracle.cpp")
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", ORACLE_DIR, "liboracle.so"], check=True, stdout=subprocess.DEVNULL)
    return ORACLE_SO


RECURSIVE, ITERATIVE = 0, 1        # loop form (oracle/rtiow_oracle.cpp LoopForm)
SKY_CURRENT, SKY_PRIMARY = 0, 1    # sky mode


def p3_levels(p3):
    """P3 text -> uint8 [H, W, 3]."""
    tok = p3.split()
    W, H = int(tok[1]), int(tok[2])
    return np.array(tok[4:], np.int32).reshape(H, W, 3).astype(np.uint8)


def to_levels(img):
    """float image -> the P3 writer's levels int(256*clamp(c, 0, 0.999)) (main.cu:374-376)."""
    return np.floor(256.0 * np.clip(np.asarray(img, np.float64), 0.0, 0.999)).astype(np.uint8)


def diff_stats(a, b):
    """ppm_diff statistics between two uint8 images: mean |d|, p99, max, per-channel bias."""
    d = a.astype(np.int32) - b.astype(np.int32)
    ad = np.abs(d)
    return {"mean": float(ad.mean()), "p99": float(np.percentile(ad, 99)), "max": int(ad.max()),
            "bias": [float(d[..., c].mean()) for c in range(3)]}


def _dt(prec):
    return np.float32 if prec == 32 else np.float64


class Oracle:
    def __init__(self):
        L = ctypes.CDLL(ensure_built())
        vp = ctypes.c_void_p
        u64 = ctypes.c_ulonglong
        L.oracle_glibc_rand.argtypes = [ctypes.c_int, vp]
        L.oracle_build_scene.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp]
        L.oracle_camera_init.argtypes = [ctypes.c_int] * 5 + [vp, vp]
        L.oracle_xorwow_init.argtypes = [u64, u64, u64, ctypes.c_int, vp]
        L.oracle_xorwow_next.argtypes = [vp]
        L.oracle_xorwow_next.restype = ctypes.c_uint
        L.oracle_uniform_f32.argtypes = [vp]
        L.oracle_uniform_f32.restype = ctypes.c_float
        L.oracle_uniform_f64.argtypes = [vp]
        L.oracle_uniform_f64.restype = ctypes.c_double
        L.oracle_render.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp, u64, ctypes.c_int, ctypes.c_int, vp, vp]
        L.oracle_render_modes.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp, u64, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, vp, vp, vp]
        L.oracle_render_serial.argtypes = [ctypes.c_int] * 5 + [vp, ctypes.c_longlong, vp]
        L.oracle_render_serial.restype = ctypes.c_longlong
        L.oracle_render_serial_modes.argtypes = [ctypes.c_int] * 7 + [vp, ctypes.c_longlong, vp]
        L.oracle_render_serial_modes.restype = ctypes.c_longlong
        L.oracle_sky.argtypes = [ctypes.c_int, vp, vp]
        L.oracle_set_threads.argtypes = [ctypes.c_int]
        L.oracle_hit_sphere_f64.argtypes = [vp, ctypes.c_double, vp, vp, ctypes.c_double, ctypes.c_double, vp, vp, vp, vp]
        L.oracle_reflect_f64.argtypes = [vp, vp, vp]
        L.oracle_refract_f64.argtypes = [vp, vp, ctypes.c_double, vp]
        L.oracle_reflectance_f64.argtypes = [ctypes.c_double, ctypes.c_double]
        L.oracle_reflectance_f64.restype = ctypes.c_double
        self.L = L

    # -- host RNG / scene / camera
    def glibc_rand(self, n):
        out = np.zeros(n, np.int32)
        self.L.oracle_glibc_rand(n, out.ctypes.data)
        return out

    def build_scene(self, scene_id, prec):
        n = self.L.oracle_scene_slots(scene_id)
        dt = _dt(prec)
        cr = np.zeros((n, 4), dt); af = np.zeros((n, 4), dt); ri = np.zeros(n, dt)
        ty = np.zeros(n, np.int32); va = np.zeros(n, np.int32)
        got = self.L.oracle_build_scene(scene_id, prec, cr.ctypes.data, af.ctypes.data, ri.ctypes.data, ty.ctypes.data, va.ctypes.data)
        assert got == n
        return {"scene_id": scene_id, "precision": prec, "center_radius": cr, "albedo_fuzz": af,
                "refraction_index": ri, "type": ty, "valid": va}

    def camera_flat(self, prec, W, H, S, B):
        ints = np.zeros(4, np.int32)
        flat = np.zeros(20, _dt(prec))
        assert self.L.oracle_camera_init(prec, W, H, S, B, ints.ctypes.data, flat.ctypes.data) == 0
        return ints, flat

    # -- XORWOW
    def xorwow_init(self, seed, subsequence, offset=0, salt=0):
        st = np.zeros(6, np.uint32)
        self.L.oracle_xorwow_init(seed, int(subsequence), offset, salt, st.ctypes.data)
        return st

    def xorwow_next(self, st):
        return self.L.oracle_xorwow_next(st.ctypes.data)

    def uniform(self, prec, st):
        return self.L.oracle_uniform_f32(st.ctypes.data) if prec == 32 else self.L.oracle_uniform_f64(st.ctypes.data)

    def xorwow_states(self, seed, subsequences):
        out = np.zeros((len(subsequences), 6), np.uint32)
        for k, s in enumerate(subsequences):
            self.L.oracle_xorwow_init(seed, int(s), 0, 0, out[k].ctypes.data)
        return out

    # -- renders
    @staticmethod
    def camera_to_flat(cam, prec):
        dt = _dt(prec)
        ints = np.array([cam.img_width, cam.img_height, cam.samples_per_pixel, cam.max_depth], np.int32)
        flat = np.array([cam.pixel_samples_scale, *cam.center, *cam.pixel00_loc, *cam.pixel_delta_u, *cam.pixel_delta_v,
                         cam.defocus_angle, *cam.defocus_disk_u, *cam.defocus_disk_v], dt)
        return ints, flat

    def render(self, prec, scene, cam, seed=1227, row0=0, row1=None, loop_form=ITERATIVE, sky_mode=SKY_PRIMARY, segments=False):
        """CUDA-policy render of a COMPACT scene (all slots valid). cam: api.CameraF32/F64.
        The defaults are the reference's GPU program (iterative loop, sky from the primary ray);
        the two switches exist for the pins in tests/test_oracle_pins.py."""
        dt = _dt(prec)
        ints, flat = self.camera_to_flat(cam, prec)
        W, H = int(ints[0]), int(ints[1])
        row1 = H if row1 is None else row1
        cr = np.ascontiguousarray(scene["center_radius"], dt); af = np.ascontiguousarray(scene["albedo_fuzz"], dt)
        ri = np.ascontiguousarray(scene["refraction_index"], dt); ty = np.ascontiguousarray(scene["type"], np.int32)
        out = np.zeros((row1 - row0, W, 3), dt)
        stats = np.zeros(4, np.uint64)
        seg = np.zeros((row1 - row0, W), np.uint32) if segments else None
        rc = self.L.oracle_render_modes(prec, len(ty), cr.ctypes.data, af.ctypes.data, ri.ctypes.data, ty.ctypes.data,
                                        ints.ctypes.data, flat.ctypes.data, seed, row0, row1, loop_form, sky_mode,
                                        out.ctypes.data, stats.ctypes.data, seg.ctypes.data if segments else None)
        assert rc == 0
        if segments:
            return out, [int(x) for x in stats], seg
        return out, [int(x) for x in stats]

    def hit_world(self, prec, center_radius, rays):
        """hit_world alone on rays [n, 6]: (t [n] (+inf: none), sphere index [n] (-1: none))."""
        dt = _dt(prec)
        cr = np.ascontiguousarray(center_radius, dt); rays = np.ascontiguousarray(rays, dt)
        t = np.zeros(len(rays), dt); idx = np.zeros(len(rays), np.int32)
        self.L.oracle_hit_world.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        assert self.L.oracle_hit_world(prec, len(cr), cr.ctypes.data, len(rays), rays.ctypes.data, t.ctypes.data, idx.ctypes.data) == 0
        return t, idx

    def render_serial(self, scene_id, W, H, S, depth, loop_form=RECURSIVE, sky_mode=SKY_CURRENT):
        """Serial-policy render (P3 text).  The defaults are the reference's serial program."""
        cap = W * H * 12 + 64
        buf = ctypes.create_string_buffer(cap)
        stats = np.zeros(4, np.uint64)
        n = self.L.oracle_render_serial_modes(scene_id, W, H, S, depth, loop_form, sky_mode, buf, cap, stats.ctypes.data)
        return buf.raw[:n], [int(x) for x in stats]

    def sky(self, policy, direction):
        """Sky term alone: policy 0 = serial (fp64), 32 / 64 = CUDA policy in that precision."""
        d = np.ascontiguousarray(direction, np.float64); out = np.zeros(3)
        assert self.L.oracle_sky(policy, d.ctypes.data, out.ctypes.data) == 0
        return out

    def set_threads(self, n):
        self.L.oracle_set_threads(n)
