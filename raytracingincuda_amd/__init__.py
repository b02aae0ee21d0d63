"""raytracingincuda_amd -- MI355X (gfx950) native `render` hot path of RayTracingInOneWeekend.

Host-side mirror (Python over the C-ABI in include/rtiow.h and include/rtiow_host.h) of the
launch sequence of the reference's src/Global{Float,Double}CUDAInOneWeekend/main.cu.
The HIP library is mandatory: nothing here falls back to a CPU path.
"""
from .api import (  # noqa: F401
    LAMBERTIAN, METAL, DIELECTRIC, SCENE_LDS, SCENE_SCALAR, SCENE_LDS_EXACT, SCENE_GRID, SCHED_STATIC, SCHED_PERSISTENT, SCHED_SORTED,
    RtiowError, Renderer, RendererGroup, GATHER_AUTO, GATHER_RCCL, GATHER_PEER, GATHER_HOST, build_scene, camera, compact_scene, ppm_filename, format_ppm, write_ppm, write_ppm_levels, levels,
    place_rows, shard_rows, scene_slots, load_hip_library, load_host_library, lib_paths, ABI_VERSION, build_id, debug_gather_schedule,
)

__all__ = [
    "LAMBERTIAN", "METAL", "DIELECTRIC", "SCENE_LDS", "SCENE_SCALAR", "SCENE_LDS_EXACT", "SCENE_GRID", "SCHED_STATIC", "SCHED_PERSISTENT", "SCHED_SORTED", "RtiowError", "Renderer", "RendererGroup", "GATHER_AUTO", "GATHER_RCCL", "GATHER_PEER", "GATHER_HOST",
    "build_scene", "camera", "compact_scene", "ppm_filename", "format_ppm", "write_ppm", "write_ppm_levels", "levels", "place_rows", "shard_rows",
    "scene_slots", "load_hip_library", "load_host_library", "lib_paths", "ABI_VERSION", "build_id", "debug_gather_schedule",
]
