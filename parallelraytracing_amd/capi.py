"""ctypes binding of include/prt.h (libprt.so).

The library is the product: there is no Python or CPU fallback.  If libprt.so is missing the import
fails loudly with the build command to run.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PRT_LIB_PATH: tuning hook (tools/ab_libs.py times alternative BUILDS of the same sources, e.g. other compiler flags)
LIB_PATH = os.environ.get("PRT_LIB_PATH") or os.path.join(_HERE, "csrc", "libprt.so")

PRT_MAX_DEPTH = 64

# shape / material / preset enums (include/prt.h)
SHAPE_CIRCLE, SHAPE_QUAD, SHAPE_TRIANGLE = 0, 1, 2
MAT_NONE, MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_EMISSIVE = 0, 1, 2, 3, 4
(PRESET_DEFAULT, PRESET_LIGHT_TEST, PRESET_MATERIAL_TEST, PRESET_CORNELL, PRESET_RANDOM_BALLS_SMALL,
 PRESET_RANDOM_BALLS_MEDIUM, PRESET_RANDOM_BALLS_LARGE) = range(7)
PRESET_NAMES = {
    "DEFAULT": PRESET_DEFAULT, "LIGHT_TEST": PRESET_LIGHT_TEST, "MATERIAL_TEST": PRESET_MATERIAL_TEST,
    "CORNELL": PRESET_CORNELL, "RANDOM_BALLS_SMALL": PRESET_RANDOM_BALLS_SMALL,
    "RANDOM_BALLS_MEDIUM": PRESET_RANDOM_BALLS_MEDIUM, "RANDOM_BALLS_LARGE": PRESET_RANDOM_BALLS_LARGE,
}


class PrtMaterial(C.Structure):
    _fields_ = [("type", C.c_uint32), ("rgb", C.c_float * 3), ("scalar", C.c_float)]


class PrtPrimitive(C.Structure):
    _fields_ = [("shape_type", C.c_uint32), ("shape_param", C.c_float * 2), ("material_id", C.c_uint32),
                ("mat", C.c_float * 16), ("inv", C.c_float * 16)]


class PrtMesh(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("indices", C.POINTER(C.c_uint32)), ("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32),
                ("material_id", C.c_uint32)]


class PrtInstance(C.Structure):
    _fields_ = [("mesh", C.c_uint32), ("material_id", C.c_uint32), ("mat", C.c_float * 16), ("inv", C.c_float * 16)]


class PrtSceneDesc(C.Structure):
    _fields_ = [("materials", C.POINTER(PrtMaterial)), ("primitives", C.POINTER(PrtPrimitive)),
                ("meshes", C.POINTER(PrtMesh)), ("n_materials", C.c_uint32), ("n_primitives", C.c_uint32),
                ("n_meshes", C.c_uint32), ("sky", C.c_float * 3),
                ("instanced_meshes", C.POINTER(PrtMesh)), ("instances", C.POINTER(PrtInstance)),
                ("n_instanced_meshes", C.c_uint32), ("n_instances", C.c_uint32)]


class PrtCameraDesc(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("front", C.c_float * 3), ("width", C.c_float), ("height", C.c_float)]


class PrtHit(C.Structure):
    _fields_ = [("prim", C.c_int32), ("front_face", C.c_uint32), ("material_id", C.c_uint32), ("d2", C.c_float),
                ("position", C.c_float * 3), ("normal", C.c_float * 3)]


class PrtStats(C.Structure):
    _fields_ = [("rays_total", C.c_uint64), ("rays_per_depth", C.c_uint64 * PRT_MAX_DEPTH), ("samples", C.c_uint64),
                ("intersect_launches", C.c_uint64), ("intersect_ms", C.c_double), ("shade_ms", C.c_double),
                ("raygen_ms", C.c_double), ("accumulate_ms", C.c_double), ("bvh_node_visits", C.c_uint64),
                ("bvh_tri_tests", C.c_uint64), ("prim_tests", C.c_uint64), ("node_lane_slots", C.c_uint64),
                ("scan_ms", C.c_double), ("rays_traversed", C.c_uint64), ("tri_lane_slots", C.c_uint64), ("max_stack_used", C.c_uint64),
                ("wave_cycles_refill", C.c_uint64), ("wave_cycles_node", C.c_uint64), ("wave_cycles_tri", C.c_uint64)]


class PrtOccupancy(C.Structure):
    _fields_ = [("blocks_per_cu", C.c_uint32), ("waves_per_cu", C.c_uint32), ("max_waves_per_cu", C.c_uint32),
                ("vgprs", C.c_uint32), ("lds_bytes_per_block", C.c_uint32), ("compute_units", C.c_uint32),
                ("resident_grid_blocks", C.c_uint32)]


class PrtSampling(C.Structure):
    _fields_ = [("jitter", C.c_uint32), ("rr_depth", C.c_uint32), ("clamp", C.c_float)]


class PrtBvhInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_triangles", C.c_uint32), ("max_depth", C.c_uint32),
                ("max_leaf_size", C.c_uint32), ("sah_cost", C.c_float), ("pad_abs", C.c_float),
                ("node_bytes", C.c_uint64), ("tri_bytes", C.c_uint64), ("n_nodes4", C.c_uint32), ("max_stack4", C.c_uint32),
                ("n_nodes8", C.c_uint32), ("depth8", C.c_uint32), ("build_ms", C.c_float),
                ("built_on_device", C.c_uint32), ("refit_ms", C.c_float), ("refits", C.c_uint32)]


# numpy dtype mirror of PrtHit (40 bytes)
HIT_DTYPE = [("prim", "<i4"), ("front_face", "<u4"), ("material_id", "<u4"), ("d2", "<f4"),
             ("position", "<f4", (3,)), ("normal", "<f4", (3,))]

_vp = C.c_void_p
_fp = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)

# name -> (restype, argtypes).  Every symbol include/prt.h declares is listed here; the CPU test-suite
# checks that each one is exported by the built library.
SIGNATURES = {
    "prt_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "prt_destroy": (None, [_vp]),
    "prt_last_error": (C.c_char_p, [_vp]),
    "prt_version": (C.c_int, []),
    "prt_set_stream": (C.c_int, [_vp, _vp]),
    "prt_get_stream": (C.c_int, [_vp, C.POINTER(_vp)]),
    "prt_get_device": (C.c_int, [_vp]),
    "prt_clone_scene": (C.c_int, [_vp, _vp]),
    "prt_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_uint32, C.POINTER(_vp)]),
    "prt_group_destroy": (None, [_vp]),
    "prt_group_last_error": (C.c_char_p, [_vp]),
    "prt_group_size": (C.c_uint32, [_vp]),
    "prt_group_transport": (C.c_char_p, [_vp]),
    "prt_group_context": (_vp, [_vp, C.c_uint32]),
    "prt_group_set_scene": (C.c_int, [_vp, C.POINTER(PrtSceneDesc)]),
    "prt_group_refit_meshes": (C.c_int, [_vp, C.POINTER(PrtMesh), C.c_uint32]),
    "prt_group_set_camera": (C.c_int, [_vp, C.POINTER(PrtCameraDesc)]),
    "prt_group_set_film": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "prt_group_film_clear": (C.c_int, [_vp]),
    "prt_group_set_sampling": (C.c_int, [_vp, C.POINTER(PrtSampling)]),
    "prt_group_set_samples_in_flight": (C.c_int, [_vp, C.c_uint32]),
    "prt_group_set_param": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "prt_group_render": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "prt_group_film_read": (C.c_int, [_vp, _fp, _fp]),
    "prt_group_film_display": (C.c_int, [_vp, C.c_float, C.c_float, C.POINTER(C.c_uint8)]),
    "prt_group_get_stats": (C.c_int, [_vp, C.POINTER(PrtStats)]),
    "prt_set_scene": (C.c_int, [_vp, C.POINTER(PrtSceneDesc)]),
    "prt_set_camera": (C.c_int, [_vp, C.POINTER(PrtCameraDesc)]),
    "prt_set_film": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "prt_film_clear": (C.c_int, [_vp]),
    "prt_render": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "prt_render_async": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "prt_synchronize": (C.c_int, [_vp]),
    "prt_set_samples_in_flight": (C.c_int, [_vp, C.c_uint32]),
    "prt_film_read": (C.c_int, [_vp, _fp, _fp]),
    "prt_film_local": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_uint64)]),
    "prt_film_resolve": (C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp]),
    "prt_film_resolve_on": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, _vp]),
    "prt_film_tonemap": (C.c_int, [_vp, _vp, _vp, C.c_float, C.c_float, _vp]),
    "prt_film_display": (C.c_int, [_vp, C.c_float, C.c_float, C.POINTER(C.c_uint8)]),
    "prt_camera_rays": (C.c_int, [_vp, C.c_uint32, _fp, _fp, _fp, _fp]),
    "prt_closest_hit": (C.c_int, [_vp, C.c_uint32, _fp, _fp, C.POINTER(PrtHit)]),
    "prt_scatter": (C.c_int, [_vp, C.c_uint32, _fp, C.POINTER(PrtHit), _u32p, _u32p, _fp, _fp, _fp, _fp]),
    "prt_enable_timing": (C.c_int, [_vp, C.c_int]),
    "prt_get_stats": (C.c_int, [_vp, C.POINTER(PrtStats)]),
    "prt_reset_stats": (C.c_int, [_vp]),
    "prt_measure_traversal": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(PrtStats)]),
    "prt_bvh_info": (C.c_int, [_vp, C.POINTER(PrtBvhInfo)]),
    "prt_kernel_occupancy": (C.c_int, [_vp, C.POINTER(PrtOccupancy)]),
    "prt_kernel_instance": (C.c_int, [_vp, C.c_char_p, C.c_uint32]),
    "prt_measure_shade_divergence": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "prt_refit_meshes": (C.c_int, [_vp, C.POINTER(PrtMesh), C.c_uint32]),
    "prt_bvh_read": (C.c_int, [_vp, _fp, _fp]),
    "prt_set_sampling": (C.c_int, [_vp, C.POINTER(PrtSampling)]),
    "prt_bvh_read4": (C.c_int, [_vp, _fp]),
    "prt_bvh_read8": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "prt_set_variant": (C.c_int, [_vp, C.c_int]),
    "prt_set_param": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "prt_mesh_load_ply": (C.c_int, [C.c_char_p, C.POINTER(_vp), C.c_char_p, C.c_size_t]),
    "prt_mesh_create": (C.c_int, [_fp, _fp, C.c_uint32, _u32p, C.c_uint32, C.POINTER(_vp)]),
    "prt_mesh_free": (None, [_vp]),
    "prt_mesh_vertex_count": (C.c_uint32, [_vp]),
    "prt_mesh_triangle_count": (C.c_uint32, [_vp]),
    "prt_mesh_positions": (_fp, [_vp]),
    "prt_mesh_normals": (_fp, [_vp]),
    "prt_mesh_indices": (_u32p, [_vp]),
    "prt_mesh_had_normals": (C.c_int, [_vp]),
    "prt_mesh_refine": (C.c_int, [_vp, C.c_uint32]),
    "prt_mesh_transform": (C.c_int, [_vp, _fp, _fp]),
    "prt_mesh_append": (C.c_int, [_vp, _vp]),
    "prt_scene_preset": (C.c_int, [C.c_int, C.POINTER(PrtMaterial), _u32p, C.POINTER(PrtPrimitive), _u32p]),
    "prt_make_transform": (None, [_fp, _fp, _fp, _fp, _fp]),
    "prt_write_ppm": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32]),
    "prt_write_pfm": (C.c_int, [C.c_char_p, _fp, C.c_uint32, C.c_uint32]),
}


def load(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension is the product and has no fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C parallelraytracing_amd/csrc`).")
    # torch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one process do
    # not coexist, so when torch is importable it must load first (libprt.so then binds to the same runtime).
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing only: the library also runs without it (system ROCm runtime)
        pass
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = load()
    return _lib
