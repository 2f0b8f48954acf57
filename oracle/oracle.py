"""ctypes wrapper of the CPU oracle (oracle/libprt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package.  PARITY UNPINNED (see prt_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from parallelraytracing_amd.capi import (HIT_DTYPE, PrtCameraDesc, PrtHit, PrtMaterial, PrtPrimitive, PrtSampling,
                                         PrtSceneDesc)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libprt_oracle.so")

_fp = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_vp = C.c_void_p


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("prt_oracle.cpp", "prt_oracle.h", "Makefile")]
    src.append(os.path.join(_HERE, "..", "include", "prt.h"))
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libprt_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    L.orc_pcg_hash.restype = C.c_uint32
    L.orc_pcg_hash.argtypes = [C.c_uint32]
    L.orc_path_seed.restype = C.c_uint32
    L.orc_path_seed.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_random.restype = C.c_float
    L.orc_random.argtypes = [_u32p]
    L.orc_random_unit_vector.restype = None
    L.orc_random_unit_vector.argtypes = [_u32p, _fp]
    L.orc_camera_basis.restype = None
    L.orc_camera_basis.argtypes = [C.POINTER(PrtCameraDesc), _fp, _fp, _fp]
    L.orc_camera_rays.restype = None
    L.orc_camera_rays.argtypes = [C.POINTER(PrtCameraDesc), C.c_uint32, _fp, _fp, _fp, _fp]
    L.orc_shape_intersect.restype = C.c_int
    L.orc_shape_intersect.argtypes = [C.c_int, _fp, _fp, _fp, _fp, _fp, C.POINTER(C.c_int)]
    L.orc_transform_point.restype = None
    L.orc_transform_point.argtypes = [_fp, _fp, _fp]
    L.orc_transform_normal.restype = None
    L.orc_transform_normal.argtypes = [_fp, _fp, _fp]
    L.orc_make_transform.restype = None
    L.orc_make_transform.argtypes = [_fp, _fp, _fp, _fp, _fp]
    L.orc_scene_preset.restype = C.c_int
    L.orc_scene_preset.argtypes = [C.c_int, C.POINTER(PrtMaterial), _u32p, C.POINTER(PrtPrimitive), _u32p]
    L.orc_scene_create.restype = _vp
    L.orc_scene_create.argtypes = [C.POINTER(PrtSceneDesc)]
    L.orc_scene_destroy.restype = None
    L.orc_scene_destroy.argtypes = [_vp]
    L.orc_scene_prim_count.restype = C.c_uint32
    L.orc_scene_prim_count.argtypes = [_vp]
    L.orc_closest_hit.restype = None
    L.orc_closest_hit.argtypes = [_vp, C.c_uint32, _fp, _fp, C.POINTER(PrtHit), C.c_int, C.c_int]
    L.orc_scatter.restype = C.c_int
    L.orc_scatter.argtypes = [C.POINTER(PrtMaterial), _fp, C.POINTER(PrtHit), _u32p, _fp, _fp, _fp, _fp]
    L.orc_scatter_batch.restype = None
    L.orc_scatter_batch.argtypes = [C.POINTER(PrtMaterial), C.c_uint32, _fp, C.POINTER(PrtHit), _u32p, _u32p, _fp, _fp, _fp, _fp]
    L.orc_fresnel.restype = C.c_float
    L.orc_fresnel.argtypes = [C.c_float, C.c_float]
    L.orc_fresnel_libm.restype = C.c_float
    L.orc_fresnel_libm.argtypes = [C.c_float, C.c_float]
    L.orc_fresnel_batch.restype = None
    L.orc_fresnel_batch.argtypes = [C.c_uint32, _fp, _fp, _fp, _fp]
    L.orc_trace.restype = None
    L.orc_trace.argtypes = [_vp, _fp, _fp, C.c_int, _u32p, C.c_int, C.c_int, _fp, _u32p]
    L.orc_render.restype = None
    L.orc_render.argtypes = [_vp, C.POINTER(PrtCameraDesc)] + [C.c_uint32] * 8 + [C.c_int, C.c_uint32, C.c_int,
                                                                                  C.c_int, C.c_int, _fp, _fp,
                                                                                  C.POINTER(C.c_uint64)]
    L.orc_render_sampling.restype = None
    L.orc_render_sampling.argtypes = [_vp, C.POINTER(PrtCameraDesc)] + [C.c_uint32] * 8 + [
        C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(PrtSampling), _fp, _fp, C.POINTER(C.c_uint64)]
    L.orc_tonemap.restype = None
    L.orc_tonemap.argtypes = [_fp, _fp, C.c_uint32, C.c_float, C.c_float, C.POINTER(C.c_uint8)]
    L.orc_aabb_intersect_p.restype = C.c_int
    L.orc_aabb_intersect_p.argtypes = [_fp, _fp, _fp, _fp]
    _lib = L
    return L


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def pcg_hash(v: int) -> int:
    return lib().orc_pcg_hash(v & 0xFFFFFFFF)


def path_seed(pixel: int, sample: int, seed: int) -> int:
    return lib().orc_path_seed(pixel, sample, seed)


def random_floats(state: int, n: int):
    st = C.c_uint32(state)
    out = np.empty(n, np.float32)
    for i in range(n):
        out[i] = lib().orc_random(C.byref(st))
    return out, st.value


def random_unit_vector(state: int):
    st = C.c_uint32(state)
    out = np.empty(3, np.float32)
    lib().orc_random_unit_vector(C.byref(st), out.ctypes.data_as(_fp))
    return out, st.value


def camera_basis(cam: PrtCameraDesc):
    f, r, u = (np.empty(3, np.float32) for _ in range(3))
    lib().orc_camera_basis(C.byref(cam), f.ctypes.data_as(_fp), r.ctypes.data_as(_fp), u.ctypes.data_as(_fp))
    return f, r, u


def camera_rays(cam: PrtCameraDesc, px, py):
    px, ppx = _f(px)
    py, ppy = _f(py)
    n = px.size
    o = np.empty((n, 3), np.float32)
    d = np.empty((n, 3), np.float32)
    lib().orc_camera_rays(C.byref(cam), n, ppx, ppy, o.ctypes.data_as(_fp), d.ctypes.data_as(_fp))
    return o, d


def shape_intersect(shape_type: int, params, o, d):
    params, pp = _f(params)
    o, po = _f(o)
    d, pd = _f(d)
    pos = np.zeros(3, np.float32)
    nrm = np.zeros(3, np.float32)
    front = C.c_int(0)
    has = lib().orc_shape_intersect(shape_type, pp, po, pd, pos.ctypes.data_as(_fp), nrm.ctypes.data_as(_fp),
                                    C.byref(front))
    return bool(has), pos, nrm, bool(front.value)


def make_transform(scale, euler_deg, translation):
    s, ps = _f(scale)
    e, pe = _f(euler_deg)
    t, pt = _f(translation)
    mat = np.empty(16, np.float32)
    inv = np.empty(16, np.float32)
    lib().orc_make_transform(ps, pe, pt, mat.ctypes.data_as(_fp), inv.ctypes.data_as(_fp))
    return mat, inv


def transform_point(m, p):
    m, pm = _f(m)
    p, pp = _f(p)
    out = np.empty(3, np.float32)
    lib().orc_transform_point(pm, pp, out.ctypes.data_as(_fp))
    return out


def transform_normal(m, n):
    m, pm = _f(m)
    n, pn = _f(n)
    out = np.empty(3, np.float32)
    lib().orc_transform_normal(pm, pn, out.ctypes.data_as(_fp))
    return out


def scene_preset(preset: int):
    nm = C.c_uint32(0)
    npr = C.c_uint32(0)
    rc = lib().orc_scene_preset(preset, None, C.byref(nm), None, C.byref(npr))
    if rc:
        raise ValueError(f"unknown preset {preset}")
    mats = (PrtMaterial * nm.value)()
    prims = (PrtPrimitive * npr.value)()
    lib().orc_scene_preset(preset, mats, C.byref(nm), prims, C.byref(npr))
    return mats, prims


class OracleScene:
    """Owns an OrcScene built from a PrtSceneDesc (the same struct the product consumes)."""

    def __init__(self, desc: PrtSceneDesc):
        self._h = lib().orc_scene_create(C.byref(desc))
        self.n_materials = desc.n_materials
        self._materials = [desc.materials[i] for i in range(desc.n_materials)]

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scene_destroy(self._h)
            self._h = None

    @property
    def prim_count(self):
        return lib().orc_scene_prim_count(self._h)

    def closest_hit(self, origins, dirs, use_bvh=False, n_threads=1):
        o, po = _f(origins)
        d, pd = _f(dirs)
        n = o.shape[0]
        hits = np.zeros(n, dtype=HIT_DTYPE)
        lib().orc_closest_hit(self._h, n, po, pd, hits.ctypes.data_as(C.POINTER(PrtHit)), int(use_bvh), n_threads)
        return hits

    def trace(self, o, d, max_depth, rng_state, iterative=False, use_bvh=False):
        o, po = _f(o)
        d, pd = _f(d)
        st = C.c_uint32(rng_state)
        L = np.empty(3, np.float32)
        segs = C.c_uint32(0)
        lib().orc_trace(self._h, po, pd, max_depth, C.byref(st), int(iterative), int(use_bvh),
                        L.ctypes.data_as(_fp), C.byref(segs))
        return L, segs.value, st.value

    def render(self, cam: PrtCameraDesc, W, H, spp=1, first_sample=0, max_depth=20, seed=0, iterative=False,
               use_bvh=False, n_threads=1, rect=None, accum=None, weights=None, sampling=None):
        """sampling: optional PrtSampling (jitter / Russian roulette / clamp); None = the reference CPU backend."""
        if accum is None:
            accum = np.zeros((H, W, 3), np.float32)
            weights = np.zeros((H, W), np.float32)
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, W, H)
        rays = C.c_uint64(0)
        if sampling is not None:
            lib().orc_render_sampling(self._h, C.byref(cam), W, H, x0, y0, x1, y1, spp, first_sample, max_depth, seed,
                                      int(iterative), int(use_bvh), n_threads, C.byref(sampling),
                                      accum.ctypes.data_as(_fp), weights.ctypes.data_as(_fp), C.byref(rays))
            return accum, weights, rays.value
        lib().orc_render(self._h, C.byref(cam), W, H, x0, y0, x1, y1, spp, first_sample, max_depth, seed,
                         int(iterative), int(use_bvh), n_threads, accum.ctypes.data_as(_fp),
                         weights.ctypes.data_as(_fp), C.byref(rays))
        return accum, weights, rays.value


def scatter(material: PrtMaterial, in_dir, hit_record, rng_state: int):
    """hit_record: one element of a HIT_DTYPE array."""
    d, pd = _f(in_dir)
    h = PrtHit()
    h.prim = int(hit_record["prim"])
    h.front_face = int(hit_record["front_face"])
    h.material_id = int(hit_record["material_id"])
    h.d2 = float(hit_record["d2"])
    for k in range(3):
        h.position[k] = float(hit_record["position"][k])
        h.normal[k] = float(hit_record["normal"][k])
    st = C.c_uint32(rng_state)
    att, em, oo, od = (np.zeros(3, np.float32) for _ in range(4))
    sc = lib().orc_scatter(C.byref(material), pd, C.byref(h), C.byref(st), att.ctypes.data_as(_fp),
                           em.ctypes.data_as(_fp), oo.ctypes.data_as(_fp), od.ctypes.data_as(_fp))
    return bool(sc), att, em, oo, od, st.value


def scatter_batch(materials, in_dirs, hits: np.ndarray, rng_state):
    """materials: list of PrtMaterial; hits: HIT_DTYPE array (material_id indexes `materials`).
    Returns (scattered, attenuation, emitted, out_origin, out_dir, rng_state_after) like HipWavefrontRenderer.scatter."""
    mats = (PrtMaterial * len(materials))(*materials)
    d, pd = _f(np.asarray(in_dirs, np.float32).reshape(-1, 3))
    n = d.shape[0]
    hits = np.ascontiguousarray(hits)
    rng = np.ascontiguousarray(rng_state, dtype=np.uint32).copy()
    sc = np.zeros(n, np.uint32)
    att, em, oo, od = (np.zeros((n, 3), np.float32) for _ in range(4))
    lib().orc_scatter_batch(mats, n, pd, hits.ctypes.data_as(C.POINTER(PrtHit)), rng.ctypes.data_as(_u32p),
                            sc.ctypes.data_as(_u32p), att.ctypes.data_as(_fp), em.ctypes.data_as(_fp),
                            oo.ctypes.data_as(_fp), od.ctypes.data_as(_fp))
    return sc.astype(bool), att, em, oo, od, rng


def fresnel_batch(cosine, ri):
    """(contract form, literal libm form) of fresnelReflectance for arrays of inputs."""
    c, pc = _f(cosine)
    r, pr = _f(ri)
    a = np.empty(c.size, np.float32)
    b = np.empty(c.size, np.float32)
    lib().orc_fresnel_batch(c.size, pc, pr, a.ctypes.data_as(_fp), b.ctypes.data_as(_fp))
    return a, b


def tonemap(accum, weights, exposure=1.0, gamma=2.2):
    a, pa = _f(accum)
    w, pw = _f(weights)
    n = w.size
    out = np.empty((n, 4), np.uint8)
    lib().orc_tonemap(pa, pw, n, exposure, gamma, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.reshape(w.shape + (4,))


def aabb_intersect_p(bmin, bmax, o, d) -> bool:
    a, pa = _f(bmin)
    b, pb = _f(bmax)
    o, po = _f(o)
    d, pd = _f(d)
    return bool(lib().orc_aabb_intersect_p(pa, pb, po, pd))
