"""Host-side mirror of the reference's scene / camera / film / renderer interfaces, over the C-ABI.

Same names and call contract as the reference (C++) so that callers and tests read alike:
  Scene(preset)                      src/core/scene.h:17-62
  Mesh(ply_path)                     src/core/mesh.h:8-21
  Camera(position, front, w, h)      src/core/camera.h:7-16
  Film(width, height)                src/core/film.h:10-47
  HipWavefrontRenderer.Init / ProgressiveRender / SetCamera     src/core/renderer.h:8-16
Nothing here computes pixels: every method marshals PODs into libprt.so (HIP).  A C++ adapter with the
same shape lives in host/prt_renderer.hpp.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from . import capi
from .capi import (PrtBvhInfo, PrtCameraDesc, PrtHit, PrtInstance, PrtMaterial, PrtMesh, PrtPrimitive, PrtSampling,
                   PrtSceneDesc, PrtStats)

_fp = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)

SKY = (0.4, 0.3, 0.6)  # src/backend/cpu/renderer.h:31
DEFAULT_MAX_DEPTH = 20  # src/backend/cpu/renderer.h:34


class PrtError(RuntimeError):
    pass


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def glm_normalize(v) -> np.ndarray:
    """glm::normalize in fp32: v * (1 / sqrt((x*x + y*y) + z*z))."""
    v = np.asarray(v, np.float32)
    d = np.float32(np.float32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
    return (v * np.float32(np.float32(1.0) / np.sqrt(d))).astype(np.float32)


def make_transform(scale, euler_deg, translation) -> Tuple[np.ndarray, np.ndarray]:
    """Scene::MakeTransform (src/core/scene.cpp:9-17): column-major mat and inverse."""
    s, e, t = _f32(scale), _f32(euler_deg), _f32(translation)
    mat = np.empty(16, np.float32)
    inv = np.empty(16, np.float32)
    capi.lib().prt_make_transform(s.ctypes.data_as(_fp), e.ctypes.data_as(_fp), t.ctypes.data_as(_fp),
                                  mat.ctypes.data_as(_fp), inv.ctypes.data_as(_fp))
    return mat, inv


class Mesh:
    """Triangle mesh: vertices / normals / indices, as the reference's Mesh (src/core/mesh.h:12-14)."""

    def __init__(self, ply_path: Optional[str] = None, *, vertices=None, normals=None, indices=None):
        L = capi.lib()
        h = C.c_void_p()
        if ply_path is not None:
            err = C.create_string_buffer(256)
            rc = L.prt_mesh_load_ply(ply_path.encode(), C.byref(h), err, len(err))
            if rc:
                raise PrtError(f"PLY load failed ({ply_path}): {err.value.decode()}")
        else:
            v = _f32(vertices).reshape(-1, 3)
            i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
            n = None if normals is None else _f32(normals).reshape(-1, 3)
            rc = L.prt_mesh_create(v.ctypes.data_as(_fp), None if n is None else n.ctypes.data_as(_fp), v.shape[0],
                                   i.ctypes.data_as(_u32p), i.shape[0], C.byref(h))
            if rc:
                raise PrtError("prt_mesh_create failed (index out of range?)")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            capi.lib().prt_mesh_free(self._h)
            self._h = None

    @property
    def n_vertices(self) -> int:
        return capi.lib().prt_mesh_vertex_count(self._h)

    @property
    def n_triangles(self) -> int:
        return capi.lib().prt_mesh_triangle_count(self._h)

    @property
    def had_normals(self) -> bool:
        return bool(capi.lib().prt_mesh_had_normals(self._h))

    def GetVertices(self) -> np.ndarray:
        return np.ctypeslib.as_array(capi.lib().prt_mesh_positions(self._h), (self.n_vertices, 3)).copy()

    def GetNormals(self) -> np.ndarray:
        return np.ctypeslib.as_array(capi.lib().prt_mesh_normals(self._h), (self.n_vertices, 3)).copy()

    def GetIndices(self) -> np.ndarray:
        return np.ctypeslib.as_array(capi.lib().prt_mesh_indices(self._h), (self.n_triangles, 3)).copy()

    def refine(self, target_triangles: int) -> "Mesh":
        if capi.lib().prt_mesh_refine(self._h, int(target_triangles)):
            raise PrtError("prt_mesh_refine failed (non-manifold edge)")
        return self

    def transform(self, mat, inv) -> "Mesh":
        m, i = _f32(mat), _f32(inv)
        capi.lib().prt_mesh_transform(self._h, m.ctypes.data_as(_fp), i.ctypes.data_as(_fp))
        return self

    def append(self, other: "Mesh") -> "Mesh":
        capi.lib().prt_mesh_append(self._h, other._h)
        return self

    def copy(self) -> "Mesh":
        return Mesh(vertices=self.GetVertices(), normals=self.GetNormals(), indices=self.GetIndices())


class Scene:
    """Materials + analytic primitives (+ triangle meshes).  Scene(preset) reproduces the reference's
    hard-coded presets (src/core/scene.cpp:42-55); the default preset is RANDOM_BALLS_LARGE
    (src/core/scene.h:20).  preset=None gives an empty scene to fill by hand."""

    def __init__(self, preset: Optional[str] = "RANDOM_BALLS_LARGE", sky=SKY):
        self.materials: List[PrtMaterial] = []
        self.primitives: List[PrtPrimitive] = []
        self.meshes: List[Tuple[Mesh, int]] = []
        self.instanced_meshes: List[Mesh] = []
        self.instances: List[PrtInstance] = []
        self.sky = tuple(float(x) for x in sky)
        self._keep = None
        if preset is not None:
            pid = capi.PRESET_NAMES[preset] if isinstance(preset, str) else int(preset)
            nm, npr = C.c_uint32(0), C.c_uint32(0)
            if capi.lib().prt_scene_preset(pid, None, C.byref(nm), None, C.byref(npr)):
                raise PrtError(f"unknown preset {preset}")
            mats = (PrtMaterial * nm.value)()
            prims = (PrtPrimitive * npr.value)()
            capi.lib().prt_scene_preset(pid, mats, C.byref(nm), prims, C.byref(npr))
            self.materials = list(mats)
            self.primitives = list(prims)

    # MaterialPool::Add* (src/core/material.h:170-192)
    def _add_material(self, mtype, rgb, scalar) -> int:
        m = PrtMaterial()
        m.type = mtype
        m.rgb[:] = [float(x) for x in rgb]
        m.scalar = float(scalar)
        self.materials.append(m)
        return len(self.materials) - 1

    def AddLambertian(self, albedo) -> int:
        return self._add_material(capi.MAT_LAMBERTIAN, albedo, 0.0)

    def AddMetal(self, albedo, roughness) -> int:
        return self._add_material(capi.MAT_METAL, albedo, roughness)

    def AddDielectric(self, ri) -> int:
        return self._add_material(capi.MAT_DIELECTRIC, (0, 0, 0), ri)

    def AddEmissive(self, emission) -> int:
        return self._add_material(capi.MAT_EMISSIVE, emission, 0.0)

    # Scene::AddPrimitive (src/core/scene.cpp:19-36)
    def _add_prim(self, shape, p0, p1, material, scale, euler_deg, translation):
        p = PrtPrimitive()
        p.shape_type = shape
        p.shape_param[0] = float(p0)
        p.shape_param[1] = float(p1)
        p.material_id = int(material)
        mat, inv = make_transform(scale, euler_deg, translation)
        p.mat[:] = mat.tolist()
        p.inv[:] = inv.tolist()
        self.primitives.append(p)

    def AddCircle(self, radius, material, scale=(1, 1, 1), euler_deg=(0, 0, 0), translation=(0, 0, 0)):
        self._add_prim(capi.SHAPE_CIRCLE, radius, 0.0, material, scale, euler_deg, translation)

    def AddQuad(self, width, height, material, scale=(1, 1, 1), euler_deg=(0, 0, 0), translation=(0, 0, 0)):
        self._add_prim(capi.SHAPE_QUAD, width, height, material, scale, euler_deg, translation)

    def AddMesh(self, mesh: Mesh, material: int):
        """World-space triangles (identity Transform); appended after all analytic primitives."""
        self.meshes.append((mesh, int(material)))

    def AddInstance(self, mesh: Mesh, material: int, scale=1.0, euler_deg=(0, 0, 0), translation=(0, 0, 0)):
        """A placed copy of `mesh`: Triangle primitives sharing one Transform (src/core/primitive.h:7-12), built by
        Scene::MakeTransform (src/core/scene.cpp:9-17).  Uniform scale only (include/prt.h, PrtInstance)."""
        for k, m in enumerate(self.instanced_meshes):
            if m is mesh:
                mi = k
                break
        else:
            self.instanced_meshes.append(mesh)
            mi = len(self.instanced_meshes) - 1
        inst = PrtInstance()
        inst.mesh = mi
        inst.material_id = int(material)
        sc = (scale, scale, scale) if np.isscalar(scale) else scale
        mat, inv = make_transform(sc, euler_deg, translation)
        inst.mat[:] = mat.tolist()
        inst.inv[:] = inv.tolist()
        self.instances.append(inst)
        return len(self.instances) - 1

    @property
    def n_triangles(self) -> int:
        return (sum(m.n_triangles for m, _ in self.meshes)
                + sum(self.instanced_meshes[i.mesh].n_triangles for i in self.instances))

    def desc(self) -> PrtSceneDesc:
        mats = (PrtMaterial * max(1, len(self.materials)))(*self.materials)
        prims = (PrtPrimitive * max(1, len(self.primitives)))(*self.primitives)
        meshes = (PrtMesh * max(1, len(self.meshes)))()
        L = capi.lib()
        for i, (m, mat) in enumerate(self.meshes):
            meshes[i].positions = L.prt_mesh_positions(m._h)
            meshes[i].normals = L.prt_mesh_normals(m._h)
            meshes[i].indices = L.prt_mesh_indices(m._h)
            meshes[i].n_vertices = m.n_vertices
            meshes[i].n_triangles = m.n_triangles
            meshes[i].material_id = mat
        d = PrtSceneDesc()
        d.materials = mats
        d.primitives = prims
        d.meshes = meshes
        d.n_materials = len(self.materials)
        d.n_primitives = len(self.primitives)
        d.n_meshes = len(self.meshes)
        d.sky[:] = self.sky
        imeshes = (PrtMesh * max(1, len(self.instanced_meshes)))()
        for i, m in enumerate(self.instanced_meshes):
            imeshes[i].positions = L.prt_mesh_positions(m._h)
            imeshes[i].normals = L.prt_mesh_normals(m._h)
            imeshes[i].indices = L.prt_mesh_indices(m._h)
            imeshes[i].n_vertices = m.n_vertices
            imeshes[i].n_triangles = m.n_triangles
        insts = (PrtInstance * max(1, len(self.instances)))(*self.instances)
        d.instanced_meshes = imeshes
        d.instances = insts
        d.n_instanced_meshes = len(self.instanced_meshes)
        d.n_instances = len(self.instances)
        self._keep = (mats, prims, meshes, imeshes, insts)
        return d


class Camera:
    """Camera(position, front, width, height) (src/core/camera.h:10-16); main() places it at (5,5,8)
    looking at the origin (src/main.cpp:142-150)."""

    def __init__(self, position=(5.0, 5.0, 8.0), front=None, width=1920.0, height=1080.0):
        self.position = tuple(float(x) for x in position)
        if front is None:  # glm::normalize(focus - center) with focus = origin (src/main.cpp:142-146)
            front = glm_normalize(-np.asarray(self.position, np.float32))
        self.front = tuple(float(x) for x in front)
        self.width = float(width)
        self.height = float(height)

    def desc(self) -> PrtCameraDesc:
        d = PrtCameraDesc()
        d.position[:] = self.position
        d.front[:] = self.front
        d.width = self.width
        d.height = self.height
        return d


class Film:
    """Accumulation buffer (src/core/film.h:10-76).  The sums live on the GPU; accum / weights are the
    host copies filled by HipWavefrontRenderer.download()."""

    def __init__(self, width: int, height: int):
        self.width = int(width)
        self.height = int(height)
        self.accum = np.zeros((self.height, self.width, 3), np.float32)
        self.weights = np.zeros((self.height, self.width), np.float32)
        self.display = np.zeros((self.height, self.width, 4), np.uint8)
        self._renderer = None

    def GetWidth(self):
        return self.width

    def GetHeight(self):
        return self.height

    def Clear(self):
        self.accum[:] = 0
        self.weights[:] = 0
        self.display[:] = 0
        if self._renderer is not None:
            self._renderer._check(capi.lib().prt_film_clear(self._renderer._ctx))

    def mean(self) -> np.ndarray:
        w = np.maximum(self.weights, 1e-30)[..., None]
        return np.where(self.weights[..., None] > 0, self.accum / w, 0.0).astype(np.float32)


class HipWavefrontRenderer:
    """The MI355X backend behind the reference's Renderer interface (src/core/renderer.h:8-16)."""

    def __init__(self, device: int = 0, max_depth: int = DEFAULT_MAX_DEPTH, seed: int = 0, rank: int = 0,
                 world_size: int = 1):
        self._ctx = C.c_void_p()
        L = capi.lib()
        rc = L.prt_create(device, C.byref(self._ctx))
        if rc:
            msg = L.prt_last_error(self._ctx).decode()
            L.prt_destroy(self._ctx)
            self._ctx = None
            raise PrtError(f"prt_create({device}) failed: {msg}")
        self.max_depth = int(max_depth)
        self.seed = int(seed)
        self.rank = int(rank)
        self.world_size = int(world_size)
        self.frame_index = 0
        self.film: Optional[Film] = None

    def __del__(self):
        if getattr(self, "_ctx", None):
            capi.lib().prt_destroy(self._ctx)
            self._ctx = None

    def _check(self, rc: int):
        if rc:
            raise PrtError(capi.lib().prt_last_error(self._ctx).decode())

    # ---- the reference interface -----------------------------------------------------------------
    def Init(self, film: Film, scene: Scene, camera: Camera):
        L = capi.lib()
        d = scene.desc()
        self._check(L.prt_set_scene(self._ctx, C.byref(d)))
        self._check(L.prt_set_film(self._ctx, film.width, film.height, self.rank, self.world_size))
        self.film = film
        film._renderer = self
        self.frame_index = 0
        self.SetCamera(camera)

    def SetCamera(self, camera: Camera):
        d = camera.desc()
        self._check(capi.lib().prt_set_camera(self._ctx, C.byref(d)))

    def ProgressiveRender(self, spp: int = 1):
        """Adds `spp` (default exactly one) sample per pixel to the film."""
        self._check(capi.lib().prt_render(self._ctx, spp, self.max_depth, self.seed, self.frame_index))
        self.frame_index += spp

    # ---- extensions ----------------------------------------------------------------------------------
    def render_async(self, spp: int = 1):
        self._check(capi.lib().prt_render_async(self._ctx, spp, self.max_depth, self.seed, self.frame_index))
        self.frame_index += spp

    def synchronize(self):
        self._check(capi.lib().prt_synchronize(self._ctx))

    def set_stream(self, hip_stream: int):
        self._check(capi.lib().prt_set_stream(self._ctx, C.c_void_p(hip_stream)))

    def set_samples_in_flight(self, n: int):
        self._check(capi.lib().prt_set_samples_in_flight(self._ctx, n))

    def set_sampling(self, jitter: int = 0, rr_depth: int = 0, clamp: float = 0.0) -> PrtSampling:
        """Optional sampling upgrades (include/prt.h PrtSampling); all zero = the reference CPU backend."""
        sp = PrtSampling(int(jitter), int(rr_depth), float(clamp))
        self._check(capi.lib().prt_set_sampling(self._ctx, C.byref(sp)))
        return sp

    def set_variant(self, v: int):
        self._check(capi.lib().prt_set_variant(self._ctx, v))

    def set_param(self, name: str, value: int):
        self._check(capi.lib().prt_set_param(self._ctx, name.encode(), int(value)))

    def download(self) -> Film:
        f = self.film
        self._check(capi.lib().prt_film_read(self._ctx, f.accum.ctypes.data_as(_fp), f.weights.ctypes.data_as(_fp)))
        return f

    def UpdateDisplay(self, exposure: float = 1.0, gamma: float = 2.2) -> np.ndarray:
        """Film::UpdateDisplayGPU (src/core/film.cu:123-132): RGBA8, row 0 = top."""
        f = self.film
        self._check(capi.lib().prt_film_display(self._ctx, exposure, gamma,
                                                f.display.ctypes.data_as(C.POINTER(C.c_uint8))))
        return f.display

    def film_local(self) -> Tuple[int, int]:
        p = C.c_void_p()
        n = C.c_uint64()
        self._check(capi.lib().prt_film_local(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def film_resolve(self, d_gathered: int, d_rgb: int, d_weight: int, stream: int = 0):
        """stream: a raw HIP stream to launch on (0 = the renderer's own stream)."""
        self._check(capi.lib().prt_film_resolve_on(self._ctx, C.c_void_p(stream), C.c_void_p(d_gathered), self.world_size,
                                                   C.c_void_p(d_rgb), C.c_void_p(d_weight)))

    def film_tonemap(self, d_rgb: int, d_weight: int, d_rgba8: int, exposure: float = 1.0, gamma: float = 2.2):
        self._check(capi.lib().prt_film_tonemap(self._ctx, C.c_void_p(d_rgb), C.c_void_p(d_weight), exposure, gamma,
                                                C.c_void_p(d_rgba8)))

    def camera_rays(self, px, py):
        px, py = _f32(px).ravel(), _f32(py).ravel()
        n = px.size
        o = np.empty((n, 3), np.float32)
        d = np.empty((n, 3), np.float32)
        self._check(capi.lib().prt_camera_rays(self._ctx, n, px.ctypes.data_as(_fp), py.ctypes.data_as(_fp),
                                               o.ctypes.data_as(_fp), d.ctypes.data_as(_fp)))
        return o, d

    def closest_hit(self, origins, dirs) -> np.ndarray:
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        hits = np.zeros(o.shape[0], dtype=capi.HIT_DTYPE)
        self._check(capi.lib().prt_closest_hit(self._ctx, o.shape[0], o.ctypes.data_as(_fp), d.ctypes.data_as(_fp),
                                               hits.ctypes.data_as(C.POINTER(PrtHit))))
        return hits

    def scatter(self, in_dirs, hits: np.ndarray, rng_state):
        d = _f32(in_dirs).reshape(-1, 3)
        n = d.shape[0]
        hits = np.ascontiguousarray(hits)
        rng = np.ascontiguousarray(rng_state, dtype=np.uint32).copy()
        sc = np.zeros(n, np.uint32)
        att, em, oo, od = (np.zeros((n, 3), np.float32) for _ in range(4))
        self._check(capi.lib().prt_scatter(self._ctx, n, d.ctypes.data_as(_fp), hits.ctypes.data_as(C.POINTER(PrtHit)),
                                           rng.ctypes.data_as(_u32p), sc.ctypes.data_as(_u32p),
                                           att.ctypes.data_as(_fp), em.ctypes.data_as(_fp), oo.ctypes.data_as(_fp),
                                           od.ctypes.data_as(_fp)))
        return sc.astype(bool), att, em, oo, od, rng

    def enable_timing(self, on: bool = True):
        self._check(capi.lib().prt_enable_timing(self._ctx, int(on)))

    def stats(self) -> PrtStats:
        s = PrtStats()
        self._check(capi.lib().prt_get_stats(self._ctx, C.byref(s)))
        return s

    def reset_stats(self):
        self._check(capi.lib().prt_reset_stats(self._ctx))

    def measure_traversal(self, sample: int = 0) -> PrtStats:
        s = PrtStats()
        self._check(capi.lib().prt_measure_traversal(self._ctx, self.max_depth, self.seed, sample, C.byref(s)))
        return s

    def bvh_info(self) -> PrtBvhInfo:
        b = PrtBvhInfo()
        self._check(capi.lib().prt_bvh_info(self._ctx, C.byref(b)))
        return b

    def kernel_occupancy(self):
        o = capi.PrtOccupancy()
        self._check(capi.lib().prt_kernel_occupancy(self._ctx, C.byref(o)))
        return o

    def Refit(self, scene: "Scene"):
        """prt_refit_meshes: `scene`'s world-space meshes carry new vertex positions / normals over the topology the
        renderer was initialised with; the 8-wide tree is refitted on the device (no rebuild)."""
        d = scene.desc()
        scene._keep = d
        self._check(capi.lib().prt_refit_meshes(self._ctx, d.meshes, d.n_meshes))
        self._scene = scene

    def measure_shade_divergence(self, sample: int = 0) -> np.ndarray:
        """prt_measure_shade_divergence: [max_depth, 16] counters of the material mix per wave of the shade kernel."""
        out = np.zeros((self.max_depth, 16), np.uint64)
        self._check(capi.lib().prt_measure_shade_divergence(self._ctx, self.max_depth, self.seed, int(sample),
                                                            out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def kernel_instance(self) -> str:
        """Name of the traversal kernel instance the scene and tunables select (prt_kernel_instance)."""
        buf = C.create_string_buffer(64)
        self._check(capi.lib().prt_kernel_instance(self._ctx, buf, 64))
        return buf.value.decode()

    def bvh_read(self):
        b = self.bvh_info()
        nodes = np.zeros((b.n_nodes, 16), np.float32)
        tris = np.zeros((b.n_triangles, 12), np.float32)
        self._check(capi.lib().prt_bvh_read(self._ctx, nodes.ctypes.data_as(_fp), tris.ctypes.data_as(_fp)))
        return nodes, tris

    def bvh_read4(self) -> np.ndarray:
        b = self.bvh_info()
        nodes4 = np.zeros((b.n_nodes4, 32), np.float32)
        self._check(capi.lib().prt_bvh_read4(self._ctx, nodes4.ctypes.data_as(_fp)))
        return nodes4

    def bvh_read8(self) -> np.ndarray:
        """The compressed 8-wide tree: [n_nodes8, 20] uint32 (layout: csrc/bvh.h)."""
        b = self.bvh_info()
        nodes8 = np.zeros((b.n_nodes8, 20), np.uint32)
        self._check(capi.lib().prt_bvh_read8(self._ctx, nodes8.ctypes.data_as(C.POINTER(C.c_uint32))))
        return nodes8

    def set_scene_host_only(self, scene: Scene):
        """For host-only contexts (device < 0): build the BVH without a GPU."""
        d = scene.desc()
        self._check(capi.lib().prt_set_scene(self._ctx, C.byref(d)))


class HipWavefrontGroupRenderer:
    """Several GPUs of one node behind the Renderer interface: the binding of the C multi-GPU host path
    (include/prt.h prt_group_*; C++ form: host/prt_renderer.hpp HipWavefrontRenderer(devices)).  One context and one host
    thread per entry of `devices`, image tiled over them, one gather per ProgressiveRender call to devices[0] (RCCL over
    xGMI, or peer copies when a device appears more than once).  Everything happens inside libprt.so."""

    def __init__(self, devices, max_depth: int = DEFAULT_MAX_DEPTH, seed: int = 0):
        self._grp = C.c_void_p()
        L = capi.lib()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = L.prt_group_create(devs, len(devices), C.byref(self._grp))
        if rc:
            msg = L.prt_group_last_error(self._grp).decode()
            L.prt_group_destroy(self._grp)
            self._grp = None
            raise PrtError(f"prt_group_create({list(devices)}) failed: {msg}")
        self.max_depth = int(max_depth)
        self.seed = int(seed)
        self.frame_index = 0
        self.film: Optional[Film] = None

    def __del__(self):
        if getattr(self, "_grp", None):
            capi.lib().prt_group_destroy(self._grp)
            self._grp = None

    def _check(self, rc: int):
        if rc:
            raise PrtError(capi.lib().prt_group_last_error(self._grp).decode())

    @property
    def transport(self) -> str:
        return capi.lib().prt_group_transport(self._grp).decode()

    @property
    def n_devices(self) -> int:
        return capi.lib().prt_group_size(self._grp)

    def Init(self, film: Film, scene: Scene, camera: Camera):
        L = capi.lib()
        d = scene.desc()
        self._check(L.prt_group_set_scene(self._grp, C.byref(d)))
        self._check(L.prt_group_set_film(self._grp, film.width, film.height))
        self.film = film
        self.frame_index = 0
        self.SetCamera(camera)

    def SetCamera(self, camera: Camera):
        d = camera.desc()
        self._check(capi.lib().prt_group_set_camera(self._grp, C.byref(d)))

    def Refit(self, scene: Scene):
        """prt_group_refit_meshes: every rank refits its copy of the tree to `scene`'s deformed meshes (same topology)."""
        d = scene.desc()
        scene._keep = d
        self._check(capi.lib().prt_group_refit_meshes(self._grp, d.meshes, d.n_meshes))

    def ProgressiveRender(self, spp: int = 1):
        self._check(capi.lib().prt_group_render(self._grp, spp, self.max_depth, self.seed, self.frame_index))
        self.frame_index += spp

    def Clear(self):
        self._check(capi.lib().prt_group_film_clear(self._grp))
        self.frame_index = 0

    def set_samples_in_flight(self, n: int):
        self._check(capi.lib().prt_group_set_samples_in_flight(self._grp, n))

    def set_param(self, name: str, value: int):
        self._check(capi.lib().prt_group_set_param(self._grp, name.encode(), int(value)))

    def set_sampling(self, jitter: int = 0, rr_depth: int = 0, clamp: float = 0.0) -> PrtSampling:
        sp = PrtSampling(int(jitter), int(rr_depth), float(clamp))
        self._check(capi.lib().prt_group_set_sampling(self._grp, C.byref(sp)))
        return sp

    def download(self) -> Film:
        f = self.film
        self._check(capi.lib().prt_group_film_read(self._grp, f.accum.ctypes.data_as(_fp), f.weights.ctypes.data_as(_fp)))
        return f

    def UpdateDisplay(self, exposure: float = 1.0, gamma: float = 2.2) -> np.ndarray:
        f = self.film
        self._check(capi.lib().prt_group_film_display(self._grp, exposure, gamma, f.display.ctypes.data_as(C.POINTER(C.c_uint8))))
        return f.display

    def stats(self) -> PrtStats:
        s = PrtStats()
        self._check(capi.lib().prt_group_get_stats(self._grp, C.byref(s)))
        return s


def write_ppm(path: str, rgba8: np.ndarray):
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if capi.lib().prt_write_ppm(path.encode(), a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[1], a.shape[0]):
        raise PrtError(f"cannot write {path}")


def write_pfm(path: str, rgb: np.ndarray):
    a = _f32(rgb)
    if capi.lib().prt_write_pfm(path.encode(), a.ctypes.data_as(_fp), a.shape[1], a.shape[0]):
        raise PrtError(f"cannot write {path}")
