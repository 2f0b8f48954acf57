"""The benchmark / parity scenes of SURVEY.md §8(d) (C1..C5), built from the bundled PLY fixtures.

The reference has no triangle scene at all (Mesh is never constructed, src/core/mesh.cpp:23; no preset
adds a Triangle, src/core/scene.cpp:62-350), so these are this project's synthetic inputs: the mesh on a
20x20 Lambertian ground quad under one emissive quad, sky (0.4,0.3,0.6), camera (5,5,8) -> origin.
"""
from __future__ import annotations

import os

import numpy as np

from .renderer import Camera, Mesh, Scene, make_transform

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets", "models")


def asset(name: str) -> str:
    return os.path.normpath(os.path.join(ASSETS, name))


def mesh_scene(mesh: Mesh, mesh_albedo=(0.8, 0.8, 0.8)) -> Scene:
    sc = Scene(preset=None)
    ground = sc.AddLambertian((0.5, 0.5, 0.5))
    light = sc.AddEmissive((15.0, 15.0, 15.0))
    body = sc.AddLambertian(mesh_albedo)
    sc.AddQuad(20.0, 20.0, ground, translation=(0.0, -1.0, 0.0))
    sc.AddQuad(4.0, 4.0, light, euler_deg=(180.0, 0.0, 0.0), translation=(0.0, 5.0, 0.0))
    sc.AddMesh(mesh, body)
    return sc


def refined(ply: str, target_triangles: int) -> Mesh:
    m = Mesh(asset(ply))
    if target_triangles > m.n_triangles:
        m.refine(target_triangles)
    return m


def triangulate_quads(base: Scene) -> Scene:
    """Every analytic quad of `base` becomes a 2x2-cell (8-triangle) world-space mesh baked with the quad's
    transform; circles stay analytic.  Used to cross-check Triangle::Intersect against Quad::Intersect."""
    sc = Scene(preset=None, sky=base.sky)
    sc.materials = list(base.materials)
    for p in base.primitives:
        if p.shape_type != 1:
            sc.primitives.append(p)
            continue
        w, h = p.shape_param[0], p.shape_param[1]
        xs = np.linspace(-w / 2, w / 2, 3, dtype=np.float32)
        zs = np.linspace(-h / 2, h / 2, 3, dtype=np.float32)
        verts = np.array([[x, 0.0, z] for z in zs for x in xs], np.float32)
        nrm = np.tile(np.array([[0.0, 1.0, 0.0]], np.float32), (9, 1))
        idx = []
        for j in range(2):
            for i in range(2):
                a, b, c, d = j * 3 + i, j * 3 + i + 1, (j + 1) * 3 + i, (j + 1) * 3 + i + 1
                idx += [[a, c, b], [b, c, d]]
        m = Mesh(vertices=verts, normals=nrm, indices=np.array(idx, np.uint32))
        m.transform(np.array(p.mat[:], np.float32), np.array(p.inv[:], np.float32))
        sc.AddMesh(m, p.material_id)
    return sc


def cornell_triangulated() -> Scene:
    """C1's 32-triangle tessellation of the CORNELL preset (src/core/scene.cpp:332-350)."""
    return triangulate_quads(Scene("CORNELL"))


# Camera of the mesh configs.  SURVEY §8d kept main()'s (5,5,8) -> origin (src/main.cpp:142-150), from where a
# [-1,1]^3 mesh covers ~4 % of a 1080p frame and the metric would measure ground-quad hits, not BVH
# traversal; the mesh configs therefore look at the mesh from ~2.3 units so it fills most of the frame height.
MESH_CAMERA = (1.2, 0.4, 1.9)
GRID_CAMERA = (0.0, 3.0, 7.5)


def config(name: str):
    """Returns (scene, camera, width, height, spp, max_depth) for C1..C5.  'B bounces' = B+1 segments."""
    name = name.upper()
    if name == "C1":
        return Scene("CORNELL"), Camera(width=256, height=256), 256, 256, 1, 2
    if name == "C1T":
        return cornell_triangulated(), Camera(width=256, height=256), 256, 256, 1, 2
    if name == "C2":
        return mesh_scene(refined("bunny.ply", 70_000)), Camera(MESH_CAMERA, width=1280, height=720), 1280, 720, 64, 5
    if name == "C3":
        return mesh_scene(refined("dragon.ply", 870_000)), Camera(MESH_CAMERA, width=1920, height=1080), 1920, 1080, 256, 5
    if name == "C4":
        return mesh_scene(refined("dragon.ply", 870_000)), Camera(MESH_CAMERA, width=3840, height=2160), 3840, 2160, 256, 9
    if name == "C5":
        base = refined("dragon.ply", 870_000)
        big = None
        for gz in range(3):
            for gx in range(4):
                inst = base.copy()
                mat, inv = make_transform((1, 1, 1), (0, 0, 0), ((gx - 1.5) * 2.2, 0.0, (gz - 1.0) * 2.2))
                inst.transform(mat, inv)
                big = inst if big is None else big.append(inst)
        return mesh_scene(big), Camera(GRID_CAMERA, width=1920, height=1080), 1920, 1080, 1024, 9
    if name == "C5I":  # the same 12 dragons as placed copies of ONE mesh (PrtInstance, two-level BVH)
        base = refined("dragon.ply", 870_000)
        sc = Scene(preset=None)
        ground = sc.AddLambertian((0.5, 0.5, 0.5))
        light = sc.AddEmissive((15.0, 15.0, 15.0))
        body = sc.AddLambertian((0.8, 0.8, 0.8))
        sc.AddQuad(20.0, 20.0, ground, translation=(0.0, -1.0, 0.0))
        sc.AddQuad(4.0, 4.0, light, euler_deg=(180.0, 0.0, 0.0), translation=(0.0, 5.0, 0.0))
        for gz in range(3):
            for gx in range(4):
                sc.AddInstance(base, body, translation=((gx - 1.5) * 2.2, 0.0, (gz - 1.0) * 2.2))
        return sc, Camera(GRID_CAMERA, width=1920, height=1080), 1920, 1080, 1024, 9
    raise ValueError(f"unknown config {name}")
