"""Generates tests/golden/oracle_golden.npz.

These vectors come from THIS repo's CPU oracle (oracle/prt_oracle.cpp), not from the reference: the reference
cannot be built or run in this image (no glm / tinyply / CUDA headers) and holds no fixtures of its own, so
parity stays UNPINNED.  The file pins the oracle (and through the GPU tests, the HIP path) against silent
drift between rounds.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import util  # noqa: E402
from util import orc, prt  # noqa: E402


def rays_for(seed, n, cam):
    rng = np.random.default_rng(seed)
    o1, d1 = util.random_rays(rng, n // 2, center=(0, 1, 0), radius=14.0, spread=6.0)
    px = rng.uniform(0, cam.width, n - n // 2).astype(np.float32)
    py = rng.uniform(0, cam.height, n - n // 2).astype(np.float32)
    o2, d2 = orc.camera_rays(cam.desc(), px, py)
    return np.concatenate([o1, o2]), np.concatenate([d1, d2])


def compute():
    out = {}
    cam = prt.Camera(width=64, height=48)
    # (1) function level: camera rays on a pixel grid, RNG streams
    xs, ys = np.meshgrid(np.arange(0, 64, 7) + 0.5, np.arange(0, 48, 5) + 0.5)
    o, d = orc.camera_rays(cam.desc(), xs.ravel(), ys.ravel())
    out["camera_dirs"] = d
    out["random_stream"] = orc.random_floats(orc.path_seed(1234, 5, 0), 64)[0]
    out["unit_vectors"] = np.stack([orc.random_unit_vector(s)[0] for s in range(32)])
    # (2) scene level: closest-hit records on the presets
    for name in util.PRESETS:
        scene = prt.Scene(name)
        o, d = rays_for(len(name), 512, cam)
        h = util.oracle_scene(scene).closest_hit(o, d)
        out[f"hit_{name}_prim"] = h["prim"]
        out[f"hit_{name}_d2"] = h["d2"]
        out[f"hit_{name}_normal"] = h["normal"]
        out[f"rays_{name}_o"] = o
        out[f"rays_{name}_d"] = d
    # (3) image level (small): C1 at 64x64, DEFAULT 48x27x4spp, both path forms
    for name, W, H, spp, depth in [("CORNELL", 64, 64, 1, 2), ("DEFAULT", 48, 27, 4, 5), ("MATERIAL_TEST", 48, 27, 2, 8)]:
        osc = util.oracle_scene(prt.Scene(name))
        c = prt.Camera(width=W, height=H).desc()
        for it in (0, 1):
            a, w, rays = osc.render(c, W, H, spp=spp, max_depth=depth, seed=0, iterative=bool(it), n_threads=8)
            out[f"img_{name}_{'iter' if it else 'rec'}"] = a
            out[f"img_{name}_{'iter' if it else 'rec'}_rays"] = np.int64(rays)
    # (4) triangle path: icosahedron refined to 320 triangles, brute force
    mesh = prt.Mesh(prt.scenes.asset("icosahedron.ply")).refine(320)
    scene = prt.scenes.mesh_scene(mesh)
    rng = np.random.default_rng(4)
    o, d = util.random_rays(rng, 512, center=(0, 0, 0), radius=6.0, spread=1.0)
    h = util.oracle_scene(scene).closest_hit(o, d, use_bvh=False)
    out["tri_rays_o"], out["tri_rays_d"] = o, d
    out["tri_hit_prim"], out["tri_hit_d2"], out["tri_hit_normal"] = h["prim"], h["d2"], h["normal"]
    c = prt.Camera(position=(2.0, 2.0, 3.0), width=32, height=32)
    a, w, rays = util.oracle_scene(scene).render(c.desc(), 32, 32, spp=2, max_depth=4, seed=1, iterative=True, n_threads=8)
    out["tri_img_iter"] = a
    return out


if __name__ == "__main__":
    data = compute()
    path = os.path.join(HERE, "oracle_golden.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes,", len(data), "arrays")
