#!/usr/bin/env python3
"""Closest hits on a FULL-SIZE config scene against the oracle's BRUTE-FORCE scan over every triangle (the reference's
PrimitiveList::Intersect, primitive.cpp:21-59; no tree on the checker's side): camera rays, diffuse bounce rays leaving the
surface, and the awkward families of tests/fuzz_parity.py --rays.   python tests/bruteforce_fullsize.py --config C3 --n 8192"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import util  # noqa: E402
from util import prt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--gpu-build", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(17)
scene, cam, W, H, _, depth = prt.scenes.config(a.config)
r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=0)
if a.gpu_build:
    r.set_param("gpu_build", a.gpu_build)
r.Init(prt.Film(W, H), scene, cam)
n = a.n
q = n // 4
# (1) camera rays of random pixels
px = rng.uniform(0, W, q).astype(np.float32)
py = rng.uniform(0, H, q).astype(np.float32)
o1, d1 = r.camera_rays(px, py)
# (2) bounce rays: from the hit points of (1), cosine-ish directions about the normal
h1 = r.closest_hit(o1, d1)
ok = h1["prim"] != (0xFFFFFFFF if h1["prim"].dtype.kind == "u" else -1)
o2 = h1["position"][ok][:q]
nn = h1["normal"][ok][:q]
v = rng.normal(size=o2.shape).astype(np.float32)
v /= np.linalg.norm(v, axis=1, keepdims=True)
d2 = (nn + v).astype(np.float32)
d2 = np.stack([prt.glm_normalize(x) if np.any(x != 0) else np.array([0, 1, 0], np.float32) for x in d2]).astype(np.float32)
# (3) awkward: zero components, axis-parallel, lattice origins, far origins
o3 = rng.uniform(-2, 2, size=(q, 3)).astype(np.float32)
d3 = rng.normal(size=(q, 3)).astype(np.float32)
d3[: q // 3, int(rng.integers(0, 3))] = 0.0
d3[q // 3: 2 * q // 3] = 0.0
d3[q // 3: 2 * q // 3, int(rng.integers(0, 3))] = 1.0
o3[2 * q // 3:] = np.round(o3[2 * q // 3:] * 4) / 4
d3 = np.stack([prt.glm_normalize(x) for x in d3]).astype(np.float32)
o4 = (rng.normal(size=(q, 3)) * 60).astype(np.float32)
d4 = (-o4 + rng.normal(size=(q, 3)).astype(np.float32) * 0.7).astype(np.float32)
d4 = np.stack([prt.glm_normalize(x) for x in d4]).astype(np.float32)
o = np.concatenate([o1, o2, o3, o4]).astype(np.float32)
d = np.concatenate([d1, d2, d3, d4]).astype(np.float32)
got = r.closest_hit(o, d)
t0 = time.time()
want = util.oracle_scene(scene).closest_hit(o, d, use_bvh=False, n_threads=16)
bad = util.hits_equal(got, want)
nh = int((want["prim"] != (0xFFFFFFFF if want["prim"].dtype.kind == "u" else -1)).sum())
print(f"{a.config} ({scene.n_triangles} triangles, gpu_build {a.gpu_build}): {len(o)} rays ({len(o1)} camera, {len(o2)} bounce, {len(o3)} awkward, {len(o4)} far), "
      f"{nh} hit, brute force took {time.time() - t0:.0f} s: {'bit-exact' if not bad else 'MISMATCH ' + str(bad)}", flush=True)
sys.exit(1 if bad else 0)
