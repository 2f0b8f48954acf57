#!/usr/bin/env python3
"""Worst case for box culling against the reference's Triangle::Intersect rounding: rays that graze a triangle's plane,
travel (nearly) along one of its edges and cross the plane a few rounding errors outside that edge, on meshes whose
leaf boxes are tight there (an axis-aligned planar grid) and on a general mesh.  GPU closest hit vs the oracle's
brute-force scan.   python tests/graze_probe.py [--n 400000]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import util  # noqa: E402
from util import prt  # noqa: E402

class _Args:
    n, gpu_build, pad_log2, cos_lo, cos_hi = 400_000, 0, 0, 1e-5, 0.08


a = _Args()
rng = np.random.default_rng(3)


def grid_mesh(n=24, size=4.0, tilt=None):
    xs = np.linspace(-size / 2, size / 2, n + 1)
    V = np.array([(x, 0.0, z) for z in xs for x in xs], dtype=np.float64)
    if tilt is not None:
        V = V @ tilt.T
    idx = []
    for j in range(n):
        for i in range(n):
            p = j * (n + 1) + i
            idx += [(p, p + 1, p + n + 2), (p, p + n + 2, p + n + 1)]
    N = np.tile(np.array([0.0, 1.0, 0.0]) if tilt is None else tilt @ np.array([0.0, 1.0, 0.0]), (len(V), 1))
    return V.astype(np.float32), N.astype(np.float32), np.array(idx, dtype=np.uint32)


def probe(name, V, N, I):
    mesh = prt.Mesh(vertices=V, normals=N, indices=I)
    sc = prt.Scene(preset=None)
    mat = sc.AddLambertian((0.7, 0.7, 0.7))
    sc.AddMesh(mesh, mat)
    r = prt.HipWavefrontRenderer(device=0, max_depth=2, seed=0)
    if a.gpu_build:
        r.set_param("gpu_build", a.gpu_build)
    if a.pad_log2:
        r.set_param("pad_log2", a.pad_log2)
    r.Init(prt.Film(16, 16), sc, prt.Camera(position=(5.0, 5.0, 8.0), width=16, height=16))
    n = a.n
    T = I[rng.integers(0, len(I), n)]
    e = rng.integers(0, 3, n)
    P0 = V[T[np.arange(n), e]].astype(np.float64)
    P1 = V[T[np.arange(n), (e + 1) % 3]].astype(np.float64)
    P2 = V[T[np.arange(n), (e + 2) % 3]].astype(np.float64)
    edge = P1 - P0
    elen = np.linalg.norm(edge, axis=1, keepdims=True)
    eu = edge / elen
    nrm = np.cross(edge, P2 - P0)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    outw = np.cross(eu, nrm)                                  # in-plane, perpendicular to the edge
    outw *= -np.sign((outw * (P2 - P0)).sum(1, keepdims=True))  # pointing away from the third vertex
    dist = np.exp(rng.uniform(np.log(0.5), np.log(40.0), (n, 1)))
    cosi = np.exp(rng.uniform(np.log(a.cos_lo), np.log(a.cos_hi), (n, 1)))  # cosine of the incidence angle (grazing)
    psi = rng.uniform(-0.08, 0.08, (n, 1))                    # in-plane deviation from the edge direction
    err = 3 * 6e-8 * dist / cosi                              # the lateral size of the rounding effect
    off = rng.uniform(-1.0, 3.0, (n, 1)) * err                # crossing point: from inside the triangle to 3 errors outside
    cross_pt = P0 + eu * (rng.uniform(0.1, 0.9, (n, 1)) * elen) + outw * off
    g = eu * np.cos(psi) + outw * np.sin(psi)
    d = g * np.sqrt(1 - cosi ** 2) - nrm * cosi * rng.choice([-1.0, 1.0], (n, 1))
    o = (cross_pt - d * dist).astype(np.float32)
    d = np.stack([prt.glm_normalize(x) for x in d.astype(np.float32)]).astype(np.float32)
    got = r.closest_hit(o, d)
    want = util.oracle_scene(sc).closest_hit(o, d, use_bvh=False, n_threads=16)
    bad = np.nonzero((got["prim"] != want["prim"]) | ~((got["d2"] == want["d2"]) | (np.isnan(got["d2"]) & np.isnan(want["d2"]))))[0]
    print(f"{name}: {len(I)} triangles, {n} grazing rays, {int((want['prim'] >= 0).sum()) if want['prim'].dtype.kind == 'i' else int((want['prim'] != 0xFFFFFFFF).sum())} hit, {len(bad)} differ", flush=True)
    edges = [1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 0.08]
    hb, _ = np.histogram(cosi[bad, 0], bins=edges)
    ha, _ = np.histogram(cosi[:, 0], bins=edges)
    print("   differing rays by cosine of incidence: " + ", ".join(f"[{edges[j]:.0e},{edges[j + 1]:.0e}): {hb[j]} of {ha[j]}" for j in range(len(hb))), flush=True)
    for k in bad[:5]:
        print(f"   ray {k}: dist {dist[k, 0]:.2f} cos {cosi[k, 0]:.2e} psi {psi[k, 0]:.3f} off/err {off[k, 0] / err[k, 0]:.2f}: got prim {got['prim'][k]} d2 {got['d2'][k]!r}, want prim {want['prim'][k]} d2 {want['d2'][k]!r}")
    return len(bad)


def tilted_grid():
    th = np.radians(33.0)
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]]) @ np.array([[1, 0, 0], [0, np.cos(0.4), -np.sin(0.4)], [0, np.sin(0.4), np.cos(0.4)]])
    return grid_mesh(tilt=R)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=400_000)
    ap.add_argument("--gpu-build", type=int, default=0)
    ap.add_argument("--pad-log2", type=int, default=0, help="culling pad 2^-n (prt_set_param pad_log2); 0 = the default")
    ap.add_argument("--cos-lo", type=float, default=1e-5)
    ap.add_argument("--cos-hi", type=float, default=0.08)
    ns = ap.parse_args()
    a.n, a.gpu_build, a.pad_log2, a.cos_lo, a.cos_hi = ns.n, ns.gpu_build, ns.pad_log2, ns.cos_lo, ns.cos_hi
    total = probe("axis-aligned planar grid", *grid_mesh())
    total += probe("tilted planar grid", *tilted_grid())
    m = prt.scenes.refined("bunny.ply", 12_000)
    total += probe("bunny 12 k", m.GetVertices(), m.GetNormals(), m.GetIndices())
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
