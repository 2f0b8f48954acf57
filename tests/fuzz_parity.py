#!/usr/bin/env python3
"""Randomized parity: random scenes (analytic primitives of every material, refined meshes, placed copies), random cameras,
builders, kernel tunables and sampling flags; every frame bit for bit against the oracle (throughput form) + ray counts.
  python tests/fuzz_parity.py --cases 200 --seed 1      (GPU box; ~0.3 s per case)
Test infrastructure (it calls the oracle): a checker, not a product path; tests/test_gpu_fuzz.py runs a seeded subset."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parallelraytracing_amd as prt  # noqa: E402
from oracle import oracle as orc  # noqa: E402

MESHES = {}


def mesh(name, tris):
    key = (name, tris)
    if key not in MESHES:
        m = prt.Mesh(prt.scenes.asset(name))
        MESHES[key] = m.refine(tris) if tris else m
    return MESHES[key]


def random_scene(rng):
    sc = prt.Scene(preset=None)
    mats = [sc.AddLambertian(tuple(rng.uniform(0.2, 0.9, 3))), sc.AddLambertian(tuple(rng.uniform(0.2, 0.9, 3))),
            sc.AddMetal(tuple(rng.uniform(0.5, 1.0, 3)), float(rng.choice([0.0, 0.05, 0.3]))),
            sc.AddDielectric(float(rng.choice([1.3, 1.5, 2.4]))), sc.AddEmissive(tuple(rng.uniform(2, 12, 3)))]
    pick = lambda: mats[int(rng.integers(0, len(mats)))]  # noqa: E731
    desc = []
    if rng.random() < 0.8:
        sc.AddQuad(float(rng.uniform(8, 40)), float(rng.uniform(8, 40)), mats[0], translation=(0.0, float(rng.uniform(-2.0, -0.5)), 0.0))
    if rng.random() < 0.7:
        sc.AddQuad(3.0, 3.0, mats[4], euler_deg=(180.0, 0.0, 0.0), translation=(float(rng.uniform(-2, 2)), float(rng.uniform(4, 8)), float(rng.uniform(-2, 2))))
    n_prims = int(rng.choice([0, 1, 3, 8, 20, 40]))
    for _ in range(n_prims):
        tr = tuple(float(v) for v in rng.uniform(-5, 5, 3))
        if rng.random() < 0.6:
            s = float(rng.uniform(0.3, 1.5))
            sc.AddCircle(float(rng.uniform(0.2, 1.2)), pick(), scale=(s, s, s), translation=tr)
        else:
            sc.AddQuad(float(rng.uniform(0.5, 3)), float(rng.uniform(0.5, 3)), pick(),
                       euler_deg=(float(rng.choice([0, 50, 90, 180])), 0.0, 0.0), translation=tr)
    desc.append(f"{n_prims} prims")
    kind = rng.choice(["none", "world", "inst", "both"], p=[0.1, 0.4, 0.25, 0.25])
    name, tris = [("icosahedron.ply", 0), ("icosahedron.ply", 300), ("bunny.ply", 0), ("bunny.ply", 4000), ("hand.ply", 0),
                  ("dragon.ply", 0)][int(rng.integers(0, 6))]
    world = None
    if kind in ("world", "both"):
        world = mesh(name, tris)
        sc.AddMesh(world, pick())
        desc.append(f"world {name}:{tris}")
    if kind in ("inst", "both"):
        n_inst = int(rng.integers(1, 7))
        name2, tris2 = [("icosahedron.ply", 0), ("bunny.ply", 0), ("icosahedron.ply", 1200)][int(rng.integers(0, 3))]
        for _ in range(n_inst):
            sc.AddInstance(mesh(name2, tris2), pick(), scale=float(rng.uniform(0.3, 2.5)),
                           euler_deg=tuple(float(v) for v in rng.uniform(-180, 180, 3)),
                           translation=tuple(float(v) for v in rng.uniform(-5, 5, 3)))
        desc.append(f"{n_inst} copies of {name2}:{tris2}")
    return sc, ", ".join(desc), kind, world


def run_case(case, seed):
    rng = np.random.default_rng([seed, case])
    scene, desc, kind, _ = random_scene(rng)
    W, H = int(rng.choice([17, 64, 96, 131])), int(rng.choice([9, 48, 72]))
    pos = rng.normal(size=3)
    pos = pos / np.linalg.norm(pos) * rng.uniform(3, 14)
    pos[1] = abs(pos[1]) + 0.3
    front = -pos + rng.uniform(-1, 1, 3)
    cam = prt.Camera(position=tuple(float(v) for v in pos), front=tuple(float(v) for v in front), width=W, height=H)
    depth = int(rng.integers(1, 9))
    spp = int(rng.integers(1, 4))
    rseed = int(rng.integers(0, 1 << 30))
    r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=rseed)
    params = {}
    has_mesh = kind != "none"
    if has_mesh and rng.random() < 0.5:
        params["gpu_build"] = int(rng.choice([1, 2]))
    if rng.random() < 0.3:
        params["node_stride"] = 8
    if rng.random() < 0.3:
        params["prim_bvh"] = 0
    if rng.random() < 0.3:
        params["compact_primary"] = 0
    if rng.random() < 0.2:
        params["primary_hit"] = 0
    if rng.random() < 0.4:
        params["path_kernel"] = int(rng.choice([0, 2]))  # (1 = the default: batches of one sample)
    if rng.random() < 0.3:
        params["fuse"] = int(rng.integers(0, 2))
    if rng.random() < 0.2:
        params["exact_grids"] = int(rng.integers(0, 3))
    if rng.random() < 0.2 and kind in ("world",):
        params["stack_cap"] = int(rng.integers(2, 6))  # forces the overflow list (host-built trees only: needs the 4-wide tree)
        params.pop("gpu_build", None)
    if rng.random() < 0.25:  # the big-grab tier of the ray hand-out, forced onto small launches
        params["big"] = int(rng.choice([2, 3, 5]))
        params["big_min"] = 1
        params["big_keep"] = int(rng.choice([0, 1, 4]))
        params["chunk"] = int(rng.choice([64, 128, 256]))
    if rng.random() < 0.3:
        params["static_small"] = int(rng.choice([0, 2, 64]))  # granules dealt round-robin / through the cursor
    if rng.random() < 0.2:
        params["refill_min"] = int(rng.choice([1, 8, 32, 64]))
    if rng.random() < 0.2:
        params["tri_min"] = int(rng.choice([1, 8, 64]))
    for k, v in params.items():
        r.set_param(k, v)
    film = prt.Film(W, H)
    try:
        r.Init(film, scene, cam)
    except prt.PrtError as e:  # e.g. a tree too deep for the two-level kernel: must be a clean error
        return f"case {case}: Init refused ({str(e)[:80]}) [{desc}]", True
    sp = None
    if rng.random() < 0.4:
        sp = r.set_sampling(jitter=int(rng.integers(0, 2)), rr_depth=int(rng.choice([0, 1, 3])), clamp=float(rng.choice([0.0, 1.5])))
    sif = int(rng.integers(1, spp + 1))
    r.set_samples_in_flight(sif)
    r.ProgressiveRender(spp)
    r.download()
    osc = orc.OracleScene(scene.desc())
    # small meshes: the oracle scans linearly as the reference does (its own BVH only for the big ones, to stay fast)
    acc, wts, rays = osc.render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=rseed, iterative=True,
                                use_bvh=scene.n_triangles > 6000, n_threads=8, sampling=sp)
    st = r.stats()
    nbad = int((film.accum != acc).any(axis=-1).sum())
    ok = nbad == 0 and np.array_equal(film.weights, wts) and st.rays_total == rays
    msg = (f"case {case}: {'ok ' if ok else 'MISMATCH'} {W}x{H} spp {spp} depth {depth} [{desc}] params {params} sampling "
           f"{(sp.jitter, sp.rr_depth, sp.clamp) if sp else None}: {nbad} bad pixels, rays {st.rays_total} vs {rays}")
    return msg, ok


def run_ray_case(case, seed, n=4096):
    """Closest hit of awkward rays against the oracle's BRUTE-FORCE scan: directions with zero components (infinite slab
    reciprocals), axis-parallel rays inside box planes, origins on vertices / inside the mesh / far away, near-degenerate
    directions; a random scene and builder as above."""
    rng = np.random.default_rng([seed, case, 77])
    scene, desc, kind, world = random_scene(rng)
    r = prt.HipWavefrontRenderer(device=0, max_depth=2, seed=0)
    params = {}
    if kind != "none" and rng.random() < 0.5:
        params["gpu_build"] = int(rng.choice([1, 2]))
    if rng.random() < 0.3:
        params["node_stride"] = 8
    if rng.random() < 0.3:
        params["prim_bvh"] = 0
    for k, v in params.items():
        r.set_param(k, v)
    film = prt.Film(16, 16)
    r.Init(film, scene, prt.Camera(position=(5.0, 5.0, 8.0), width=16, height=16))
    o = rng.uniform(-6, 6, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    q = n // 8
    d[0:q, int(rng.integers(0, 3))] = 0.0                      # one zero component
    ax = int(rng.integers(0, 3))
    d[q:2 * q] = 0.0
    d[q:2 * q, ax] = rng.choice([-1.0, 1.0], size=q)           # axis-parallel
    o[2 * q:3 * q] = np.round(o[2 * q:3 * q] * 2) / 2           # origins on a coarse lattice (often on box planes / quads)
    d[3 * q:4 * q] *= np.float32(1e-3)                          # short direction vectors (normalised below like any other)
    d[4 * q:5 * q, 1] = rng.choice([1e-9, -1e-9, 1e-7], size=q).astype(np.float32)  # grazing the y = const quads
    o[5 * q:6 * q] *= np.float32(50.0)                          # far away, aimed at the origin
    d[5 * q:6 * q] = -o[5 * q:6 * q] + rng.normal(size=(q, 3)).astype(np.float32)
    verts = world.GetVertices() if world is not None else None
    if verts is not None and len(verts):
        sel = verts[rng.integers(0, len(verts), size=q)]
        d[6 * q:7 * q] = sel - o[6 * q:7 * q]                   # aimed exactly at mesh vertices (if the mesh is in the scene)
    d = np.stack([prt.glm_normalize(v) if np.any(v != 0) else np.array([0, 0, 1], np.float32) for v in d]).astype(np.float32)
    got = r.closest_hit(o, d)
    want = orc.OracleScene(scene.desc()).closest_hit(o, d, use_bvh=False, n_threads=8)
    bad = []
    for f in ("prim", "front_face", "material_id", "d2", "position", "normal"):
        same = got[f] == want[f]
        if got[f].dtype.kind == "f":
            same = same | (np.isnan(got[f]) & np.isnan(want[f]))
        if not np.all(same):
            bad.append(f)
    ok = bad == []
    if not ok:
        idx = np.nonzero((got["prim"] != want["prim"]) | (got["d2"] != want["d2"]))[0]
        for i in idx[:6]:
            print(f"   ray {i}: o {o[i].tolist()} d {d[i].tolist()} got prim {got['prim'][i]} d2 {got['d2'][i]!r} front {got['front_face'][i]} "
                  f"want prim {want['prim'][i]} d2 {want['d2'][i]!r} front {want['front_face'][i]}", flush=True)
        print(f"   {len(idx)} rays differ; n_prims {len(scene.primitives)}", flush=True)
    nhit = int((got["prim"] != 0xFFFFFFFF).sum()) if got["prim"].dtype.kind == "u" else int((got["prim"] >= 0).sum())
    return f"ray case {case}: {'ok ' if ok else 'MISMATCH ' + str(bad)} [{desc}] params {params}: {nhit} of {n} hit", ok


def run_sequence_case(case, seed):
    """ONE long-lived renderer driven through a random sequence of the interface's calls (ProgressiveRender in chunks of
    any size, SetCamera, film Clear, sampling flags, samples in flight, run-time tunables, re-Init with another scene, a
    second film size, a group of 2-3 contexts on the same GPU): after every render the film must equal what the oracle
    accumulates for the same sequence (samples first_sample .. of the current camera / scene / flags added in order)."""
    rng = np.random.default_rng([seed, case, 4242])
    depth = int(rng.integers(1, 7))
    rseed = int(rng.integers(0, 1 << 30))
    use_group = rng.random() < 0.25
    k_ctx = int(rng.integers(2, 4))
    r = prt.HipWavefrontGroupRenderer([0] * k_ctx, max_depth=depth, seed=rseed) if use_group else \
        prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=rseed)
    log = [f"group x{k_ctx}" if use_group else "single"]
    state = {}

    def new_camera(W, H):
        pos = rng.normal(size=3)
        pos = pos / np.linalg.norm(pos) * rng.uniform(3, 12)
        pos[1] = abs(pos[1]) + 0.3
        return prt.Camera(position=tuple(float(v) for v in pos), front=tuple(float(v) for v in (-pos + rng.uniform(-1, 1, 3))), width=W, height=H)

    def init():
        scene, desc, kind, _ = random_scene(rng)
        W, H = int(rng.choice([17, 40, 96])), int(rng.choice([9, 33, 48]))
        film = prt.Film(W, H)
        cam = new_camera(W, H)
        if not use_group and kind != "none" and rng.random() < 0.4:
            r.set_param("gpu_build", int(rng.choice([0, 1, 2])))
        r.Init(film, scene, cam)
        state.update(scene=scene, osc=orc.OracleScene(scene.desc()), W=W, H=H, film=film, cam=cam, fi=0, sp=None,
                     acc=np.zeros((H, W, 3), np.float32), wts=np.zeros((H, W), np.float32), rays=0)
        r.set_sampling(0, 0, 0.0)
        log.append(f"Init {W}x{H} [{desc}]")

    init()
    n_ops = int(rng.integers(4, 14))
    for _ in range(n_ops):
        op = rng.choice(["render", "render", "render", "camera", "clear", "sampling", "sif", "param", "reinit"])
        if op == "render":
            k = int(rng.integers(1, 5))
            if use_group:
                r.frame_index = state["fi"]
            r.ProgressiveRender(k)
            st = state
            st["osc"].render(st["cam"].desc(), st["W"], st["H"], spp=k, first_sample=st["fi"], max_depth=depth, seed=rseed,
                             iterative=True, use_bvh=st["scene"].n_triangles > 6000, n_threads=8, accum=st["acc"],
                             weights=st["wts"], sampling=st["sp"])
            st["fi"] += k
            f = r.download()
            log.append(f"render {k}")
            if not (np.array_equal(f.accum, st["acc"]) and np.array_equal(f.weights, st["wts"])):
                nbad = int((f.accum != st["acc"]).any(axis=-1).sum())
                return f"sequence case {case}: MISMATCH after {log}: {nbad} bad pixels", False
        elif op == "camera":
            state["cam"] = new_camera(state["W"], state["H"])
            r.SetCamera(state["cam"])
            log.append("SetCamera")
        elif op == "clear":
            if use_group:
                r.Clear()
            else:
                state["film"].Clear()
                r.frame_index = 0
            state["fi"] = 0
            state["acc"][:] = 0
            state["wts"][:] = 0
            log.append("Clear")
        elif op == "sampling":
            state["sp"] = r.set_sampling(jitter=int(rng.integers(0, 2)), rr_depth=int(rng.choice([0, 1, 3])), clamp=float(rng.choice([0.0, 1.5])))
            if state["sp"].jitter == 0 and state["sp"].rr_depth == 0 and state["sp"].clamp == 0.0:
                state["sp"] = None
            log.append("sampling")
        elif op == "sif":
            r.set_samples_in_flight(int(rng.integers(1, 6)))
            log.append("sif")
        elif op == "param":
            name, val = [("fuse", int(rng.integers(0, 2))), ("exact_grids", int(rng.integers(0, 3))), ("refill_min", int(rng.choice([1, 16, 64]))),
                         ("tri_min", int(rng.choice([1, 24, 64]))), ("compact_primary", int(rng.integers(0, 2))), ("steal", int(rng.choice([0, 8]))),
                         ("primary_hit", int(rng.integers(0, 2))),
                         ("path_kernel", int(rng.integers(0, 3)))][int(rng.integers(0, 8))]
            r.set_param(name, val)
            log.append(f"{name}={val}")
        else:
            init()
    return f"sequence case {case}: ok  {log}", True


def run_graze_case(case, seed, n=100_000):
    """Spheres only, more than 16 of them (so the walk over their world boxes runs), rays aimed to graze them within a few
    rounding margins of the reference's discriminant (2e-7 * dist^2 / R), from 0.5 ... 2000 units away, random radii down
    to 0.01 and scales; against the oracle's linear scan."""
    rng = np.random.default_rng([seed, case, 991])
    sc = prt.Scene(preset=None)
    mat = sc.AddLambertian((0.7, 0.7, 0.7))
    k = int(rng.integers(17, 120))
    centres, radii = [], []
    spread = float(rng.choice([2.0, 8.0, 40.0]))
    for _ in range(k):
        c = rng.uniform(-spread, spread, 3)
        s = float(np.exp(rng.uniform(np.log(0.1), np.log(3.0))))
        r = float(np.exp(rng.uniform(np.log(0.01), np.log(1.0))))
        sc.AddCircle(r, mat, scale=(s, s, s), translation=tuple(float(v) for v in c))
        centres.append(c)
        radii.append(r * s)
    if rng.random() < 0.3:  # the classic ground: one huge sphere
        sc.AddCircle(1000.0, mat, translation=(0.0, -1000.0 - spread, 0.0))
    centres, radii = np.array(centres), np.array(radii)
    i = rng.integers(0, k, n)
    lo, hi = [(0.5, 30.0), (5.0, 300.0), (30.0, 2000.0)][int(rng.integers(0, 3))]
    dist = np.exp(rng.uniform(np.log(lo), np.log(hi), n)) + radii[i] * 1.01
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = rng.normal(size=(n, 3))
    v -= (v * u).sum(1, keepdims=True) * u
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    m = rng.uniform(-3e-7, 6e-7, n) * dist * dist / radii[i]
    ang = np.arcsin(np.clip((radii[i] + m) / dist, 0.0, 1.0))
    o = (centres[i] + u * dist[:, None]).astype(np.float32)
    d = (-u * np.cos(ang)[:, None] + v * np.sin(ang)[:, None]).astype(np.float32)
    inv = 1.0 / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)
    d = (d * inv[:, None]).astype(np.float32)  # (x * (1 / sqrt(dot)): glm's normalize, vectorised)
    r = prt.HipWavefrontRenderer(device=0, max_depth=2, seed=0)
    r.Init(prt.Film(16, 16), sc, prt.Camera(position=(5.0, 5.0, 8.0), width=16, height=16))
    got = r.closest_hit(o, d)
    want = orc.OracleScene(sc.desc()).closest_hit(o, d, use_bvh=False, n_threads=16)
    bad = []
    for f in ("prim", "front_face", "material_id", "d2", "position", "normal"):
        same = got[f] == want[f]
        if got[f].dtype.kind == "f":
            same = same | (np.isnan(got[f]) & np.isnan(want[f]))
        if not np.all(same):
            bad.append(f)
    oo, dd = o.astype(np.float64), d.astype(np.float64)
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    miss_by = np.linalg.norm(np.cross(oo - centres[i], dd), axis=1) - radii[i]
    phantom = int(((want["prim"] == i) & (miss_by > 1e-6 * dist)).sum())
    ok = bad == []
    return f"graze case {case}: {'ok ' if ok else 'MISMATCH ' + str(bad)} {k} spheres, spread {spread}, dist {lo}..{hi}: {phantom} phantom hits in {n} rays", ok


def run_graze_quads_case(case, seed, n=100_000, cos_lo=1e-5, cos_hi=0.08):
    """More than 16 quads (so the walk over the primitives' world boxes runs), rays that graze a quad's plane, travel along
    one of its edges and cross the plane within a few rounding errors (3u * dist / cos) of it; against the linear scan.
    Returns the message, ok, and the cosines of incidence of the rays that differ."""
    rng = np.random.default_rng([seed, case, 555])
    sc = prt.Scene(preset=None)
    mat = sc.AddLambertian((0.7, 0.7, 0.7))
    k = int(rng.integers(17, 60))
    quads = []
    for _ in range(k):
        c = rng.uniform(-6, 6, 3)
        w, h = float(rng.uniform(0.5, 4.0)), float(rng.uniform(0.5, 4.0))
        ex = float(rng.choice([0.0, 50.0, 90.0, 180.0, float(rng.uniform(0, 360))]))
        sc.AddQuad(w, h, mat, euler_deg=(ex, 0.0, 0.0), translation=tuple(float(v) for v in c))
        a = np.radians(ex)
        R = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        quads.append((c, w, h, R))
    i = rng.integers(0, k, n)
    C = np.array([quads[j][0] for j in i])
    Wd = np.array([quads[j][1] for j in i])
    Hd = np.array([quads[j][2] for j in i])
    Rm = np.array([quads[j][3] for j in i])
    ax = Rm[:, :, 0]          # local x in world
    nz = Rm[:, :, 2]          # local z in world
    nrm = Rm[:, :, 1]         # local y (the normal)
    along_x = rng.random(n) < 0.5
    eu = np.where(along_x[:, None], ax, nz)                     # edge direction
    outw = np.where(along_x[:, None], nz, ax) * rng.choice([-1.0, 1.0], (n, 1))
    half_out = np.where(along_x, Hd, Wd) * 0.5
    half_along = np.where(along_x, Wd, Hd) * 0.5
    dist = np.exp(rng.uniform(np.log(0.5), np.log(40.0), (n, 1)))
    cosi = np.exp(rng.uniform(np.log(cos_lo), np.log(cos_hi), (n, 1)))
    psi = rng.uniform(-0.08, 0.08, (n, 1))
    err = 3 * 6e-8 * dist / cosi
    off = rng.uniform(-1.0, 3.0, (n, 1)) * err
    cross_pt = C + eu * (rng.uniform(-0.8, 0.8, (n, 1)) * half_along[:, None]) + outw * (half_out[:, None] + off)
    g = eu * np.cos(psi) + outw * np.sin(psi)
    d = g * np.sqrt(1 - cosi ** 2) - nrm * cosi * rng.choice([-1.0, 1.0], (n, 1))
    o = (cross_pt - d * dist).astype(np.float32)
    d = d.astype(np.float32)
    inv = (1.0 / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])).astype(np.float32)
    d = (d * inv[:, None]).astype(np.float32)
    r = prt.HipWavefrontRenderer(device=0, max_depth=2, seed=0)
    r.Init(prt.Film(16, 16), sc, prt.Camera(position=(5.0, 5.0, 8.0), width=16, height=16))
    got = r.closest_hit(o, d)
    want = orc.OracleScene(sc.desc()).closest_hit(o, d, use_bvh=False, n_threads=16)
    diff = (got["prim"] != want["prim"]) | ~((got["d2"] == want["d2"]) | (np.isnan(got["d2"]) & np.isnan(want["d2"])))
    bad = np.nonzero(diff)[0]
    ok = len(bad) == 0
    extra = "" if ok else f"; cosines of the differing rays: min {cosi[bad, 0].min():.2e} max {cosi[bad, 0].max():.2e}"
    return f"quad graze case {case}: {'ok ' if ok else 'MISMATCH'} {k} quads, {len(bad)} of {n} rays differ{extra}", ok, cosi[bad, 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--sequences", action="store_true", help="random call sequences on one long-lived renderer (or a group of contexts)")
    ap.add_argument("--graze", action="store_true", help="rays grazing spheres within the reference's rounding margins (walk over the primitives' boxes)")
    ap.add_argument("--graze-quads", action="store_true", help="rays grazing quads along their edges (walk over the primitives' boxes)")
    ap.add_argument("--cos-lo", type=float, default=1e-5, help="--graze-quads: smallest cosine of incidence")
    ap.add_argument("--rays", action="store_true", help="closest-hit cases with awkward rays against the brute-force scan")
    a = ap.parse_args()
    t0 = time.time()
    bad = 0
    for case in range(a.first, a.first + a.cases):
        msg, ok = (run_ray_case(case, a.seed) if a.rays else run_sequence_case(case, a.seed) if a.sequences else
                   run_graze_case(case, a.seed) if a.graze else
                   run_graze_quads_case(case, a.seed, cos_lo=a.cos_lo)[:2] if a.graze_quads else run_case(case, a.seed))
        if not ok:
            bad += 1
        if a.verbose or not ok or "refused" in msg:
            print(msg, flush=True)
        if (case - a.first) % 25 == 24:
            print(f"... {case - a.first + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz_parity: {a.cases} cases from {a.first} (seed {a.seed}): {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
