"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on the same seeded
inputs.  Integer / index / hit results must be bit-exact; images are compared bit-exactly against the
oracle's throughput form (TraceRayGPU restatement) and within the north-star tolerance of 1e-4 per-pixel L2
against its recursive form (CPURenderer::TraceRay restatement)."""
import numpy as np
import pytest

import util
from util import orc, prt

pytestmark = pytest.mark.gpu

TOL_L2 = 1e-4  # BASELINE.json north_star: "within 1e-4 per-pixel L2 after the same sample count"


@pytest.fixture(scope="module")
def dev():
    r = prt.HipWavefrontRenderer(device=0)
    yield r


def make_renderer(scene, W, H, max_depth=20, seed=0, cam=None, **kw):
    cam = cam or prt.Camera(width=W, height=H)
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=max_depth, seed=seed, **kw)
    r.Init(film, scene, cam)
    return r, film, cam


# ---- function level ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("W,H", [(256, 256), (1280, 720), (1920, 1080), (3840, 2160), (37, 19)])
def test_camera_rays_bit_exact(W, H):
    cam = prt.Camera(width=W, height=H)
    r = prt.HipWavefrontRenderer(device=0)
    r.SetCamera(cam)
    rng = np.random.default_rng(W)
    xs = np.concatenate([np.arange(0, W, max(1, W // 61)) + 0.5, rng.uniform(0, W, 500)]).astype(np.float32)
    ys = np.resize(np.concatenate([np.arange(0, H, max(1, H // 47)) + 0.5, rng.uniform(0, H, 500)]), xs.size)
    ys = ys.astype(np.float32)
    o, d = r.camera_rays(xs, ys)
    oo, od = orc.camera_rays(cam.desc(), xs, ys)
    assert np.array_equal(o, oo) and np.array_equal(d, od)


@pytest.mark.parametrize("preset", util.PRESETS + ["RANDOM_BALLS_LARGE"])
def test_closest_hit_presets_bit_exact(preset):
    scene = prt.Scene(preset)
    r, film, cam = make_renderer(scene, 64, 48)
    rng = np.random.default_rng(hash(preset) % 1000)
    n = 4096
    o1, d1 = util.random_rays(rng, n // 2, center=(0, 1, 0), radius=14.0, spread=6.0)
    px = rng.uniform(0, 64, n // 2).astype(np.float32)
    py = rng.uniform(0, 48, n // 2).astype(np.float32)
    o2, d2 = orc.camera_rays(cam.desc(), px, py)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    got = r.closest_hit(o, d)
    want = util.oracle_scene(scene).closest_hit(o, d, use_bvh=False)
    assert util.hits_equal(got, want) == []
    assert (got["prim"] >= 0).sum() > n // 8


def _mesh_rays(rng, n, extent=1.0):
    """Mix of far primary-like rays, rays starting on/near the surface region and grazing rays."""
    o1, d1 = util.random_rays(rng, n // 2, center=(0, 0, 0), radius=9.0, spread=extent)
    o2 = rng.uniform(-extent, extent, size=(n - n // 2, 3)).astype(np.float32)
    d2 = rng.normal(size=o2.shape).astype(np.float32)
    d2 = np.stack([prt.glm_normalize(v) for v in d2])
    return np.concatenate([o1, o2]), np.concatenate([d1, d2])


@pytest.mark.parametrize("ply,target", [("icosahedron.ply", 0), ("bunny.ply", 0), ("hand.ply", 0)])
def test_closest_hit_mesh_vs_linear_scan_bit_exact(ply, target):
    mesh = prt.Mesh(prt.scenes.asset(ply))
    scene = prt.scenes.mesh_scene(mesh)
    r, _, _ = make_renderer(scene, 16, 16)
    o, d = _mesh_rays(np.random.default_rng(5), 3000)
    got = r.closest_hit(o, d)
    want = util.oracle_scene(scene).closest_hit(o, d, use_bvh=False, n_threads=8)
    assert util.hits_equal(got, want) == []
    assert (got["prim"] >= 2).sum() > 80  # plenty of triangle hits (prims 0,1 are the analytic quads)


def test_closest_hit_rays_through_vertices_and_edges_bit_exact():
    """Adversarial culling cases: rays aimed exactly at mesh vertices / edge midpoints from far away."""
    mesh = prt.Mesh(prt.scenes.asset("bunny.ply"))
    scene = prt.scenes.mesh_scene(mesh)
    r, _, _ = make_renderer(scene, 16, 16)
    v = mesh.GetVertices()
    idx = mesh.GetIndices()
    rng = np.random.default_rng(9)
    sel = rng.choice(len(v), 800, replace=False)
    tri = idx[rng.choice(len(idx), 800, replace=False)]
    targets = np.concatenate([v[sel], (v[tri[:, 0]] + v[tri[:, 1]]) * np.float32(0.5)]).astype(np.float32)
    o = (rng.normal(size=targets.shape) * 30).astype(np.float32)
    d = np.stack([prt.glm_normalize(x) for x in (targets - o)])
    got = r.closest_hit(o, d)
    want = util.oracle_scene(scene).closest_hit(o, d, use_bvh=False, n_threads=8)
    assert util.hits_equal(got, want) == []


def test_closest_hit_refined_mesh_vs_oracle_bvh_bit_exact():
    mesh = prt.scenes.refined("bunny.ply", 70_000)
    scene = prt.scenes.mesh_scene(mesh)
    r, _, _ = make_renderer(scene, 16, 16)
    o, d = _mesh_rays(np.random.default_rng(6), 20000)
    got = r.closest_hit(o, d)
    want = util.oracle_scene(scene).closest_hit(o, d, use_bvh=True, n_threads=8)
    assert util.hits_equal(got, want) == []


def test_closest_hit_full_size_mesh_vs_oracle_bvh_bit_exact():
    """The headline mesh (dragon refined to 870,000 triangles), every traversal kernel instance."""
    scene, cam, W, H, spp, depth = prt.scenes.config("C3")
    r, _, _ = make_renderer(scene, 64, 36, cam=prt.Camera(cam.position, width=64, height=36))
    info = r.bvh_info()
    assert info.n_triangles == 870_000 and info.max_stack4 <= 63
    rng = np.random.default_rng(12)
    o, d = _mesh_rays(rng, 30000)
    px = rng.uniform(0, 64, 10000).astype(np.float32)
    py = rng.uniform(0, 36, 10000).astype(np.float32)
    o2, d2 = r.camera_rays(px, py)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    got = r.closest_hit(o, d)
    want = util.oracle_scene(scene).closest_hit(o, d, use_bvh=True, n_threads=8)
    assert util.hits_equal(got, want) == []
    assert (got["prim"] >= 2).sum() > 8000
    for v in (1, 2):  # the one-thread-per-ray kernels over the binary tree give the same hits
        r.set_variant(v)
        assert util.hits_equal(r.closest_hit(o, d), want) == []
    r.set_variant(0)
    assert info.n_nodes8 > 0 and info.depth8 <= 16
    # every tunable / kernel instance gives the same hits: the compressed 8-wide tree (default, wide = 2) at both
    # occupancies, the 4-wide tree's instances (wide = 1), the binary tree (wide = 0)
    for wide, name, val in ((2, "stack_lds", 4), (2, "stack_lds", 5), (2, "stack_lds", 6), (2, "chunk", 64), (2, "xcd_affinity", 1), (2, "exit_max", 0), (2, "refill_min", 1),
                            (2, "steal", 0), (2, "steal", 1), (2, "steal", 64), (2, "tail", 0), (2, "tail", 8),
                            (1, "stack_lds", 0), (1, "stack_lds", 1), (1, "stack_lds", 2), (1, "stack_lds", 3),
                            (1, "stack_lds", 39), (1, "xcd_affinity", 1), (1, "chunk", 64), (0, "stack_lds", 0)):
        r.set_param("wide", wide)
        r.set_param(name, val)
        assert util.hits_equal(r.closest_hit(o, d), want) == [], (wide, name, val)
        for k_, v_ in (("stack_lds", 0), ("chunk", 256), ("xcd_affinity", 0), ("exit_max", 8), ("refill_min", 32), ("steal", 8), ("tail", 1)):
            r.set_param(k_, v_)
    r.set_param("wide", 2)
    st = r.measure_traversal()
    assert st.max_stack_used < info.depth8 and st.bvh_node_visits > 0
    # the overflow path of the 8-wide kernel ("never happens": the stack covers the tree): with the stack capped at 3
    # entries most rays are handed to the spill-capable 4-wide instance through the overflow list -- same hits
    for inst in (0, 4):
        r.set_param("stack_lds", inst)
        r.set_param("stack_cap", 3)
        assert util.hits_equal(r.closest_hit(o, d), want) == [], inst
        r.set_param("stack_cap", 0)
    r.set_param("stack_lds", 0)


def test_closest_hit_small_launches_subtree_stealing_bit_exact():
    """Launches of 1 ... 5000 rays spend their whole life in the state the end of a big launch is in: the ray buffer is
    exhausted at once, waves drain, and idle lanes take over pending subtrees of the remaining rays.  Diffuse
    bounce rays off the headline mesh (the long ones: up to ~130 node steps), most aggressive stealing (one idle lane is
    enough), bit-exact against the oracle."""
    scene, cam, W, H, spp, depth = prt.scenes.config("C3")
    r, _, _ = make_renderer(scene, 64, 36, cam=prt.Camera(cam.position, width=64, height=36))
    rng = np.random.default_rng(21)
    o, d = r.camera_rays(rng.uniform(0, 64, 20000).astype(np.float32), rng.uniform(0, 36, 20000).astype(np.float32))
    h = r.closest_hit(o, d)
    on_mesh = h["prim"] >= 2
    pos, nrm = h["position"][on_mesh], h["normal"][on_mesh]
    v = rng.normal(size=pos.shape).astype(np.float32)
    dirs = nrm + v / np.linalg.norm(v, axis=1, keepdims=True)
    dirs = np.stack([prt.glm_normalize(x) for x in dirs[:5000]])
    pos = pos[:5000]
    osc = util.oracle_scene(scene)
    want = osc.closest_hit(pos, dirs, use_bvh=True, n_threads=8)
    assert (want["prim"] >= 2).sum() > 300
    for steal in (1, 8, 0):
        r.set_param("steal", steal)
        for n in (1, 63, 64, 65, 257, 1000, 5000):
            assert util.hits_equal(r.closest_hit(pos[:n], dirs[:n]), want[:n]) == [], (steal, n)


def test_closest_hit_every_tier_of_the_ray_hand_out_bit_exact():
    """The traversal kernel hands its ray buffer out in three tiers: big grabs of `big` chunks at the front of big
    launches, ordinary chunks, single 64-ray granules at the end (or throughout, for small launches).  A small resident grid
    and forced thresholds put all three tiers, odd sizes and the tier boundaries into one 200 k-ray launch: every ray must
    come out exactly once, with the oracle's hit."""
    mesh = prt.scenes.refined("bunny.ply", 30_000)
    scene = prt.scenes.mesh_scene(mesh)
    r, _, _ = make_renderer(scene, 16, 16)
    rng = np.random.default_rng(5)
    o, d = util.random_rays(rng, 200_003, center=(0, 0.3, 0), radius=4.0, spread=1.5)
    want = util.oracle_scene(scene).closest_hit(o, d, use_bvh=True, n_threads=8)
    assert (want["prim"] >= 2).sum() > 20_000
    for params in ({"grid_blocks": 8, "big": 3, "big_min": 1, "big_keep": 4, "chunk": 128, "static_small": 0},
                   {"grid_blocks": 64, "static_small": 4096},  # every granule dealt round-robin, no cursor
                   {"grid_blocks": 8, "static_small": 8},
                   {"grid_blocks": 8, "big": 2, "big_min": 1, "big_keep": 0, "chunk": 256, "tail": 3},
                   {"grid_blocks": 24, "big": 5, "big_min": 8, "big_keep": 1, "chunk": 64, "tail": 0},
                   {"grid_blocks": 1024, "big": 2, "big_min": 96, "big_keep": 32, "chunk": 256, "tail": 1}):
        for k, v in params.items():
            r.set_param(k, v)
        for n in (200_003, 65_537):
            assert util.hits_equal(r.closest_hit(o[:n], d[:n]), want[:n]) == [], (params, n)


def test_scatter_bit_exact_all_materials():
    scene = prt.Scene("DEFAULT")
    scene.AddMetal((0.9, 0.8, 0.7), 0.0)
    scene.AddMetal((0.9, 0.8, 0.7), 0.6)
    scene.AddDielectric(1.5)
    scene.AddDielectric(1.0)
    r, _, _ = make_renderer(scene, 8, 8)
    rng = np.random.default_rng(11)
    n = 6000
    hits = np.zeros(n, dtype=prt.capi.HIT_DTYPE)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm = np.stack([prt.glm_normalize(x) for x in nrm])
    ind = rng.normal(size=(n, 3)).astype(np.float32)
    ind = np.stack([prt.glm_normalize(x) for x in ind])
    flip = (np.einsum("ij,ij->i", nrm, ind) > 0)
    nrm[flip] *= -1  # shading normals always face the incoming ray
    hits["prim"] = 0
    hits["front_face"] = rng.integers(0, 2, n)
    hits["material_id"] = rng.integers(0, len(scene.materials), n)
    hits["position"] = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    hits["normal"] = nrm
    state = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    sc, att, em, oo, od, st = r.scatter(ind, hits, state)
    for i in range(n):
        w = orc.scatter(scene.materials[int(hits["material_id"][i])], ind[i], hits[i], int(state[i]))
        assert bool(sc[i]) == w[0], i
        assert np.array_equal(att[i], w[1]) and np.array_equal(em[i], w[2]), i
        assert np.array_equal(oo[i], w[3]) and np.array_equal(od[i], w[4]), (i, od[i], w[4])
        assert int(st[i]) == w[5], i


def test_dielectric_scatter_sweep_bit_exact():
    """10^6 dielectric scatter events (DielectricMaterial::Scatter + fresnelReflectance, material.h:76-109): grazing,
    near-normal and total-internal-reflection cases, both faces, five indices of refraction.  The Schlick term goes
    through a double pow(x, 5): device and oracle compute the correctly rounded x^5 (csrc/prt_device.h pow5_rn), so a
    one-ulp disagreement there would flip a reflect / refract decision or an RNG state somewhere in a sweep this size."""
    scene = prt.Scene(preset=None)
    for ior in (1.5, 1.33, 2.4, 1.0, 1.05):
        scene.AddDielectric(ior)
    scene.AddQuad(1, 1, 0)
    r, _, _ = make_renderer(scene, 8, 8)
    rng = np.random.default_rng(17)
    n = 1_000_000
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    # incoming directions at a chosen angle to the normal: uniform in cos, plus a dense band near grazing / near normal
    cos_t = np.concatenate([rng.uniform(0, 1, n // 2), rng.uniform(0, 0.02, n // 4), rng.uniform(0.98, 1, n - n // 2 - n // 4)])
    t = np.cross(nrm, rng.normal(size=(n, 3)))
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    ind = (-cos_t[:, None] * nrm + np.sqrt(1 - cos_t ** 2)[:, None] * t).astype(np.float32)
    l = np.sqrt((ind.astype(np.float64) ** 2).sum(axis=1, keepdims=True))
    ind = (ind / l).astype(np.float32)
    hits = np.zeros(n, dtype=prt.capi.HIT_DTYPE)
    hits["front_face"] = rng.integers(0, 2, n)
    hits["material_id"] = rng.integers(0, 5, n)
    hits["position"] = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    hits["normal"] = nrm
    state = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    sc, att, em, oo, od, st = r.scatter(ind, hits, state)
    wsc, watt, wem, woo, wod, wst = orc.scatter_batch(scene.materials, ind, hits, state)
    assert np.array_equal(sc, wsc) and np.array_equal(st, wst)
    assert np.array_equal(att, watt) and np.array_equal(em, wem) and np.array_equal(oo, woo)
    bad = np.flatnonzero((od != wod).any(axis=1))
    assert bad.size == 0, (bad[:5], od[bad[:5]], wod[bad[:5]])
    drew = st != state  # the RNG is drawn only when refraction is possible (short-circuit, material.h:88)
    assert 0.5 < drew.mean() < 0.99 and sc.all()


def test_failed_init_after_a_valid_init_leaves_no_scene():
    """prt_set_scene rewrites the context in place; when it fails the context must end up WITHOUT a scene (an error on the
    next render), never with half of the new one (null dereference in k_raygen)."""
    good = prt.Scene("CORNELL")
    r, film, cam = make_renderer(good, 32, 32, max_depth=3)
    r.ProgressiveRender()
    mesh = prt.Mesh(prt.scenes.asset("icosahedron.ply"))
    bad = prt.Scene("CORNELL")
    bad.AddInstance(mesh, 0, scale=(1.0, 3.0, 1.0))  # non-uniform scale: rejected after the host copies were rewritten
    with pytest.raises(prt.PrtError, match="uniform scale"):
        r.Init(film, bad, cam)
    with pytest.raises(prt.PrtError, match="prt_set_scene"):
        r.ProgressiveRender()
    with pytest.raises(prt.PrtError, match="prt_set_scene"):
        r.closest_hit(np.zeros((4, 3), np.float32), np.tile(np.float32([0, 0, 1]), (4, 1)))
    bad2 = prt.Scene("CORNELL")
    bad2.AddMesh(mesh, 99)  # material out of range
    with pytest.raises(prt.PrtError):
        r.Init(film, bad2, cam)
    with pytest.raises(prt.PrtError, match="prt_set_scene"):
        r.ProgressiveRender()
    r.Init(film, good, cam)  # and a valid scene works again on the same context
    film.Clear()
    r.ProgressiveRender()
    r.download()
    acc, wts, rays = util.oracle_scene(good).render(cam.desc(), 32, 32, spp=1, max_depth=3, seed=0, iterative=True)
    assert np.array_equal(film.accum, acc)


# ---- image level ------------------------------------------------------------------------------------------------------
IMAGE_CASES = [
    # name, preset, W, H, spp, max_depth, seed
    ("C1_cornell", "CORNELL", 256, 256, 1, 2, 0),
    ("default", "DEFAULT", 128, 72, 16, 5, 0),
    ("balls_small", "RANDOM_BALLS_SMALL", 128, 72, 4, 5, 3),
    ("material_test", "MATERIAL_TEST", 96, 64, 8, 20, 1),
    ("light_test", "LIGHT_TEST", 96, 64, 4, 6, 2),
    ("ragged", "DEFAULT", 37, 19, 3, 4, 5),  # width/height not multiples of the 8x8 tile
]


@pytest.mark.parametrize("name,preset,W,H,spp,depth,seed", IMAGE_CASES)
def test_image_parity(name, preset, W, H, spp, depth, seed):
    scene = prt.Scene(preset)
    r, film, cam = make_renderer(scene, W, H, max_depth=depth, seed=seed)
    for _ in range(spp):
        r.ProgressiveRender()  # exactly one sample per call (renderer.h:14 contract)
    r.download()
    osc = util.oracle_scene(scene)
    acc_it, w_it, rays_it = osc.render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=seed, iterative=True,
                                       n_threads=8)
    acc_re, w_re, rays_re = osc.render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=seed, iterative=False,
                                       n_threads=8)
    assert np.array_equal(film.weights, w_it) and (film.weights == spp).all()
    # throughput form: same arithmetic, same order -> identical bits
    nbad = int((film.accum != acc_it).any(axis=-1).sum())
    assert nbad == 0, f"{nbad} pixels differ from the oracle's throughput form"
    # the reference CPU backend's recursion differs only in fp32 association
    l2 = util.image_l2(film.accum / spp, acc_re / spp)
    assert l2 <= TOL_L2, l2
    st = r.stats()
    assert st.rays_total == rays_it == rays_re
    assert st.samples == spp


def test_image_parity_mesh_scene_vs_linear_scan_oracle():
    mesh = prt.Mesh(prt.scenes.asset("icosahedron.ply")).refine(1500)
    scene = prt.scenes.mesh_scene(mesh)
    W, H, spp, depth = 64, 64, 4, 5
    cam = prt.Camera(position=(2.0, 2.0, 3.0), width=W, height=H)
    r, film, _ = make_renderer(scene, W, H, max_depth=depth, seed=4, cam=cam)
    r.ProgressiveRender(spp)
    r.download()
    osc = util.oracle_scene(scene)
    acc, wts, rays = osc.render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=4, iterative=True, use_bvh=False,
                                n_threads=8)
    assert np.array_equal(film.accum, acc) and np.array_equal(film.weights, wts)
    assert r.stats().rays_total == rays


@pytest.mark.parametrize("fuse", [0, 1, "exact", "exact-fuse", "exact-fullrecords", "exact-nopixelhit"])
def test_image_parity_bunny_vs_oracle_bvh(fuse):
    """fuse = 1: the producers shade one analytic-only segment in place (paths advance at different rates).
    "exact": k_shade grids sized from the ray counts the host reads back while the traversal runs (the mode big batches
    use), including the early stop at the first bounce without rays.  The default pipeline stores compact primary rays
    (12 B per path + per-pixel records); "fullrecords" switches that off."""
    mesh = prt.scenes.refined("bunny.ply", 30_000)
    scene = prt.scenes.mesh_scene(mesh)
    W, H, spp, depth = 160, 90, 2, 5
    cam = prt.Camera(position=(2.0, 1.5, 3.0), width=W, height=H)
    r, film, _ = make_renderer(scene, W, H, max_depth=depth, seed=8, cam=cam)
    if isinstance(fuse, str):
        r.set_param("exact_grids", 2)
        r.set_param("fuse", 1 if fuse.endswith("fuse") else 0)
        if fuse.endswith("fullrecords"):  # k_raygen stores full 56-B ray records instead of compact primary rays
            r.set_param("compact_primary", 0)
        if fuse.endswith("nopixelhit"):  # the first k_shade rebuilds the primary hit per sample (no k_primary_hit records)
            r.set_param("primary_hit", 0)
    else:
        r.set_param("fuse", fuse)
    r.ProgressiveRender(spp)
    r.download()
    osc = util.oracle_scene(scene)
    acc, wts, rays = osc.render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=8, iterative=True, use_bvh=True,
                                n_threads=8)
    assert np.array_equal(film.accum, acc)
    acc_re, _, _ = osc.render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=8, iterative=False, use_bvh=True,
                              n_threads=8)
    assert util.image_l2(film.accum / spp, acc_re / spp) <= TOL_L2
    st = r.stats()
    assert st.rays_total == rays and st.rays_per_depth[0] == spp * W * H
    assert all(st.rays_per_depth[d] >= st.rays_per_depth[d + 1] for d in range(depth)) and st.rays_per_depth[depth] == 0


def test_triangulated_quads_match_analytic_quads():
    """SURVEY §7 hard part 1: a scene whose quads are tessellated into triangles must render like the
    analytic quads (same geometry, different intersection routine -> equal up to rounding, apart from a
    few pixels where an edge hit/miss decision flips).  CORNELL itself is unsuitable: its three rotated
    quads are coplanar and overlap (scene.cpp:343-349), so the winner there is decided by rounding."""
    base = prt.Scene(preset=None)
    g = base.AddLambertian((0.7, 0.7, 0.4))
    e = base.AddEmissive((3, 4, 2))
    m = base.AddMetal((0.9, 0.8, 0.7), 0.05)
    base.AddQuad(20, 20, g)
    base.AddQuad(8, 8, e, euler_deg=(50, 0, 0), translation=(-4, 7, 7))
    base.AddQuad(6, 6, g, euler_deg=(90, 0, 0), translation=(0, 3, -5))
    base.AddCircle(1.0, m, translation=(0, 1, 0))
    W = H = 96
    ra, fa, cam = make_renderer(base, W, H, max_depth=4, seed=2)
    rt, ft, _ = make_renderer(prt.scenes.triangulate_quads(base), W, H, max_depth=4, seed=2)
    ra.ProgressiveRender(8)
    rt.ProgressiveRender(8)
    ra.download()
    rt.download()
    close = np.isclose(fa.accum, ft.accum, rtol=1e-3, atol=1e-3).all(axis=-1)
    assert close.mean() > 0.97, close.mean()
    assert abs(fa.accum.mean() - ft.accum.mean()) < 0.02 * fa.accum.mean()


@pytest.mark.parametrize("jitter,rr_depth,clamp", [(1, 0, 0.0), (0, 2, 0.0), (0, 0, 0.8), (1, 1, 2.5)])
@pytest.mark.parametrize("kind", ["preset", "mesh"])
def test_image_parity_with_sampling_upgrades(kind, jitter, rr_depth, clamp):
    """PrtSampling (jittered primary rays as in the reference's OptiX backend, Russian roulette, firefly clamp):
    same bits as the oracle's throughput form with the same options, same ray count."""
    if kind == "preset":
        scene, W, H, spp, depth, seed = prt.Scene("MATERIAL_TEST"), 96, 64, 4, 8, 5
        cam = prt.Camera(width=W, height=H)
    else:
        scene = prt.scenes.mesh_scene(prt.scenes.refined("bunny.ply", 20_000))
        W, H, spp, depth, seed = 128, 72, 3, 6, 9
        cam = prt.Camera(position=(2.0, 1.5, 3.0), width=W, height=H)
    r, film, _ = make_renderer(scene, W, H, max_depth=depth, seed=seed, cam=cam)
    sp = r.set_sampling(jitter=jitter, rr_depth=rr_depth, clamp=clamp)
    r.ProgressiveRender(spp)
    r.download()
    acc, wts, rays = util.oracle_scene(scene).render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=seed, iterative=True,
                                                     use_bvh=(kind == "mesh"), n_threads=8, sampling=sp)
    assert np.array_equal(film.accum, acc) and np.array_equal(film.weights, wts)
    assert r.stats().rays_total == rays
    if clamp:
        assert film.accum.max() <= spp * clamp
    r.set_sampling()  # all off again
    film.Clear()
    r.frame_index = 0
    r.ProgressiveRender(1)
    r.download()
    acc0, _, _ = util.oracle_scene(scene).render(cam.desc(), W, H, spp=1, max_depth=depth, seed=seed, iterative=True,
                                                 use_bvh=(kind == "mesh"), n_threads=8)
    assert np.array_equal(film.accum, acc0)


@pytest.mark.parametrize("preset", ["RANDOM_BALLS_MEDIUM", "RANDOM_BALLS_LARGE"])
def test_many_analytic_primitives_bvh_scan_equals_the_linear_scan(preset):
    """Scenes with more than 16 analytic primitives (the reference's default scene has 809 and scans all of them for
    every ray, primitive.cpp:26) walk a BVH over the primitives' world boxes inside the producers: same closest hits
    and same image as the reference's linear scan (oracle), and as the product's own linear scan (prim_bvh = 0)."""
    scene = prt.Scene(preset)
    W, H, spp, depth = 96, 54, 2, 8
    r, film, cam = make_renderer(scene, W, H, max_depth=depth, seed=5)
    rng = np.random.default_rng(41)
    o, d = util.random_rays(rng, 4000)
    got = r.closest_hit(o, d)
    want = util.oracle_scene(scene).closest_hit(o, d, use_bvh=False, n_threads=8)
    assert util.hits_equal(got, want) == [] and (got["prim"] >= 0).sum() > 2000
    r.ProgressiveRender(spp)
    r.download()
    acc, wts, rays = util.oracle_scene(scene).render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=5, iterative=True,
                                                     n_threads=8)
    assert np.array_equal(film.accum, acc) and r.stats().rays_total == rays
    r2 = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=5)
    r2.set_param("prim_bvh", 0)
    f2 = prt.Film(W, H)
    r2.Init(f2, scene, cam)
    r2.ProgressiveRender(spp)
    r2.download()
    assert np.array_equal(f2.accum, film.accum)


# ---- placed mesh copies (PrtInstance): two-level traversal ----------------------------------------------------------------
INST_PLACEMENTS = [(1.0, (0, 0, 0), (0, 2.6, 0)), (0.5, (0, 40, 0), (2.5, 0, 0)), (1.5, (30, 0, 0), (-3, 0.5, -1)),
                   (0.8, (10, 70, 25), (0, 0.3, 3)), (2.0, (0, 180, 0), (4.0, 1.0, -4.0))]


def _instanced_scene(mesh, with_world_mesh):
    sc = prt.Scene(preset=None)
    g = sc.AddLambertian((0.5, 0.5, 0.5))
    b = sc.AddLambertian((0.8, 0.6, 0.4))
    m = sc.AddMetal((0.9, 0.9, 0.9), 0.1)
    e = sc.AddEmissive((6, 6, 6))
    sc.AddQuad(30, 30, g, translation=(0, -1.2, 0))
    sc.AddQuad(4, 4, e, euler_deg=(180, 0, 0), translation=(0, 7, 0))
    if with_world_mesh:
        sc.AddMesh(mesh, b)
    for i, (scale, euler, tr) in enumerate(INST_PLACEMENTS):
        sc.AddInstance(mesh, b if i % 2 == 0 else m, scale=scale, euler_deg=euler, translation=tr)
    return sc


@pytest.mark.parametrize("with_world_mesh", [False, True])
def test_closest_hit_instances_vs_linear_scan_bit_exact(with_world_mesh):
    """Every placed triangle is a reference Primitive{Triangle, Material, Transform}: the two-level walk must return
    what PrimitiveList::Intersect's linear scan returns (oracle, brute force), bit for bit."""
    mesh = prt.Mesh(prt.scenes.asset("icosahedron.ply")).refine(1200)
    scene = _instanced_scene(mesh, with_world_mesh)
    r, _, _ = make_renderer(scene, 16, 16)
    info = r.bvh_info()
    assert info.n_nodes8 > 0 and info.depth8 <= 12
    rng = np.random.default_rng(21)
    o, d = util.random_rays(rng, 6000, center=(0, 0.5, 0), radius=11.0, spread=4.5)
    o2 = rng.uniform(-4, 4, size=(3000, 3)).astype(np.float32)
    d2 = np.stack([prt.glm_normalize(v) for v in rng.normal(size=o2.shape).astype(np.float32)])
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    got = r.closest_hit(o, d)
    want = util.oracle_scene(scene).closest_hit(o, d, use_bvh=False, n_threads=8)
    assert util.hits_equal(got, want) == []
    nt = mesh.n_triangles
    owners = set(((got["prim"][got["prim"] >= 2] - 2) // nt).tolist())
    assert owners == set(range(len(INST_PLACEMENTS) + (1 if with_world_mesh else 0)))


def test_identity_instance_renders_like_the_world_space_mesh():
    mesh = prt.scenes.refined("bunny.ply", 20_000)
    a = prt.scenes.mesh_scene(mesh)
    b = prt.Scene(preset=None)
    b.materials = list(a.materials)
    b.primitives = list(a.primitives)
    b.AddInstance(mesh, a.meshes[0][1])
    W, H, spp, depth = 96, 54, 2, 5
    cam = prt.Camera(position=(2.0, 1.5, 3.0), width=W, height=H)
    ra, fa, _ = make_renderer(a, W, H, max_depth=depth, seed=3, cam=cam)
    rb, fb, _ = make_renderer(b, W, H, max_depth=depth, seed=3, cam=cam)
    ra.ProgressiveRender(spp)
    rb.ProgressiveRender(spp)
    ra.download()
    rb.download()
    assert np.array_equal(fa.accum, fb.accum) and ra.stats().rays_total == rb.stats().rays_total


def test_image_parity_instanced_scene_vs_oracle():
    mesh = prt.scenes.refined("bunny.ply", 12_000)
    scene = _instanced_scene(mesh, with_world_mesh=True)
    W, H, spp, depth = 128, 72, 2, 5
    cam = prt.Camera(position=(6.0, 4.0, 9.0), width=W, height=H)
    r, film, _ = make_renderer(scene, W, H, max_depth=depth, seed=11, cam=cam)
    r.ProgressiveRender(spp)
    r.download()
    acc, wts, rays = util.oracle_scene(scene).render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=11, iterative=True,
                                                     use_bvh=True, n_threads=8)
    assert np.array_equal(film.accum, acc) and np.array_equal(film.weights, wts)
    assert r.stats().rays_total == rays
    st = r.measure_traversal()
    assert st.bvh_node_visits > 0 and st.bvh_tri_tests > 0 and st.max_stack_used <= 12


def test_traversal_error_flag_reaches_the_caller_and_clears():
    """The two-level walk has no fallback for a full stack: the kernel raises its error flag, prt_synchronize (whose copy of
    the flag to pinned memory is enqueued before its one stream wait) reports it, clears it, and the context stays usable."""
    mesh = prt.scenes.refined("bunny.ply", 12_000)
    scene = _instanced_scene(mesh, with_world_mesh=True)
    W, H, depth = 96, 54, 4
    cam = prt.Camera(position=(6.0, 4.0, 9.0), width=W, height=H)
    r, film, _ = make_renderer(scene, W, H, max_depth=depth, seed=2, cam=cam)
    r.set_param("stack_cap", 1)
    with pytest.raises(prt.PrtError, match="overflow"):
        r.ProgressiveRender(2)
    r.set_param("stack_cap", 0)
    film.Clear()
    r.reset_stats()
    r.ProgressiveRender(2)  # no stale flag
    r.download()
    acc, wts, rays = util.oracle_scene(scene).render(cam.desc(), W, H, spp=2, max_depth=depth, seed=2, iterative=True,
                                                     use_bvh=True, n_threads=8)
    assert np.array_equal(film.accum, acc) and r.stats().rays_total == rays


def test_placed_copies_next_to_many_analytic_primitives():
    """Both acceleration structures at once: 109 analytic primitives (primitive BVH in the producers) + placed copies of
    a mesh (two-level tree in the traversal kernel), jittered and with roulette: image bit-exact vs the oracle."""
    scene = prt.Scene("RANDOM_BALLS_SMALL")
    mesh = prt.Mesh(prt.scenes.asset("icosahedron.ply")).refine(600)
    body = scene.AddMetal((0.9, 0.7, 0.5), 0.2)
    scene.AddInstance(mesh, body, scale=1.5, euler_deg=(0, 30, 0), translation=(0.0, 1.6, 0.0))
    scene.AddInstance(mesh, body, scale=0.7, euler_deg=(45, 0, 20), translation=(3.0, 0.8, 1.0))
    W, H, spp, depth = 96, 54, 2, 6
    r, film, cam = make_renderer(scene, W, H, max_depth=depth, seed=13)
    sp = r.set_sampling(jitter=1, rr_depth=2)
    r.ProgressiveRender(spp)
    r.download()
    acc, wts, rays = util.oracle_scene(scene).render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=13, iterative=True,
                                                     use_bvh=True, n_threads=8, sampling=sp)
    assert np.array_equal(film.accum, acc) and r.stats().rays_total == rays


def test_instances_must_be_similarity_transforms():
    mesh = prt.Mesh(prt.scenes.asset("icosahedron.ply"))
    sc = prt.Scene(preset=None)
    sc.AddInstance(mesh, sc.AddLambertian((1, 1, 1)), scale=(1.0, 2.0, 1.0))
    r = prt.HipWavefrontRenderer(device=0)
    with pytest.raises(prt.PrtError, match="uniform scale"):
        r.Init(prt.Film(8, 8), sc, prt.Camera(width=8, height=8))


def test_headline_frame_one_sample_bit_exact():
    """BASELINE's headline workload at its FULL size: C3 (870,000 triangles), 1920x1080, max_depth 5, one sample per
    pixel = 3.7 M ray segments, every pixel compared with the oracle's throughput form (own BVH), plus the ray
    count; then the same frame rendered as two batches with 64 samples in flight must continue it bit for bit."""
    scene, cam, W, H, _, depth = prt.scenes.config("C3")
    r, film, _ = make_renderer(scene, W, H, max_depth=depth, seed=0, cam=cam)
    r.ProgressiveRender(1)
    r.download()
    acc, wts, rays = util.oracle_scene(scene).render(cam.desc(), W, H, spp=1, max_depth=depth, seed=0, iterative=True,
                                                     use_bvh=True, n_threads=16)
    assert np.array_equal(film.weights, wts) and (wts == 1).all()
    nbad = int((film.accum != acc).any(axis=-1).sum())
    assert nbad == 0, f"{nbad} of {W * H} pixels differ from the oracle"
    st = r.stats()
    assert st.rays_total == rays and st.rays_per_depth[0] == W * H


# ---- device-side builder (prt_set_param("gpu_build", 1)) --------------------------------------------------------------
@pytest.mark.parametrize("mode", [1, "1-top-on-device", 2])
@pytest.mark.parametrize("ply,target", [("icosahedron.ply", 0), ("bunny.ply", 0), ("bunny.ply", 70_000)])
def test_device_built_tree_is_valid_and_gives_the_same_hits(ply, target, mode, monkeypatch):
    """The 8-wide trees built on the GPU (gpu_build 1: PLOC + optimal collapse, with the top of the tree from the host's SAH
    sweep or, PRT_PLOC_TOP=device, from full-search clustering passes on the device; 2: Morton octree): structurally valid
    (same checker as the host builder's tree), and - because the closest hit does not depend on the tree - bit-exact hits
    against the oracle."""
    monkeypatch.setenv("PRT_PLOC_TOP", "device" if mode == "1-top-on-device" else "")
    mode = 1 if mode == "1-top-on-device" else mode
    mesh = prt.scenes.refined(ply, target) if target else prt.Mesh(prt.scenes.asset(ply))
    scene = prt.scenes.mesh_scene(mesh)
    r = prt.HipWavefrontRenderer(device=0)
    r.set_param("gpu_build", mode)
    film = prt.Film(16, 16)
    r.Init(film, scene, prt.Camera(width=16, height=16))
    info = r.bvh_info()
    assert info.built_on_device == 1 and info.n_nodes8 > 0 and info.depth8 <= 15 and info.build_ms > 0
    n8 = r.bvh_read8()
    _, tris = r.bvh_read()
    fill, depth = util.check_bvh8(n8, tris)
    assert depth == info.depth8 and min(fill) >= 1 and (np.mean(fill) >= 2.0)
    prim = tris[:, 3].view(np.uint32).astype(np.int64) - len(scene.primitives)
    assert sorted(prim.tolist()) == list(range(mesh.n_triangles))
    o, d = _mesh_rays(np.random.default_rng(31), 6000)
    got = r.closest_hit(o, d)
    want = util.oracle_scene(scene).closest_hit(o, d, use_bvh=True, n_threads=8)
    assert util.hits_equal(got, want) == []
    assert (got["prim"] >= 2).sum() > 300


def test_device_built_tree_renders_the_same_image():
    mesh = prt.scenes.refined("bunny.ply", 30_000)
    scene = prt.scenes.mesh_scene(mesh)
    W, H, spp, depth = 128, 72, 2, 5
    cam = prt.Camera(position=(2.0, 1.5, 3.0), width=W, height=H)
    imgs = []
    for gpu_build in (0, 1, 2):
        r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=8)
        r.set_param("gpu_build", gpu_build)
        film = prt.Film(W, H)
        r.Init(film, scene, cam)
        r.ProgressiveRender(spp)
        r.download()
        imgs.append((film.accum.copy(), r.stats().rays_total, r.bvh_info().built_on_device))
    assert imgs[0][2] == 0 and imgs[1][2] == 1 and imgs[2][2] == 1
    assert np.array_equal(imgs[0][0], imgs[1][0]) and imgs[0][1] == imgs[1][1]
    assert np.array_equal(imgs[0][0], imgs[2][0]) and imgs[0][1] == imgs[2][1]


@pytest.mark.parametrize("with_world_mesh", [False, True])
def test_device_built_two_level_tree_gives_the_same_hits_and_image(with_world_mesh):
    """gpu_build = 1 on a scene with placed copies: every instanced mesh's tree (in its own space), the world-space meshes'
    tree and the top-level tree over the copies' boxes come from the device builder; hits bit-exact against the oracle's
    brute-force scan over all placed triangles, image bit-exact against the oracle and against the host-built scene."""
    mesh = prt.scenes.refined("bunny.ply", 12_000)
    scene = _instanced_scene(mesh, with_world_mesh)
    W, H, spp, depth = 128, 72, 2, 5
    cam = prt.Camera(position=(6.0, 4.0, 9.0), width=W, height=H)
    films = []
    for gpu_build in (1, 0):
        r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=11)
        r.set_param("gpu_build", gpu_build)
        film = prt.Film(W, H)
        r.Init(film, scene, cam)
        info = r.bvh_info()
        assert info.built_on_device == gpu_build and info.n_nodes8 > 0 and info.depth8 <= 12
        if gpu_build:
            assert info.build_ms > 0
            rng = np.random.default_rng(23)
            o, d = util.random_rays(rng, 4000, center=(0, 0.5, 0), radius=11.0, spread=4.5)
            got = r.closest_hit(o, d)
            want = util.oracle_scene(scene).closest_hit(o, d, use_bvh=False, n_threads=8)
            assert util.hits_equal(got, want) == []
            assert (got["prim"] >= 2).sum() > 500
        r.ProgressiveRender(spp)
        r.download()
        films.append((film.accum.copy(), r.stats().rays_total))
    assert np.array_equal(films[0][0], films[1][0]) and films[0][1] == films[1][1]
    acc, wts, rays = util.oracle_scene(scene).render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=11, iterative=True,
                                                     use_bvh=True, n_threads=8)
    assert np.array_equal(films[0][0], acc) and films[0][1] == rays


@pytest.mark.parametrize("gpu_build", [0, 1])
def test_one_node_per_cache_line_layout_gives_the_same_hits_and_image(gpu_build):
    """prt_set_param("node_stride", 8): the 8-wide nodes in 128-B slots (what upload_scene picks by itself for trees far
    beyond the L2s, e.g. config C5) instead of packed 80-B records: same closest hits as the oracle's brute force, same
    image as the packed layout, for host-built and device-built trees and for a scene with placed copies."""
    mesh = prt.scenes.refined("bunny.ply", 12_000)
    W, H, spp, depth = 128, 72, 2, 5
    for scene, cam in ((prt.scenes.mesh_scene(mesh), prt.Camera(position=(1.5, 1.0, 2.5), width=W, height=H)),
                       (_instanced_scene(mesh, True), prt.Camera(position=(6.0, 4.0, 9.0), width=W, height=H))):
        films = []
        for stride in (8, 5):
            r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=5)
            r.set_param("gpu_build", gpu_build)
            r.set_param("node_stride", stride)
            film = prt.Film(W, H)
            r.Init(film, scene, cam)
            if stride == 8:
                rng = np.random.default_rng(29)
                o, d = util.random_rays(rng, 3000, center=(0, 0.5, 0), radius=9.0, spread=4.0)
                got = r.closest_hit(o, d)
                want = util.oracle_scene(scene).closest_hit(o, d, use_bvh=False, n_threads=8)
                assert util.hits_equal(got, want) == []
            r.ProgressiveRender(spp)
            r.download()
            films.append((film.accum.copy(), r.stats().rays_total))
        assert np.array_equal(films[0][0], films[1][0]) and films[0][1] == films[1][1]
        assert films[0][0].sum() > 0


def test_kernel_occupancy_report():
    """prt_kernel_occupancy: the static wavefront occupancy bench.py reports next to the roofline."""
    scene = prt.scenes.mesh_scene(prt.Mesh(prt.scenes.asset("bunny.ply")))
    r, _, _ = make_renderer(scene, 16, 16)
    o = r.kernel_occupancy()
    assert o.max_waves_per_cu == 32 and o.compute_units == 256
    assert 1 <= o.blocks_per_cu <= 8 and o.waves_per_cu == 4 * o.blocks_per_cu
    assert 64 <= o.vgprs <= 128 and 30_000 < o.lds_bytes_per_block <= 40_960
    assert o.resident_grid_blocks == o.blocks_per_cu * o.compute_units  # the persistent grid fills the chip exactly


def test_cpp_adapter_cli_renders_the_same_image(tmp_path):
    """The C++ host path (prt_render: reference-shaped adapter over the C-ABI, offline framebuffer dump) against the
    oracle: CORNELL 64x64, 2 spp, 3 segments, seed 7 -> PFM of mean radiance, bit-exact."""
    import os
    import subprocess
    exe = os.path.join(util.ROOT, "parallelraytracing_amd", "csrc", "prt_render")
    out = str(tmp_path / "frame")
    p = subprocess.run([exe, "--preset", "CORNELL", "--width", "64", "--height", "64", "--spp", "2", "--depth", "3",
                        "--seed", "7", "--out", out], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    raw = open(out + ".pfm", "rb").read()
    hdr = b"PF\n64 64\n-1.0\n"
    assert raw.startswith(hdr)
    img = np.frombuffer(raw[len(hdr):], "<f4").reshape(64, 64, 3)[::-1]  # PFM stores the bottom row first
    scene = prt.Scene("CORNELL")
    cam = prt.Camera(width=64, height=64)
    acc, wts, rays = util.oracle_scene(scene).render(cam.desc(), 64, 64, spp=2, max_depth=3, seed=7, iterative=True)
    assert np.array_equal(img, acc / wts[..., None])
    assert f"{rays} rays" in p.stdout
    ppm = open(out + ".ppm", "rb").read()
    assert ppm.startswith(b"P6\n64 64\n255\n") and len(ppm) == 13 + 64 * 64 * 3


# ---- properties that hold at any size -----------------------------------------------------------------------------------
def test_batching_and_samples_in_flight_do_not_change_the_image():
    scene = prt.Scene("DEFAULT")
    W, H, depth = 80, 48, 6
    ref_r, ref_f, _ = make_renderer(scene, W, H, max_depth=depth, seed=1)
    for _ in range(6):
        ref_r.ProgressiveRender()
    ref_r.download()
    for S, calls in [(1, [6]), (4, [6]), (3, [2, 4]), (8, [1, 5])]:
        r, f, _ = make_renderer(scene, W, H, max_depth=depth, seed=1)
        r.set_samples_in_flight(S)
        for c in calls:
            r.ProgressiveRender(c)
        r.download()
        assert np.array_equal(f.accum, ref_f.accum) and np.array_equal(f.weights, ref_f.weights), (S, calls)


def test_switching_between_one_sample_and_multi_sample_calls_on_a_mesh_scene():
    """A batch of ONE sample skips k_primary_hit (its per-pixel records are blanked once and stay blank), a larger batch
    writes them again: any order of call sizes gives the frame and the ray counts of the oracle, with exact grids (the
    count read-backs on their own stream) as well."""
    mesh = prt.scenes.refined("bunny.ply", 30_000)
    scene = prt.scenes.mesh_scene(mesh)
    W, H, depth, seed = 96, 54, 5, 3
    cam = prt.Camera(position=(2.0, 1.5, 3.0), width=W, height=H)
    osc = util.oracle_scene(scene)
    acc, wts, rays = osc.render(cam.desc(), W, H, spp=10, max_depth=depth, seed=seed, iterative=True, use_bvh=True, n_threads=8)
    for exact in (0, 2):
        r, film, _ = make_renderer(scene, W, H, max_depth=depth, seed=seed, cam=cam)
        r.set_param("exact_grids", exact)
        r.set_samples_in_flight(8)
        for calls in (1, 5, 1, 1, 2):
            r.ProgressiveRender(calls)
        r.download()
        assert np.array_equal(film.accum, acc) and np.array_equal(film.weights, wts), exact
        st = r.stats()
        assert st.rays_total == rays and st.rays_per_depth[0] == 10 * W * H, exact
        # a measurement run counts into its own set of counters: the context's stay what they were
        r.measure_traversal(sample=3)
        st2 = r.stats()
        assert st2.rays_total == rays and list(st2.rays_per_depth) == list(st.rays_per_depth)


def test_partitioned_render_equals_single_render():
    """world_size 1 vs 3 ranks (three contexts on the one GPU): same film bits after gather + resolve."""
    import torch
    scene = prt.Scene("MATERIAL_TEST")
    W, H, depth, spp = 100, 52, 5, 3
    r1, f1, _ = make_renderer(scene, W, H, max_depth=depth, seed=6)
    r1.ProgressiveRender(spp)
    r1.download()
    world = 3
    payloads = []
    rs = []
    for rank in range(world):
        r, f, _ = make_renderer(scene, W, H, max_depth=depth, seed=6, rank=rank, world_size=world)
        r.ProgressiveRender(spp)
        rs.append(r)
        payloads.append(prt.dist.local_payload_tensor(r, "cuda:0"))  # zero-copy view of prt_film_local
    gathered = torch.cat(payloads).contiguous()
    rgb = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda:0")
    wts = torch.zeros(H * W, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    rs[0].film_resolve(gathered.data_ptr(), rgb.data_ptr(), wts.data_ptr())
    rs[0].synchronize()
    assert np.array_equal(rgb.cpu().numpy().reshape(H, W, 3), f1.accum)
    assert np.array_equal(wts.cpu().numpy().reshape(H, W), f1.weights)
    total = sum(r.stats().rays_total for r in rs)
    assert total == r1.stats().rays_total


def test_film_clear_and_progressive_accumulation():
    scene = prt.Scene("CORNELL")
    r, f, _ = make_renderer(scene, 64, 64, max_depth=3)
    r.ProgressiveRender(2)
    r.download()
    a2 = f.accum.copy()
    assert (f.weights == 2).all()
    f.Clear()
    r.frame_index = 0
    r.ProgressiveRender(2)
    r.download()
    assert np.array_equal(f.accum, a2)
    r.ProgressiveRender(1)
    r.download()
    assert (f.weights == 3).all() and (f.accum >= a2).all()


def test_tonemap_matches_oracle_within_one_lsb():
    scene = prt.Scene("DEFAULT")
    r, f, _ = make_renderer(scene, 128, 72, max_depth=5)
    r.ProgressiveRender(4)
    r.download()
    disp = r.UpdateDisplay().astype(np.int32)
    want = orc.tonemap(f.accum, f.weights).astype(np.int32)
    diff = np.abs(disp - want)
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01  # powf on the GPU vs glibc: rare 1-LSB flips
    assert (disp[..., 3] == 255).all()


def test_error_paths():
    r = prt.HipWavefrontRenderer(device=0)
    with pytest.raises(prt.PrtError):
        r.ProgressiveRender()  # nothing set yet
    bad = prt.Scene(preset=None)
    bad.AddQuad(1, 1, material=3)  # material out of range
    with pytest.raises(prt.PrtError):
        r.Init(prt.Film(8, 8), bad, prt.Camera(width=8, height=8))
    with pytest.raises(prt.PrtError):
        prt.HipWavefrontRenderer(device=99)


@pytest.mark.parametrize("W,H", [(1920, 1080)])
def test_full_size_properties(W, H):
    """At BASELINE's full frame size the oracle is too slow for a whole image; check size-independent
    properties instead: weights, ray-count bounds, determinism, and an oracle crop."""
    scene = prt.Scene("DEFAULT")
    depth, seed = 5, 0
    r, f, cam = make_renderer(scene, W, H, max_depth=depth, seed=seed)
    r.ProgressiveRender(2)
    r.download()
    a = f.accum.copy()
    st = r.stats()
    assert (f.weights == 2).all()
    assert st.rays_per_depth[0] == 2 * W * H
    assert all(st.rays_per_depth[d] >= st.rays_per_depth[d + 1] for d in range(depth))
    assert st.rays_per_depth[depth] == 0 and np.isfinite(a).all() and (a >= 0).all()
    r2, f2, _ = make_renderer(scene, W, H, max_depth=depth, seed=seed)
    r2.set_samples_in_flight(2)
    r2.ProgressiveRender(2)
    r2.download()
    assert np.array_equal(f2.accum, a)
    rect = (900, 500, 1028, 564)
    acc, wts, _ = util.oracle_scene(scene).render(cam.desc(), W, H, spp=2, max_depth=depth, seed=seed, iterative=True,
                                                  n_threads=8, rect=rect)
    x0, y0, x1, y1 = rect
    assert np.array_equal(acc[y0:y1, x0:x1], a[y0:y1, x0:x1])


# ---- refit (prt_refit_meshes: new vertex positions over the existing 8-wide topology, on the device) --------------------
def _deformed(mesh, amp, seed):
    """The same topology with every vertex moved: a smooth bend plus per-vertex noise of `amp`."""
    v = mesh.GetVertices().astype(np.float64)
    rng = np.random.default_rng(seed)
    v[:, 0] += amp * 4.0 * np.sin(3.0 * v[:, 1])
    v[:, 2] += amp * 4.0 * np.cos(2.0 * v[:, 0])
    v += rng.normal(size=v.shape) * amp
    n = mesh.GetNormals()
    return prt.Mesh(vertices=v.astype(np.float32), normals=n, indices=mesh.GetIndices())


@pytest.mark.parametrize("builder,target", [(0, 10_000), (1, 30_000), (2, 30_000), (0, 120_000)])
def test_refitted_tree_is_valid_and_renders_the_deformed_mesh_bit_exact(builder, target):
    """The tree built for the ORIGINAL mesh, refitted on the device to a deformed copy: structurally valid (every quantized
    box contains what is below it), closest hits and image bit-exact against the oracle on the deformed mesh (the oracle
    scans / builds its own tree from the new geometry), and equal to a fresh build of the deformed scene."""
    base = prt.scenes.refined("bunny.ply", target)
    moved = _deformed(base, 0.01, 5)
    W, H, spp, depth = 96, 54, 2, 5
    cam = prt.Camera(position=(2.0, 1.5, 3.0), width=W, height=H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=3)
    r.set_param("gpu_build", builder)
    film = prt.Film(W, H)
    r.Init(film, prt.scenes.mesh_scene(base), cam)
    r.ProgressiveRender(1)  # (the renderer has worked with the old geometry)
    scene2 = prt.scenes.mesh_scene(moved)
    r.Refit(scene2)
    n8 = r.bvh_read8()
    _, tris = r.bvh_read()
    fill, levels = util.check_bvh8(n8, tris)
    assert levels == r.bvh_info().depth8
    prim = tris[:, 3].view(np.uint32).astype(np.int64) - len(scene2.primitives)
    assert sorted(prim.tolist()) == list(range(moved.n_triangles))
    osc = util.oracle_scene(scene2)
    o, d = _mesh_rays(np.random.default_rng(7), 5000)
    got = r.closest_hit(o, d)
    want = osc.closest_hit(o, d, use_bvh=target > 10_000, n_threads=8)  # (the small case against the linear scan)
    assert util.hits_equal(got, want) == []
    assert (got["prim"] >= 2).sum() > 300
    film.Clear()
    r.frame_index = 0
    r.reset_stats()
    r.ProgressiveRender(spp)
    r.download()
    acc, wts, rays = osc.render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=3, iterative=True, use_bvh=True, n_threads=8)
    assert np.array_equal(film.accum, acc) and r.stats().rays_total == rays
    # a fresh build of the deformed scene gives the same frame
    r2 = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=3)
    film2 = prt.Film(W, H)
    r2.Init(film2, scene2, cam)
    r2.ProgressiveRender(spp)
    r2.download()
    assert np.array_equal(film2.accum, film.accum)


def test_refit_rejects_another_topology_and_placed_copies():
    base = prt.scenes.refined("bunny.ply", 12_000)   # (bunny.ply itself has 10,000 triangles)
    r = prt.HipWavefrontRenderer(device=0)
    film = prt.Film(16, 16)
    r.Init(film, prt.scenes.mesh_scene(base), prt.Camera(width=16, height=16))
    other = prt.scenes.refined("bunny.ply", 15_000)
    with pytest.raises(prt.PrtError):
        r.Refit(prt.scenes.mesh_scene(other))
    # the scene the renderer holds is untouched by the refused call
    o, d = _mesh_rays(np.random.default_rng(3), 500)
    assert util.hits_equal(r.closest_hit(o, d), util.oracle_scene(prt.scenes.mesh_scene(base)).closest_hit(o, d, use_bvh=False, n_threads=8)) == []


def test_device_builder_copes_with_thousands_of_coincident_triangles():
    """a run of clusters with identical boxes made PLOC's nearest-neighbour search merge ONE pair per pass (ties
    to the lowest position), so > 8 k coincident triangles ran into the pass guard and a valid mesh was refused.  The
    search now ranks pairs by a symmetric key (area, distance, parity, position): such a run halves per pass; and if a
    device builder ever gives up, prt_set_scene falls back to the host builder instead of failing."""
    n = 12_000
    tri = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.2]], np.float32)
    verts = np.tile(tri, (n, 1))
    # (a few distinct triangles around them, so that the tree has something else to separate)
    extra = np.random.default_rng(2).uniform(-1, 1, (300, 3)).astype(np.float32)
    verts = np.concatenate([verts, extra])
    idx = np.arange(len(verts), dtype=np.uint32).reshape(-1, 3)
    nrm = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (len(verts), 1))
    mesh = prt.Mesh(vertices=verts, normals=nrm, indices=idx)
    scene = prt.scenes.mesh_scene(mesh)
    for mode in (1, 2):
        r = prt.HipWavefrontRenderer(device=0)
        r.set_param("gpu_build", mode)
        film = prt.Film(16, 16)
        r.Init(film, scene, prt.Camera(width=16, height=16))  # must not raise
        assert r.bvh_info().n_nodes8 > 0
        o, d = _mesh_rays(np.random.default_rng(4), 1500)
        got = r.closest_hit(o, d)
        want = util.oracle_scene(scene).closest_hit(o, d, use_bvh=False, n_threads=8)
        assert util.hits_equal(got, want) == []


# ---- the PATH instance of the traversal kernel: whole paths in one launch (small batches, one sample per call) -------------
@pytest.mark.parametrize("sampling", [None, {"jitter": 1}, {"jitter": 1, "rr_depth": 2, "clamp": 4.0}])
@pytest.mark.parametrize("mode", ["one-sample-calls", "forced-batch"])
def test_path_kernel_frames_are_bit_exact(mode, sampling):
    """prt_set_param("path_kernel", 1): a batch of ONE sample runs as a single launch in which every lane carries a whole path
    (primary ray, walks, shade steps); 2: any small batch.  The frame must equal the oracle's (throughput form) and the frame
    of the raygen / traverse / shade pipeline bit for bit, with the ray counts."""
    mesh = prt.scenes.refined("bunny.ply", 30_000)
    scene = prt.scenes.mesh_scene(mesh)
    W, H, spp, depth = 120, 68, 3, 6
    cam = prt.Camera(position=(2.0, 1.5, 3.0), width=W, height=H)
    frames = {}
    for pk in (0, 1 if mode == "one-sample-calls" else 2):
        r, film, _ = make_renderer(scene, W, H, max_depth=depth, seed=11, cam=cam)
        r.set_param("path_kernel", pk)
        sp = r.set_sampling(**sampling) if sampling else None
        if mode == "one-sample-calls":
            for _ in range(spp):
                r.ProgressiveRender()  # one sample per call: the reference's contract
        else:
            r.ProgressiveRender(spp)
        r.download()
        st = r.stats()
        frames[pk] = (film.accum.copy(), film.weights.copy(), int(st.rays_total), [int(st.rays_per_depth[d]) for d in range(depth)])
    a, b = frames[0], frames[1 if mode == "one-sample-calls" else 2]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3]
    osc = util.oracle_scene(scene)
    acc, wts, rays = osc.render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=11, iterative=True, use_bvh=True, n_threads=8, sampling=sp)
    assert np.array_equal(b[0], acc) and np.array_equal(b[1], wts) and b[2] == rays


def test_path_kernel_partial_tiles_and_partition():
    """A frame whose size is not a multiple of the 8x8 tiles, rendered by three contexts (ranks) one sample per call: the
    path instance must leave out-of-image pixels of partial tiles alone and honour the tile map."""
    mesh = prt.scenes.refined("bunny.ply", 12_000)
    scene = prt.scenes.mesh_scene(mesh)
    W, H, spp, depth = 53, 37, 2, 5
    cam = prt.Camera(position=(2.0, 1.5, 3.0), width=W, height=H)
    osc = util.oracle_scene(scene)
    acc, wts, rays = osc.render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=2, iterative=True, use_bvh=True, n_threads=8)
    total = np.zeros_like(acc)
    n_rays = 0
    for rank in range(3):
        film = prt.Film(W, H)
        r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=2, rank=rank, world_size=3)
        r.Init(film, scene, cam)
        r.set_param("path_kernel", 1)
        for _ in range(spp):
            r.ProgressiveRender()
        r.download()
        total += film.accum
        n_rays += int(r.stats().rays_total)
    assert np.array_equal(total, acc) and n_rays == rays
