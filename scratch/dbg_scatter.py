import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, util
from util import orc, prt
scene = prt.Scene("DEFAULT")
scene.AddMetal((0.9, 0.8, 0.7), 0.0); scene.AddMetal((0.9, 0.8, 0.7), 0.6); scene.AddDielectric(1.5); scene.AddDielectric(1.0)
film = prt.Film(8,8); r = prt.HipWavefrontRenderer(device=0); r.Init(film, scene, prt.Camera(width=8,height=8))
rng = np.random.default_rng(11); n = 6000
hits = np.zeros(n, dtype=prt.capi.HIT_DTYPE)
nrm = np.stack([prt.glm_normalize(x) for x in rng.normal(size=(n, 3)).astype(np.float32)])
ind = np.stack([prt.glm_normalize(x) for x in rng.normal(size=(n, 3)).astype(np.float32)])
flip = (np.einsum("ij,ij->i", nrm, ind) > 0); nrm[flip] *= -1
hits["prim"] = 0; hits["front_face"] = rng.integers(0, 2, n); hits["material_id"] = rng.integers(0, len(scene.materials), n)
hits["position"] = rng.uniform(-5, 5, (n, 3)).astype(np.float32); hits["normal"] = nrm
state = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
sc, att, em, oo, od, st = r.scatter(ind, hits, state)
bad = 0
from collections import Counter
cnt = Counter()
for i in range(n):
    m = scene.materials[int(hits["material_id"][i])]
    w = orc.scatter(m, ind[i], hits[i], int(state[i]))
    ok = bool(sc[i]) == w[0] and np.array_equal(att[i], w[1]) and np.array_equal(em[i], w[2]) and np.array_equal(oo[i], w[3]) and np.array_equal(od[i], w[4]) and int(st[i]) == w[5]
    if not ok:
        cnt[(m.type, bool(sc[i]), w[0])] += 1
        if bad < 6:
            print(i, 'type', m.type, 'gpu sc', sc[i], 'orc sc', w[0], '\n  pos', hits['position'][i], '\n  oo', oo[i], w[3], '\n  od', od[i], w[4], '\n att', att[i], w[1], 'rng', st[i], w[5])
        bad += 1
print('bad', bad, cnt)
