"""One-off fuzz of the visibility definition: random triangle soups (sizes, scales, offsets, slivers, K) on the GPU
against the brute-force oracle (every triangle tested for every ray).  Exact equality of ray counts and F expected.
   python tools/fuzz_visibility.py [n_scenes] [first_seed]      FUZZ_SCALE_EXP=lo,hi: decimal exponents of the scene scale (-3,3)"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from daisyriot_amd import api, scenes
from oracle import binding as ob

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
SCALE_EXP = tuple(float(x) for x in os.environ.get("FUZZ_SCALE_EXP", "-3,3").split(","))
bad = 0
t0 = time.time()
for seed in range(seed0, seed0 + n_scenes):
    rs = np.random.RandomState(seed)
    n = int(rs.randint(40, 420))
    scale = float(10.0 ** rs.uniform(*SCALE_EXP))
    offset = float(rs.choice([0.0, 0.0, 3.0, 250.0, -4000.0])) * scale
    K = int(rs.choice([1, 7, 32, 50, 50, 64, 100]))
    rule = int(rs.randint(0, 2))
    c0 = rs.random_sample((n, 1, 3)) * 2 - 1
    size = rs.random_sample((n, 1, 1)) ** 3 * rs.choice([0.2, 0.6, 1.5]) + 1e-3
    tri = (c0 + (rs.random_sample((n, 3, 3)) - 0.5) * size)
    if rs.rand() < 0.5:                                   # some axis-aligned sheets: rays parallel to slabs
        k = rs.randint(0, 3)
        tri[::3, :, k] = np.round(tri[::3, :1, k], 1)
    tri[::17, 2] = tri[::17, 1] + (tri[::17, 1] - tri[::17, 0]) * 1e-4       # slivers
    tri = tri * scale + offset * np.array([1.0, -0.4, 0.25])
    v = tri.reshape(-1, 3).astype(np.float32)
    tv = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    nrm = rs.normal(size=(32, 3)).astype(np.float32)
    tn = rs.randint(0, 32, size=(n, 3)).astype(np.int32)
    uv = scenes.visibility_samples(K, seed=seed)
    with api.Context(0) as c:
        c.set_mesh(v, nrm, tv, tn)
        c.assemble(uv, rule=rule, keep_visibility=True)
        vis, F = c.read_visibility(0, n), c.read_rows(0, n)
    Fo, viso, _ = ob.assemble_rows(ob.Mesh(v, nrm, tv, tn), uv, rule=rule, bvh=False)
    ok = np.array_equal(vis, viso) and np.array_equal(F.view(np.uint32), Fo.view(np.uint32))
    bad += not ok
    print("seed", seed, "n", n, "scale %.3g" % scale, "offset %.3g" % offset, "K", K, "rule", rule,
          "traced", int((viso != 255).sum()), "OK" if ok else "MISMATCH %d" % int((vis != viso).sum()), flush=True)
print("scenes", n_scenes, "mismatching", bad, "in %.0f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
