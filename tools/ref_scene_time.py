import sys, time
sys.path.insert(0,'.')
import numpy as np
from daisyriot_amd import api, scenes
uv = scenes.visibility_samples(50)
for name, S in (("cornellbox_blacklight", 3), ("colorballs", 8)):
    g = np.load("tests/golden/scene_%s.npz" % name)
    N = g["tri_v"].shape[0]
    c = api.Context(0)
    c.set_mesh(g["vertices"], g["normals"], g["tri_v"], g["tri_n"])
    c.assemble(uv); i = c.info()
    E = np.zeros((N, S), np.float32); E[g["Ke"][g["mat"]].sum(1) > 0] = 7.0
    M = np.stack([np.eye(S, dtype=np.float32) * 0.7] * g["Kd"].shape[0])
    c.solver_init(E, M, g["mat"])
    c.step(5); c.synchronize(); c.profile(True); c.profile_reset(); c.step(200); c.synchronize()
    j = c.info()
    t = time.perf_counter(); it = c.converge(1e-4, per_bin=True, max_iters=100000); dt = time.perf_counter() - t
    print(name, "N", N, "bvh ms", round(i.last_bvh_ms, 3), "assemble ms", round(i.last_assemble_ms, 1), "pairs traced", i.pairs_traced,
          "pass us", round(1e3 * j.sweep_ms_total / j.sweep_launches, 1), "converge passes", it, "in ms", round(dt * 1e3, 1), flush=True)
    c.close()
