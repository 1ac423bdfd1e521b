import sys, os, time
sys.path.insert(0, '.')
import numpy as np
from daisyriot_amd import api, scenes
N = int(os.environ.get("NPATCH", "65536")); S = 8
sc = scenes.cornell_box(N, S=S)
uv = scenes.visibility_samples(50)
E = sc.emission(7.0)
c = api.Context(0)
c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
c.assemble(uv)
c.solver_init(E, sc.M, sc.mat_of_patch)
for skip in (False, True, False, True):
    c.skip_zero_blocks(skip)
    c.reset(); c.step(5); c.synchronize()
    c.profile(True); c.profile_reset()
    t = time.perf_counter(); c.step(50); c.synchronize(); wall = (time.perf_counter() - t) / 50
    i = c.info(); c.profile(False)
    ms = i.sweep_ms_total / i.sweep_launches
    print("skip", skip, "kernel ms", round(ms, 4), "wall ms", round(wall * 1e3, 4), "blocks", i.blocks_nonzero, "/", i.blocks_total, flush=True)
