import sys, os, time
sys.path.insert(0, '.')
import numpy as np
from daisyriot_amd import api, scenes
N = int(os.environ.get("NPATCH", "32768"))
c = api.Context(0)
sc0 = scenes.cornell_box(N, S=8)
c.set_mesh(sc0.vertices, sc0.normals, sc0.tri_v, sc0.tri_n)
c.integrand_only()
for S in [int(x) for x in sys.argv[1:]]:
    sc = scenes.cornell_box(N, S=S)
    E = sc.emission(7.0)
    c.solver_init(E, sc.M, sc.mat_of_patch)
    c.step(3); c.synchronize()
    c.profile(True); c.profile_reset()
    c.step(20); c.synchronize()
    i = c.info(); c.profile(False)
    ms = i.sweep_ms_total / i.sweep_launches
    b = 4*N*N + 4*N*S + 12*N*S
    print("S", S, "ms", round(ms,4), "GB/s", round(b/ms/1e6,1), "frac", round(b/ms/1e6/8000,3), flush=True)
