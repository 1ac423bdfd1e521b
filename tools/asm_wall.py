"""wall time of dr_formfactors_assemble against its kernel time (hipEvents): what the call costs beyond the kernel"""
import sys, os, time
sys.path.insert(0, '.')
import numpy as np
from daisyriot_amd import api, scenes
N = int(os.environ.get("NPATCH", "65536"))
sc = scenes.cornell_box(N, S=8)
uv = scenes.visibility_samples(50)
c = api.Context(0)
t0 = time.perf_counter(); c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n); t1 = time.perf_counter()
for rep in range(2):
    t2 = time.perf_counter(); c.assemble(uv); t3 = time.perf_counter()
    i = c.info()
    print("RES", N, "set_mesh wall ms", round((t1 - t0) * 1e3, 1), "bvh ms", round(i.last_bvh_ms, 2), "assemble wall ms", round((t3 - t2) * 1e3, 1),
          "kernel ms", round(i.last_assemble_ms, 1), flush=True)
