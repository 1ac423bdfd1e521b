import sys, os, time
sys.path.insert(0, '.')
import numpy as np
from daisyriot_amd import api, scenes
if os.environ.get("DR_LIB"):
    api.LIB_PATH = os.environ["DR_LIB"]          # A/B of two builds of the library
N = int(os.environ.get("NPATCH", "16384"))
sc = scenes.cornell_box(N, S=8)
uv = scenes.visibility_samples(50)
c = api.Context(0)
c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
c.assemble(uv)
i = c.info()
print("RES", N, "ms", round(i.last_assemble_ms,1), "traced", i.pairs_traced, "bvh_ms", round(i.last_bvh_ms,2), flush=True)
