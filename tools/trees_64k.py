"""one-off: the Morton tree and the SAH topology give the same F and ray counts at 65 536 patches (24 sampled blocks of 256 rows, hashed)"""
import sys, os, hashlib
sys.path.insert(0, '.')
import numpy as np
from daisyriot_amd import api, scenes
sc = scenes.cornell_box(65536, S=8)
uv = scenes.visibility_samples(50)
rows = np.random.RandomState(3).randint(0, 65536 - 256, 24)
got = {}
for tree in ("lbvh", "sah"):
    os.environ["DR_BVH"] = tree
    c = api.Context(0)
    c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
    c.assemble(uv, keep_visibility=True)
    h = hashlib.sha256()
    for r in rows:
        h.update(c.read_rows(int(r), 256).view(np.uint32).tobytes()); h.update(c.read_visibility(int(r), 256).tobytes())
    got[tree] = (h.hexdigest(), c.info().pairs_traced, round(c.info().last_assemble_ms))
    c.close()
print(got, "EQUAL" if got["lbvh"][:2] == got["sah"][:2] else "DIFFERENT")
