"""Load balance of a P-way assembly with ray-count exchange, measured rank by rank on one GPU: every rank's first launch (its own
x own pairs + its share of the pairs it has with other ranks) -- kernel ms and traced pairs.  The slowest rank sets the time
of the real P-GPU assembly (the all-to-all waits for it).   NPATCH (65536), NSHARD (8)"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from daisyriot_amd import api, scenes
N = int(os.environ.get("NPATCH", "65536")); P = int(os.environ.get("NSHARD", "8"))
sc = scenes.cornell_box(N, S=8)
uv = scenes.visibility_samples(50)
ms, traced = [], []
for r in range(P):
    with api.Context(0) as c:
        c.set_shard(r, P); c.comm_manual()
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        c.assemble_split(uv)
        i = c.info()
        ms.append(i.last_assemble_ms); traced.append(int(i.pairs_traced))
        print("RANK", r, "first launch ms", round(i.last_assemble_ms, 1), "pairs traced", i.pairs_traced, flush=True)
ms = np.array(ms); traced = np.array(traced, np.float64)
print("BALANCE N", N, "P", P, "ms max/mean %.3f" % (ms.max() / ms.mean()), "pairs max/mean %.3f" % (traced.max() / traced.mean()),
      "sum ms", round(ms.sum(), 1), "max ms", round(ms.max(), 1), "ideal 8-GPU speed-up over the sum %.2f" % (ms.sum() / ms.max()))
