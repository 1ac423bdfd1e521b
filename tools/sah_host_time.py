"""host-only: time of the SAH topology build (dr_debug_sah_topology) against the thread count"""
import time, numpy as np, os, sys
sys.path.insert(0, '.')
from daisyriot_amd import api, scenes
for N in (65536, 262144):
    sc = scenes.cornell_box(N, S=3)
    v = sc.vertices[sc.tri_v]
    b = np.concatenate([v.min(axis=1), v.max(axis=1)], axis=1).astype(np.float32)
    api.sah_topology(b)
    for th in ("1", "2", "4", "8", "16"):
        os.environ["DR_SAH_THREADS"] = th
        ts = []
        for r in range(3):
            t0 = time.perf_counter(); t = api.sah_topology(b); ts.append(time.perf_counter() - t0)
        print("N", N, "threads", th, "ms", round(min(ts) * 1e3, 1), flush=True)
