"""One rank of a multi-process run of the hot path on ONE GPU, with the loop-back RCCL stand-in (DR_RCCL_LIB).
   python rank_main.py rank world N S rule id_file out_dir      (started by tests/test_gpu_fake_rccl.py; no torch here)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from daisyriot_amd import api, scenes  # noqa: E402

rank, world, N, S, rule = (int(x) for x in sys.argv[1:6])
id_file, out_dir = sys.argv[6], sys.argv[7]
sc = scenes.cornell_box(N, S=S, fluorescent=(S >= 8))
uv = scenes.visibility_samples(50)
E = sc.emission(7.0)
c = api.Context(0)
c.set_shard(rank, world)
if rank == 0:
    np.save(id_file + ".tmp.npy", api.comm_unique_id())
    os.rename(id_file + ".tmp.npy", id_file)
t0 = time.time()
while not os.path.exists(id_file):
    time.sleep(0.01)
    assert time.time() - t0 < 120, "no unique id from rank 0"
c.comm_init(np.load(id_file), rank, world)
assert c.comm_info() == (rank, world)
c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
c.assemble(uv, rule=rule, keep_visibility=True)          # world > 1 and a communicator: the ray-count exchange path
row0, nrows, rpr = c.shard()
out = {"row0": row0, "nrows": nrows, "traced": int(c.info().pairs_traced)}
out["F"] = c.read_rows(row0, nrows)
out["V"] = c.read_visibility(row0, nrows)
c.solver_init(E, sc.M, sc.mat_of_patch)
tot = c.step(4, want_sum=True)
B, R = c.read()
out["B4"], out["R4"], out["sum4"] = B[row0:row0 + nrows], R, tot
c.reset()
thr, per_bin = (1e-3, True) if S == 3 else (0.5, False)
out["iters"] = c.converge(thr, per_bin=per_bin, max_iters=300)
B, R = c.read()
out["Bc"], out["Rc"] = B[row0:row0 + nrows], R
np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
c.close()
