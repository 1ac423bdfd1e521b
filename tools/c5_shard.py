import sys, os, time
sys.path.insert(0, '.')
import numpy as np
from daisyriot_amd import api, scenes
N = 262144; S = 8; P = 8
t=time.time(); sc = scenes.cornell_box(N, S=S, fluorescent=True); print("scene", round(time.time()-t,1), flush=True)
uv = scenes.visibility_samples(50)
E = sc.emission(7.0)
c = api.Context(0)
c.set_shard(0, P); c.comm_manual()
t=time.time(); c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n); print("set_mesh", round(time.time()-t,2), "bvh ms", c.info().last_bvh_ms, "nodes", c.info().n_bvh_nodes, flush=True)
t=time.time(); c.assemble(uv); i=c.info(); print("assemble shard s", round(time.time()-t,1), "kernel ms", round(i.last_assemble_ms), "traced", i.pairs_traced, "F GB", i.bytes_F/1e9, flush=True)
F = c.read_rows(1000, 2); print("row sums", F.sum(1), flush=True)
c.solver_init(E, sc.M, sc.mat_of_patch)
c.step(3); c.synchronize(); c.profile(True); c.profile_reset(); c.step(20); c.synchronize()
i = c.info(); ms = i.sweep_ms_total/i.sweep_launches
b = 4*i.nrows*N + 4*N*S + 12*i.nrows*S
print("sweep ms", round(ms,3), "GB/s", round(b/ms/1e6), flush=True)
