"""Converge-mode vs step-mode ms per pass on rank 0 of P of the 64k problem (rank-local, DR_NO_COMM): the convergence
test is fused into the pass (sums in the residual tails, decision on the device), so queuing K passes through
dr_solver_converge should cost what K dr_solver_step passes cost.   python tools/conv_bench.py 8 1"""
import os, subprocess, sys
code = r'''
import sys, os, time
sys.path.insert(0, '.')
import numpy as np
from daisyriot_amd import api, scenes
N = 65536; S = 8; K = 200
P = int(os.environ["NSHARD"])
sc = scenes.cornell_box(N, S=S)
E = sc.emission(7.0)
c = api.Context(0)
c.set_shard(0, P)
c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
c.integrand_only()
c.solver_init(E, sc.M, sc.mat_of_patch)
c.step(10); c.synchronize()
t = time.perf_counter(); c.step(K); c.synchronize(); t_step = (time.perf_counter() - t) / K
res = []
for every in (1, 8, 32):
    c.set_check_interval(every)
    c.reset(); c.synchronize()
    t = time.perf_counter(); it = c.converge(-1.0, per_bin=False, max_iters=K); t_conv = (time.perf_counter() - t) / K
    assert it == K
    res.append("check every %d: %.4f ms/pass (%+.1f %%)" % (every, t_conv * 1e3, (t_conv / t_step - 1) * 100))
print("RES P", P, "rows", c.info().nrows, "step %.4f ms/pass;" % (t_step * 1e3), "; ".join(res), flush=True)
'''
open('tools/_cv.py', 'w').write(code)
for P in sys.argv[1:]:
    env = dict(os.environ, NSHARD=P, DR_NO_COMM="1")
    r = subprocess.run([sys.executable, 'tools/_cv.py'], env=env, capture_output=True, text=True, timeout=400)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAIL " + r.stderr[-600:]), flush=True)
