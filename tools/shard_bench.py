import os, subprocess, sys
code = r'''
import sys, os, time
sys.path.insert(0, '.')
import numpy as np
from daisyriot_amd import api, scenes
N = 65536; S = 8
P = int(os.environ["NSHARD"])
sc = scenes.cornell_box(N, S=S)
E = sc.emission(7.0)
c = api.Context(0)
c.set_shard(0, P)
c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
c.integrand_only()
c.solver_init(E, sc.M, sc.mat_of_patch)
c.step(5); c.synchronize()
c.profile(True); c.profile_reset()
t=time.perf_counter(); c.step(50); c.synchronize(); wall=(time.perf_counter()-t)/50
i = c.info()
ms = i.sweep_ms_total / i.sweep_launches
b = 4*i.nrows*N + 4*N*S + 12*i.nrows*S
print("RES P", P, "ksplit", os.environ.get("DR_SWEEP_KSPLIT","auto"), "nrows", i.nrows, "kernel ms", round(ms,4), "wall ms/step", round(wall*1e3,4), "GB/s", round(b/ms/1e6,1), flush=True)
'''
open('tools/_sh.py','w').write(code)
for a in sys.argv[1:]:
    P, ks = a.split(':')
    env = dict(os.environ, NSHARD=P, DR_NO_COMM="1")
    if ks != 'a': env["DR_SWEEP_KSPLIT"] = ks
    r = subprocess.run([sys.executable, 'tools/_sh.py'], env=env, capture_output=True, text=True, timeout=300)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAIL " + r.stderr[-400:]), flush=True)
