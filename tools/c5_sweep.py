import sys, os, time, subprocess
code = r'''
import sys, os, time
sys.path.insert(0, '.')
import numpy as np
from daisyriot_amd import api, scenes
N = int(os.environ.get("NPATCH", "262144")); S = int(os.environ.get("NBINS", "8")); P = int(os.environ.get("NSHARD", "8"))
sc = scenes.cornell_box(N, S=S, fluorescent=True)
E = sc.emission(7.0)
c = api.Context(0)
c.set_shard(0, P); c.comm_manual()
c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
c.integrand_only()
c.solver_init(E, sc.M, sc.mat_of_patch)
c.step(3); c.synchronize(); c.profile(True); c.profile_reset(); c.step(20); c.synchronize()
i = c.info(); ms = i.sweep_ms_total/i.sweep_launches
b = 4*i.nrows*N + 4*N*S + 12*i.nrows*S
print("RES skew", os.environ.get("DR_SWEEP_SKEW","def"), "ksplit", os.environ.get("DR_SWEEP_KSPLIT","auto"), "sweep ms", round(ms,3), "GB/s", round(b/ms/1e6), flush=True)
'''
open('tools/_c5s.py','w').write(code)
for a in sys.argv[1:]:
    skew, ks = a.split(':')
    env = dict(os.environ)
    if skew != 'd': env["DR_SWEEP_SKEW"] = skew
    if ks != 'a': env["DR_SWEEP_KSPLIT"] = ks
    r = subprocess.run([sys.executable, 'tools/_c5s.py'], env=env, capture_output=True, text=True, timeout=400)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAIL " + r.stderr[-400:]), flush=True)
