"""Dump the device BVH and triangle records of a synthetic Cornell box (for offline study of the walk: tools/sim_walk.py)."""
import sys, os
sys.path.insert(0, '.')
import numpy as np
from daisyriot_amd import api, scenes
N = int(os.environ.get("NPATCH", "16384"))
sc = scenes.cornell_box(N, S=8)
c = api.Context(0)
c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
bvh = c.read_bvh()
tri = c.read_array(0, 64 * N).view(np.float32).reshape(N, 16)
tris = c.read_array(1, 64 * N).view(np.float32).reshape(N, 16)
np.savez_compressed("gpurun_out/bvh_%d.npz" % N, lo=bvh["lo"], hi=bvh["hi"], skip=bvh["skip"], tri=bvh["tri"], trirec=tri, trisorted=tris)
print("nodes", len(bvh))
