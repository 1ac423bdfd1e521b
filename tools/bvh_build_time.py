"""BVH build time (dr_info.last_bvh_ms: stream events around the whole build) and depth of the three trees at several sizes.
   python tools/bvh_build_time.py [sizes...]"""
import sys
sys.path.insert(0, '.')
from daisyriot_amd import api, scenes
sizes = [int(x) for x in sys.argv[1:]] or [6400, 16384, 65536, 262144]
for n in sizes:
    sc = scenes.cornell_box(n, S=3)
    for name, opts in (("lbvh", dict(tree=api.TREE_LBVH)), ("sah-device", dict(tree=api.TREE_SAH)), ("sah-host", dict(tree=api.TREE_SAH, sah_on_host=1))):
        ms = []
        for rep in range(3):
            with api.Context(0) as c:
                c.set_options(**opts)
                c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
                i = c.info()
                ms.append(i.last_bvh_ms)
        print("BVH", n, name, "ms", " ".join("%.2f" % m for m in ms), "depth", i.tree_depth, "nodes", i.n_bvh_nodes, flush=True)
