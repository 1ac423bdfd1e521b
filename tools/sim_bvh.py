# build alternative BVHs over the same triangles and count walk visits with the simulator's machinery
import numpy as np, sys, os
sys.setrecursionlimit(100000)
exec(open("tools/sim_walk.py").read().split("rs = np.random.RandomState(1)")[0])
gl, gh = ts[:N,10:13].astype(np.float64), ts[:N,13:16].astype(np.float64)
pad = float((gl.min(0) - lo[0]).max())
print("node pad ~", pad)
def area(l, h):
    d = np.maximum(h - l, 0); return 2 * (d[...,0]*d[...,1] + d[...,1]*d[...,2] + d[...,2]*d[...,0])

def build(kind):
    """returns lo,hi,skip,leaf in pre-order; primitives = Morton-sorted tris (index into ts); kind: 'lbvh_sah' split by SAH in Morton order;
    'sah' full binned SAH with reordering (then the tri order changes: returns perm too)"""
    nodes = []  # (lo,hi,skip,leaf)
    perm = np.arange(N)
    out_lo, out_hi, out_skip, out_leaf = [], [], [], []
    order_out = []
    def rec(idx):
        # idx: array of primitive ids (in current order)
        me = len(out_lo)
        l, h = gl[idx].min(0), gh[idx].max(0)
        out_lo.append(l - pad); out_hi.append(h + pad); out_skip.append(0); out_leaf.append(-1)
        n = len(idx)
        if n <= 2:
            f = len(order_out); order_out.extend(idx.tolist())
            out_leaf[me] = f * 8 + (n - 1)
        else:
            if kind == 'lbvh_sah':
                pl = np.minimum.accumulate(gl[idx], 0); ph = np.maximum.accumulate(gh[idx], 0)
                sl = np.minimum.accumulate(gl[idx][::-1], 0)[::-1]; sh = np.maximum.accumulate(gh[idx][::-1], 0)[::-1]
                k = np.arange(1, n)
                cost = area(pl[:-1], ph[:-1]) * k + area(sl[1:], sh[1:]) * (n - k)
                # keep quads together: only even splits when possible
                if n > 3: cost[0::2] += 1e30
                s = int(np.argmin(cost)) + 1
                L, R = idx[:s], idx[s:]
            else:
                c = 0.5 * (gl[idx] + gh[idx]); best = None
                for a in range(3):
                    o = np.argsort(c[:, a], kind='stable'); ii = idx[o]
                    pl = np.minimum.accumulate(gl[ii], 0); ph = np.maximum.accumulate(gh[ii], 0)
                    sl = np.minimum.accumulate(gl[ii][::-1], 0)[::-1]; sh = np.maximum.accumulate(gh[ii][::-1], 0)[::-1]
                    k = np.arange(1, n)
                    cost = area(pl[:-1], ph[:-1]) * k + area(sl[1:], sh[1:]) * (n - k)
                    s = int(np.argmin(cost))
                    if best is None or cost[s] < best[0]: best = (cost[s], ii, s + 1)
                _, ii, s = best
                L, R = ii[:s], ii[s:]
            rec(L); rec(R)
        out_skip[me] = len(out_lo)
    rec(np.arange(N))
    return np.array(out_lo), np.array(out_hi), np.array(out_skip), np.array(out_leaf), np.array(order_out)

kind = os.environ.get("KIND", "lbvh_sah")
lo, hi, skip, leaf, order_out = build(kind)
NN = len(skip)
sA, sE1, sE2, sid = sA[order_out], sE1[order_out], sE2[order_out], sid[order_out]
print(kind, "nodes", NN, "SAH cost", (area(lo, hi)[leaf < 0].sum() + 2 * area(lo, hi)[leaf >= 0].sum()) / area(lo[0], hi[0]))
leaf_of = np.full(N, -1)
for n in range(NN):
    if leaf[n] >= 0:
        f, cnt = leaf[n] >> 3, (leaf[n] & 3) + 1
        for k in range(cnt): leaf_of[sid[f + k]] = n
exec("rs = np.random.RandomState(1)" + open("tools/sim_walk.py").read().split("rs = np.random.RandomState(1)")[1])
