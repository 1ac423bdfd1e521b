"""Offline estimate (CPU, numpy) for packing the rays of queue-neighbour pairs 64 to a walk: node visits and leaf visits per
PAIR when a wave carries 50 rays of one pair (today) against 64 consecutive rays of the flattened (pair, ray) sequence.
Uses the replay machinery of tools/sim_walk.py on the device's own BVH (profiles/r02/bvh_16384.npz)."""
import numpy as np, sys, os
src = open("tools/sim_walk.py").read().split("rs = np.random.RandomState(1)")[0]
src = src.replace("(sid[t][:, None] != hiid[i2][:, None])", "(sid[t][:, None] != hiid[i2])")
exec(src)
rs = np.random.RandomState(2)
nT = N // 64
K = 50
tot = dict(single=[0, 0, 0], packed=[0, 0, 0])
for it in range(int(os.environ.get("NTILE", "30"))):
    tI, tJ = sorted(rs.randint(0, nT, 2))
    gI = np.arange(tI * 64, tI * 64 + 64); gJ = np.arange(tJ * 64, tJ * 64 + 64)
    i = gI[rs.randint(0, 64)]
    # the queue of row i: all facing j of tile J in order
    js = gJ[gJ != i]
    l = np.minimum(i, js); h_ = np.maximum(i, js)
    dv = cen[h_] - cen[l]; dl = np.linalg.norm(dv, axis=1, keepdims=True); dvn = dv / dl
    facing = ((nrm[l] * dvn).sum(1) > 1e-6) & ((nrm[h_] * -dvn).sum(1) > 1e-6)
    l, h_ = l[facing], h_[facing]
    P = len(l)
    if P < 8: continue
    srcp = A[l][:, None, :] + E1[l][:, None, :] * uv[None, :, 0:1] + E2[l][:, None, :] * uv[None, :, 1:2]
    dst = A[h_][:, None, :] + E1[h_][:, None, :] * uv[None, :, 0:1] + E2[h_][:, None, :] * uv[None, :, 1:2]
    dvv = dst - srcp; tm = np.linalg.norm(dvv, axis=2); dn = dvv / tm[..., None]
    org = srcp + dn * 1e-6; tm = tm - 1e-6
    hid = np.repeat(h_[:, None], K, 1)
    v, lv = walk(org, dn, tm, hid, np.ones((P, K), bool), "root")
    tot["single"][0] += v.sum(); tot["single"][1] += lv.sum(); tot["single"][2] += P
    # packed: flatten (pair, ray), cut into groups of 64 (last group padded with dead lanes)
    F = P * K; G = (F + 63) // 64
    pad = G * 64 - F
    def fl(x, fill):
        x = x.reshape((F,) + x.shape[2:])
        if pad: x = np.concatenate([x, np.broadcast_to(fill, (pad,) + x.shape[1:])])
        return x.reshape((G, 64) + x.shape[1:])
    v2, lv2 = walk(fl(org, org[0, 0]), fl(dn, dn[0, 0]), fl(tm, tm[0, 0]), fl(hid, hid[0, 0]),
                   np.concatenate([np.ones(F, bool), np.zeros(pad, bool)]).reshape(G, 64), "root")
    tot["packed"][0] += v2.sum(); tot["packed"][1] += lv2.sum(); tot["packed"][2] += P
for k, (v, lv, p) in tot.items():
    print("%-8s pairs %d  node visits/pair %.1f  leaf visits/pair %.2f" % (k, p, v / p, lv / p))
