"""Offline study of the BVH walk of k_ff_tiles (CPU, numpy): replays the wave-level walk (a node is entered when ANY of a pair's 50 rays
hits it) on the device's own BVH (tools/dump_bvh.py -> gpurun_out/bvh_16384.npz) for sampled tile pairs and counts node visits per
pair for: the walk from the root; the same without the always-hit ancestors of the two end patches; the tree pruned to the tile
pair's shaft; both.  Reproduces the device's DR_TILE_STATS count (93 visits per pair at 16 384 patches).  MORTON=1: tiles = 64
consecutive patches in the BVH's Morton order instead of the scene's order."""
import numpy as np, sys, os
sys.path.insert(0,'.')
from daisyriot_amd import scenes
d = np.load(os.environ.get("BVH_NPZ", "profiles/r02/bvh_16384.npz"))
lo, hi, skip, leaf = d["lo"].astype(np.float64), d["hi"].astype(np.float64), d["skip"].astype(np.int64), d["tri"].astype(np.int64)
tr = d["trirec"]; ts = d["trisorted"]
NN = len(skip); N = tr.shape[0]
A, E1, E2 = tr[:,0:3].astype(np.float64), tr[:,3:6].astype(np.float64), tr[:,6:9].astype(np.float64)
glo, ghi = tr[:,10:13].astype(np.float64), tr[:,13:16].astype(np.float64)
sA, sE1, sE2 = ts[:,0:3].astype(np.float64), ts[:,3:6].astype(np.float64), ts[:,6:9].astype(np.float64)
sid = ts[:,9].view(np.int32).astype(np.int64)
uv = scenes.visibility_samples(50).astype(np.float64)
nrm = np.cross(E1, E2); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
cen = A + (E1 + E2) / 3
# leaf node of each patch (node index) and ancestors
leaf_of = np.full(N, -1)
for n in range(NN):
    if leaf[n] >= 0:
        f, cnt = leaf[n] >> 3, (leaf[n] & 3) + 1
        for k in range(cnt): leaf_of[sid[f + k]] = n
depth_stats = []

def shaft_mask(bIlo, bIhi, bJlo, bJhi, pad=1e-4):
    """nodes whose box meets the hull of the two boxes (AABB + connecting planes)"""
    bIlo, bIhi, bJlo, bJhi = bIlo - pad, bIhi + pad, bJlo - pad, bJhi + pad
    ulo, uhi = np.minimum(bIlo, bJlo), np.maximum(bIhi, bJhi)
    ok = np.all((hi >= ulo) & (lo <= uhi), axis=1)
    c, h = 0.5 * (lo + hi), 0.5 * (hi - lo)
    for a in range(3):
        p, q = (a + 1) % 3, (a + 2) % 3
        for sp in (-1, 1):
            for sq in (-1, 1):
                Ap = bIhi[p] if sp > 0 else bIlo[p]; Aq = bIhi[q] if sq > 0 else bIlo[q]
                Bp = bJhi[p] if sp > 0 else bJlo[p]; Bq = bJhi[q] if sq > 0 else bJlo[q]
                dp, dq = Bp - Ap, Bq - Aq
                if dp * sp * dq * sq >= 0: continue
                n_p, n_q = sp * abs(dq), sq * abs(dp)
                d0 = max(n_p * Ap + n_q * Aq, n_p * Bp + n_q * Bq)
                s = n_p * c[:, p] + n_q * c[:, q] - (abs(n_p) * h[:, p] + abs(n_q) * h[:, q])
                ok &= ~(s > d0 + 1e-9)
    return ok

def walk(org, dn, tmax, hiid, alive0, mode, inshaft=None, anc=None):
    """org,dn: [P,K,3]; tmax [P,K]; returns visits[P], leaves[P].  mode: 'root' | 'pruned' (skip nodes not in shaft free of charge)
    anc: [P,NN] bool optional -> known ancestors are not counted (and always entered)"""
    P = org.shape[0]
    inv = 1.0 / np.where(dn == 0, 1e-300, dn)
    off = np.zeros(P, np.int64); visits = np.zeros(P, np.int64); leaves = np.zeros(P, np.int64)
    alive = alive0.copy(); act = alive.any(axis=1)
    while True:
        act &= off < NN
        if not act.any(): break
        idx = np.nonzero(act)[0]
        o = off[idx]
        if inshaft is not None:
            out = ~inshaft[o]
            if out.any():
                off[idx[out]] = skip[o[out]]
                continue
        t0 = (lo[o][:, None, :] - org[idx]) * inv[idx]; t1 = (hi[o][:, None, :] - org[idx]) * inv[idx]
        tn = np.minimum(t0, t1).max(axis=2); tf = np.maximum(t0, t1).min(axis=2)
        h = (np.maximum(tn, 0) <= np.minimum(tf, tmax[idx])) & alive[idx]
        hit = h.any(axis=1)
        if anc is not None:
            isanc = anc[idx, o]
            visits[idx] += ~isanc
            hit = hit | isanc
        else:
            visits[idx] += 1
        lf = leaf[o]
        # misses
        off[idx[~hit]] = skip[o[~hit]]
        off[idx[hit]] = o[hit] + 1
        hl = hit & (lf >= 0)
        if hl.any():
            ii = idx[hl]; leaves[ii] += 1
            f = lf[hl] >> 3; cnt = (lf[hl] & 3) + 1
            for k in range(2):
                m = cnt > k
                if not m.any(): continue
                i2 = ii[m]; t = f[m] + k
                a, e1, e2 = sA[t][:, None, :], sE1[t][:, None, :], sE2[t][:, None, :]
                pv = np.cross(dn[i2], e2); det = (e1 * pv).sum(-1)
                with np.errstate(all='ignore'):
                    iv = 1.0 / det; tv = org[i2] - a
                    u = (tv * pv).sum(-1) * iv; qv = np.cross(tv, e1); v = (dn[i2] * qv).sum(-1) * iv; tt = (e2 * qv).sum(-1) * iv
                    blk = (u >= 0) & (v >= 0) & (u + v <= 1) & (tt > 1e-9) & (tt < tmax[i2] - 1e-7) & (sid[t][:, None] != hiid[i2][:, None])
                alive[i2] &= ~blk
            act[ii] = alive[ii].any(axis=1)
    return visits, leaves

rs = np.random.RandomState(1)
nT = N // 64
res = {}
def tile_box(t):
    g = np.arange(t * 64, t * 64 + 64)
    return glo[g].min(0), ghi[g].max(0)
order = None
if os.environ.get("MORTON"):
    # tiles = 64 consecutive patches in the BVH's Morton order
    order = sid[:N].copy()
tot = {k: [0, 0] for k in ("root", "root_noanc", "pruned", "pruned_noanc")}
npairs_tot = 0
shaft_nodes = []
for it in range(int(os.environ.get("NTILE", "40"))):
    tI, tJ = sorted(rs.randint(0, nT, 2))
    gI = np.arange(tI * 64, tI * 64 + 64); gJ = np.arange(tJ * 64, tJ * 64 + 64)
    if order is not None: gI, gJ = order[gI], order[gJ]
    ii = rs.randint(0, 64, 256); jj = rs.randint(0, 64, 256)
    a, b = gI[ii], gJ[jj]
    keep = a != b
    a, b = a[keep], b[keep]
    l, h_ = np.minimum(a, b), np.maximum(a, b)
    dv = cen[h_] - cen[l]; dl = np.linalg.norm(dv, axis=1, keepdims=True); dvn = dv / dl
    facing = ((nrm[l] * dvn).sum(1) > 1e-6) & ((nrm[h_] * -dvn).sum(1) > 1e-6)
    l, h_ = l[facing], h_[facing]
    if len(l) == 0: continue
    P = len(l)
    src = A[l][:, None, :] + E1[l][:, None, :] * uv[None, :, 0:1] + E2[l][:, None, :] * uv[None, :, 1:2]
    dst = A[h_][:, None, :] + E1[h_][:, None, :] * uv[None, :, 0:1] + E2[h_][:, None, :] * uv[None, :, 1:2]
    dvv = dst - src; tm = np.linalg.norm(dvv, axis=2); dn = dvv / tm[..., None]
    org = src + dn * 1e-6; tm = tm - 1e-6
    alive0 = np.ones((P, 50), bool)
    bIlo, bIhi = glo[gI].min(0), ghi[gI].max(0); bJlo, bJhi = glo[gJ].min(0), ghi[gJ].max(0)
    ins = shaft_mask(bIlo, bIhi, bJlo, bJhi)
    shaft_nodes.append(ins.sum())
    # ancestors of leaf(l) and leaf(h)
    anc = np.zeros((P, NN), bool)
    ar = np.arange(NN)
    for k in range(P):
        for lf in (leaf_of[l[k]], leaf_of[h_[k]]):
            anc[k] |= (ar <= lf) & (skip > lf) & (leaf < 0)
    for name, kw in (("root", {}), ("root_noanc", dict(anc=anc)), ("pruned", dict(inshaft=ins)), ("pruned_noanc", dict(inshaft=ins, anc=anc))):
        v, lv = walk(org, dn, tm, h_, alive0, name, **kw)
        tot[name][0] += v.sum(); tot[name][1] += lv.sum()
    npairs_tot += P
print("pairs", npairs_tot, "mean nodes in shaft", np.mean(shaft_nodes), "of", NN)
for k, (v, lv) in tot.items(): print("%-14s visits/pair %.1f leaves/pair %.1f" % (k, v / npairs_tot, lv / npairs_tot))
