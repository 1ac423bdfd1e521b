"""The BVH node test of k_ff_tiles (geom_kernels.hip: DR_NODE_TEST_SX / DR_NODE_TEST_X and the per-ray set-up around them) is a
different piece of arithmetic from the triangles' gate test (box_hit_mask, part of the hit definition): fma form, the ray
parameter in units of the ray's own length, the VOP3 clamp on one near and on the far value, a strict compare, boxes grown by
node_pad.  It only has to be CONSERVATIVE: never reject a node that holds a triangle whose gate accepts.  This replays both
in float32 (numpy; fma through float64, whose product of two float32 is exact) on rays aimed to graze boxes, with the pads the
library uses (dr_api.cpp: box_pad, node_pad; k_emit: centre / half-extent and the outward-rounded corners), and checks that
implication -- over scene scales, offsets from the origin, axis-parallel rays and rays that start inside the box."""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def gate_accepts(lo, hi, org, inv, tmax):
    """box_hit_mask: (plane - org) * inv, min / max, tf * 1.00001, one compare"""
    with np.errstate(all="ignore"):
        t0 = (lo - org) * inv
        t1 = (hi - org) * inv
        tn = np.fmax(np.fmax(np.fmax(np.fmin(t0[:, 0], t1[:, 0]), np.fmin(t0[:, 1], t1[:, 1])), np.fmin(t0[:, 2], t1[:, 2])), f32(0))
        tf = np.fmin(np.fmin(np.fmax(t0[:, 0], t1[:, 0]), np.fmax(t0[:, 1], t1[:, 1])), np.fmax(t0[:, 2], t1[:, 2]))
        return tn <= np.fmin(tf * f32(1.00001), tmax)


def node_boxes(glo, ghi, node_pad):
    """k_emit: centre / half-extent holding [lo - pad, hi + pad]; then lower / upper corners rounded outwards"""
    c = f32(0.5) * glo + f32(0.5) * ghi
    h = np.maximum(ghi - c, c - glo) * f32(1.000001) + node_pad
    m = f32(4e-7) * (np.abs(c) + h)
    return c, h, (c - h) - m, (c + h) + m


def clamp01(x):
    with np.errstate(all="ignore"):
        return np.where(np.isnan(x), f32(0), np.minimum(np.maximum(x, f32(0)), f32(1))).astype(f32)


def node_accepts(c, h, nlo, nhi, org, dn, inv, tmax, ts_max):
    with np.errstate(all="ignore"):
        s = np.minimum((f32(1) / tmax) * f32(0.99999), ts_max).astype(f32)
        iv = (np.clip(inv, f32(-1e18), f32(1e18)) * s[:, None]).astype(f32)
        k = -(org * iv)
        neg = dn < 0
        near = np.where(neg, nhi, nlo)
        far = np.where(neg, nlo, nhi)
        tn = fma(near, iv, k)
        tf = fma(far, iv, k)
        N = np.maximum(np.maximum(clamp01(tn[:, 0]), tn[:, 1]), tn[:, 2])
        F = clamp01(np.minimum(np.minimum(tf[:, 0], tf[:, 1]), tf[:, 2]))
        octant_form = N < F
        tc = fma(c, iv, k)
        a = np.abs(iv)
        tn2 = fma(h, -a, tc)
        tf2 = fma(h, a, tc)
        N2 = np.maximum(np.maximum(clamp01(tn2[:, 0]), tn2[:, 1]), tn2[:, 2])
        F2 = clamp01(np.minimum(np.minimum(tf2[:, 0], tf2[:, 1]), tf2[:, 2]))
        return octant_form, N2 < F2


def _scene(rs, n, scale, offset):
    """n triangles' gate boxes inside a scene of the given scale / offset, each inside a node box over a few neighbours"""
    slo = (np.array([-1, -1, -1]) * scale + offset).astype(f32)
    shi = (np.array([1, 1, 1]) * scale + offset).astype(f32)
    ext = f32((shi - slo).max())
    box_pad = f32(1e-4) * ext + f32(1e-30)
    diag = f32(np.sqrt((((shi - slo) + 2 * box_pad) ** 2).sum()))
    maxabs = f32(max(np.abs(slo).max(), np.abs(shi).max()) + box_pad)
    node_pad = f32(3e-5) * diag + f32(4e-6) * maxabs
    ts_max = f32(min(1e19 / float(maxabs + diag), 1e18))
    cen = (rs.uniform(-0.9, 0.9, (n, 3)) * scale + offset)
    size = (10.0 ** rs.uniform(-3, -0.5, (n, 1))) * scale * rs.uniform(0.0, 1.0, (n, 3))      # flat boxes included
    tlo, thi = (cen - size).astype(f32) - box_pad, (cen + size).astype(f32) + box_pad
    # the node: the triangle's gate box united with a neighbour's
    grow = (rs.uniform(0, 1, (n, 3)) * size * rs.choice([0.0, 1.0, 4.0], (n, 1))).astype(f32)
    glo, ghi = tlo - grow, thi + grow * f32(0.5)
    return tlo, thi, glo, ghi, node_pad, ts_max, float(diag)


def _rays(rs, tlo, thi, scale, offset, diag, mode):
    n = tlo.shape[0]
    # a target on (or a hair off) the gate box, so that the gate's verdict hangs on roundings
    t = rs.uniform(0, 1, (n, 3))
    tgt = tlo + (thi - tlo) * t.astype(f32)
    face = rs.randint(0, 3, n)
    side = rs.randint(0, 2, n)
    tgt[np.arange(n), face] = np.where(side == 0, tlo[np.arange(n), face], thi[np.arange(n), face])
    tgt = tgt + ((rs.uniform(-1, 1, (n, 3)) * (10.0 ** rs.uniform(-8, -4, (n, 1))) * scale)).astype(f32)
    if mode == "inside":
        org = (tlo + (thi - tlo) * rs.uniform(0, 1, (n, 3)).astype(f32)).astype(f32)
    else:
        org = (rs.uniform(-1, 1, (n, 3)) * scale + offset).astype(f32)
    d = (tgt - org).astype(np.float64)
    if mode == "axis":
        kill = rs.randint(0, 3, n)
        d[np.arange(n), kill] = 0.0
        d[np.arange(n), (kill + 1) % 3] *= rs.choice([0.0, 1.0], n)
    L = np.sqrt((d * d).sum(1))
    ok = L > 0
    dn = np.zeros_like(d); dn[ok] = d[ok] / L[ok, None]
    dn = dn.astype(f32)
    with np.errstate(all="ignore"):
        inv = np.where(dn == 0, f32(3.0e38), f32(1.0) / dn).astype(f32)
    tmax = (L * rs.uniform(0.3, 3.0, n)).astype(f32)
    tmax = np.minimum(tmax, f32(diag))
    return org[ok], dn[ok], inv[ok], tmax[ok], ok


def test_node_test_never_rejects_what_a_gate_accepts():
    rs = np.random.RandomState(4)
    checked = accepted = 0
    for scale, off in ((1.0, 0.0), (1e-3, 0.0), (1e3, 0.0), (1.0, 300.0), (2.0, -4000.0 * 2.0), (1e-3, 4.0), (50.0, 1e5)):
        for mode in ("graze", "inside", "axis"):
            tlo, thi, glo, ghi, node_pad, ts_max, diag = _scene(rs, 200_000, scale, off)
            org, dn, inv, tmax, ok = _rays(rs, tlo, thi, scale, off, diag, mode)
            tlo, thi, glo, ghi = tlo[ok], thi[ok], glo[ok], ghi[ok]
            live = tmax > 0
            g = gate_accepts(tlo, thi, org, inv, tmax) & live
            c, h, nlo, nhi = node_boxes(glo, ghi, node_pad)
            o_form, g_form = node_accepts(c, h, nlo, nhi, org, dn, inv, tmax, ts_max)
            bad = g & ~(o_form & g_form)
            assert not bad.any(), (scale, off, mode, int(bad.sum()), org[bad][:2], dn[bad][:2], tmax[bad][:2])
            checked += g.size; accepted += int(g.sum())
    assert accepted > 0.2 * checked                               # the rays do test the implication (many accepted gates) ...
    assert accepted < 0.95 * checked                              # ... and many rejected ones right beside them
