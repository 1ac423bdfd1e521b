"""Synthetic scenes for tests and benches: the subdivided Cornell box of
SURVEY.md 8(d), a closed box and two facing squares.

Every scene is returned as a `Scene` holding exactly the arrays the reference's
MeshS exposes (vs/MeshS.h:14-20: vertices, normals, triangleIndices{vertex,
normal}, materialIndexPerTriangle) plus solver inputs E / M / mat_of_patch in
the layout of include/daisyriot_hip.h (E: N x S patch-major, M: n_mat x S x S).
No randomness except the fixed visibility sample set (`visibility_samples`).
"""
from dataclasses import dataclass, field

import numpy as np


@dataclass
class Scene:
    vertices: np.ndarray          # (V,3) f32
    normals: np.ndarray           # (Nn,3) f32
    tri_v: np.ndarray             # (N,3) i32
    tri_n: np.ndarray             # (N,3) i32
    mat_of_patch: np.ndarray      # (N,) i32
    material_names: list = field(default_factory=list)
    # per material: diffuse spectrum rho[S], emission spectrum e[S], optional full M[S,S]
    rho: np.ndarray = None        # (n_mat,S)
    emit: np.ndarray = None       # (n_mat,S)
    M: np.ndarray = None          # (n_mat,S,S)

    @property
    def N(self):
        return int(self.tri_v.shape[0])

    @property
    def S(self):
        return int(self.M.shape[1])

    def emission(self, emission_value=1.0):
        """E[i,s] = e_s(mat_i)*emission_value where > 0 (vs/Lightning.h:263-273)."""
        e = self.emit[self.mat_of_patch] * np.float32(emission_value)
        return np.where(self.emit[self.mat_of_patch] > 0, e, 0).astype(np.float32)


def vertex_adjacency(tri_v, V):
    """MeshS::trianglesPerVertex (vs/MeshS.cpp:113-115) as CSR (offsets V+1, triangle ids): per vertex the
    triangles that use it, in load order -- the lists Drawer::interpolate averages over."""
    tri_v = np.asarray(tri_v)
    flat_v = tri_v.reshape(-1)
    flat_t = np.repeat(np.arange(tri_v.shape[0], dtype=np.int32), 3)
    order = np.argsort(flat_v, kind="stable")               # stable: load order inside each vertex's list
    off = np.zeros(V + 1, np.int32)
    off[1:] = np.cumsum(np.bincount(flat_v, minlength=V))
    return off, flat_t[order].astype(np.int32)


def visibility_samples(K=50, seed=20191):
    """K (u,v) pairs by the reference's formula (vs/OptixPrimeFunctionality.cpp:57-62:
    v *= 1-u) from mt19937(seed); the same K samples serve every pair."""
    rs = np.random.RandomState(seed)
    u = rs.random_sample(K).astype(np.float32)
    v = rs.random_sample(K).astype(np.float32)
    v = (v * (np.float32(1) - u)).astype(np.float32)
    return np.stack([u, v], axis=1).astype(np.float32)


# ----------------------------------------------------------------------------------
class _Builder:
    def __init__(self):
        self.v, self.n, self.tv, self.tn, self.mat = [], [], [], [], []
        self.nv = 0

    def quad(self, p0, eu, ev, nu, nv, mat):
        """nu x nv cells, two right triangles each; cross(eu,ev) is the facing normal."""
        p0, eu, ev = (np.asarray(a, np.float64) for a in (p0, eu, ev))
        nrm = np.cross(eu, ev)
        nrm = nrm / np.linalg.norm(nrm)
        ni = len(self.n)
        self.n.append(nrm)
        iu = np.arange(nu + 1) / nu
        iv = np.arange(nv + 1) / nv
        P = p0[None, None, :] + iu[:, None, None] * eu[None, None, :] + iv[None, :, None] * ev[None, None, :]
        base = self.nv
        self.v.append(P.reshape(-1, 3))
        self.nv += (nu + 1) * (nv + 1)
        a, b = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
        i00 = base + a * (nv + 1) + b
        i10 = i00 + (nv + 1)
        i01 = i00 + 1
        i11 = i10 + 1
        t1 = np.stack([i00, i10, i11], -1).reshape(-1, 3)
        t2 = np.stack([i00, i11, i01], -1).reshape(-1, 3)
        tris = np.empty((2 * nu * nv, 3), np.int64)
        tris[0::2] = t1
        tris[1::2] = t2
        self.tv.append(tris)
        self.tn.append(np.full((tris.shape[0], 3), ni, np.int64))
        self.mat.append(np.full(tris.shape[0], mat, np.int64))

    def arrays(self):
        return (np.concatenate(self.v), np.stack(self.n), np.concatenate(self.tv),
                np.concatenate(self.tn), np.concatenate(self.mat))


def _bisect_to(v, tv, tn, mat, n_target):
    """Split triangles at the midpoint of their longest edge until there are exactly
    n_target (each split adds one). Splits are spread evenly over the mesh."""
    extra = n_target - tv.shape[0]
    if extra <= 0:
        return v, tv, tn, mat
    pick = (np.arange(extra) * (tv.shape[0] / extra)).astype(np.int64)
    v = list(v)
    tv, tn, mat = tv.copy(), [tn], [mat]
    new_tv, new_tn, new_mat = [], [], []
    base = len(v)
    for k, t in enumerate(pick):
        idx = tv[t]
        p = np.array([v[idx[0]], v[idx[1]], v[idx[2]]])
        el = [np.linalg.norm(p[(e + 1) % 3] - p[e]) for e in range(3)]
        e = int(np.argmax(el))
        i0, i1, i2 = idx[e], idx[(e + 1) % 3], idx[(e + 2) % 3]
        v.append(0.5 * (p[e] + p[(e + 1) % 3]))
        m = base + k
        tv[t] = [i0, m, i2]
        new_tv.append([m, i1, i2])
        new_tn.append(tn[0][t])
        new_mat.append(mat[0][t])
    tvo = np.concatenate([tv, np.array(new_tv, np.int64).reshape(-1, 3)])
    tno = np.concatenate([tn[0], np.array(new_tn, np.int64).reshape(-1, 3)])
    mo = np.concatenate([mat[0], np.array(new_mat, np.int64)])
    return np.array(v), tvo, tno, mo


def _rot_y(deg):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])


def _box_faces(center, size, deg):
    """Five outward faces (no bottom) of a box standing on y = center.y - size.y/2."""
    R = _rot_y(deg)
    c = np.asarray(center, np.float64)
    hx, hy, hz = (s / 2 for s in size)
    ex, ey, ez = R @ [1, 0, 0], np.array([0, 1.0, 0]), R @ [0, 0, 1]
    f = []
    # (p0, eu, ev) with cross(eu,ev) pointing outward
    f.append((c + hy * ey - hx * ex - hz * ez, 2 * hz * ez, 2 * hx * ex))          # top, +y
    f.append((c + hx * ex - hy * ey - hz * ez, 2 * hy * ey, 2 * hz * ez))          # +x
    f.append((c - hx * ex - hy * ey - hz * ez, 2 * hz * ez, 2 * hy * ey))          # -x
    f.append((c + hz * ez - hy * ey - hx * ex, 2 * hx * ex, 2 * hy * ey))          # +z
    f.append((c - hz * ez - hy * ey - hx * ex, 2 * hy * ey, 2 * hx * ex))          # -z
    return f


WAVELENGTHS_8 = np.arange(250.0, 601.0, 50.0, dtype=np.float32)     # S = 8
WAVELENGTHS_9 = np.arange(200.0, 601.0, 50.0, dtype=np.float32)     # main.cpp:94


def _smooth_spectrum(rgb, wavelengths):
    """A smooth, bounded reflectance spectrum for an RGB triple (three raised-cosine
    lobes).  Synthetic stand-in for the Jakob-Hanika table, which the reference tree
    does not ship (.MISSING_LARGE_BLOBS)."""
    w = np.asarray(wavelengths, np.float64)
    lobes = [(600.0, 90.0), (540.0, 70.0), (450.0, 80.0)]
    out = np.zeros_like(w)
    for c, (mu, sig) in zip(rgb, lobes):
        out += c * np.exp(-0.5 * ((w - mu) / sig) ** 2)
    return np.clip(out, 0.0, 0.98).astype(np.float32)


def _materials(S, fluorescent):
    if S == 1:
        wl = None
    elif S == 3:
        wl = None
    elif S == 8:
        wl = WAVELENGTHS_8
    elif S == 9:
        wl = WAVELENGTHS_9
    else:
        wl = np.linspace(250.0, 600.0, S).astype(np.float32)
    names = ["white", "red", "green", "lamp"]
    rgb = [(0.75, 0.75, 0.75), (0.63, 0.065, 0.05), (0.14, 0.45, 0.091), (0.78, 0.78, 0.78)]
    emit_rgb = [(0, 0, 0), (0, 0, 0), (0, 0, 0), (1, 1, 1)]
    if fluorescent:
        names += ["fluorescent", "uvlamp"]
        rgb += [(0.6, 0.6, 0.6), (0, 0, 0)]
        emit_rgb += [(0, 0, 0), (0, 0, 0)]
    n_mat = len(names)
    rho = np.zeros((n_mat, S), np.float32)
    emit = np.zeros((n_mat, S), np.float32)
    for m in range(n_mat):
        if S == 1:
            rho[m, 0] = np.float32(np.mean(rgb[m]))
            emit[m, 0] = np.float32(np.mean(emit_rgb[m]))
        elif S == 3:
            rho[m] = rgb[m]
            emit[m] = emit_rgb[m]
        else:
            flat = len(set(rgb[m])) == 1
            rho[m] = np.float32(rgb[m][0]) if flat else _smooth_spectrum(rgb[m], wl)
            emit[m] = np.float32(emit_rgb[m][0])
    M = np.zeros((n_mat, S, S), np.float32)
    for m in range(n_mat):
        M[m] = np.diag(rho[m])                      # vs/Material.cpp:17-20
    if fluorescent and wl is not None:
        f = names.index("fluorescent")
        blue = _smooth_spectrum((0.05, 0.2, 0.9), wl)
        M[f] = np.eye(S, dtype=np.float32)          # vs/Material.cpp:90-100: M = I, ...
        for s in range(S):
            if 300.0 < wl[s] < 400.0:
                M[f][:, s] = blue                   # ... column of every UV bin = blacklight spectrum
        u = names.index("uvlamp")
        emit[u] = np.exp(-((wl.astype(np.float64) - 350.0) ** 2) / 200.0).astype(np.float32)  # Material.cpp:69-77
        rho[u] = 0
        M[u] = 0                                    # deviation: reference reads vec3 out of bounds
    return names, rho, emit, M


def cornell_box(n_patches, S=8, fluorescent=False):
    """Subdivided Cornell box with exactly n_patches triangles."""
    assert n_patches >= 64
    quads = []  # (p0, eu, ev, material)
    quads.append(((-1, -1, -1), (0, 0, 2), (2, 0, 0), 0))     # floor  +y
    quads.append(((-1, 1, -1), (2, 0, 0), (0, 0, 2), 0))      # ceiling -y
    quads.append(((-1, -1, -1), (2, 0, 0), (0, 2, 0), 0))     # back   +z
    quads.append(((-1, -1, 1), (0, 2, 0), (2, 0, 0), 0))      # front  -z
    quads.append(((-1, -1, -1), (0, 2, 0), (0, 0, 2), 1))     # left   +x  red
    quads.append(((1, -1, -1), (0, 0, 2), (0, 2, 0), 2))      # right  -x  green
    lamp_mat = 3
    quads.append(((-0.25, 0.99, -0.25), (0.5, 0, 0), (0, 0, 0.5), lamp_mat))   # lamp -y
    short_mat = 4 if fluorescent else 0
    for f in _box_faces((0.33, -0.7, 0.35), (0.6, 0.6, 0.6), 18.0):
        quads.append((*f, short_mat))
    for f in _box_faces((-0.35, -0.4, -0.3), (0.6, 1.2, 0.6), -18.0):
        quads.append((*f, 0))
    if fluorescent:
        quads.append(((0.45, 0.99, 0.45), (0.3, 0, 0), (0, 0, 0.3), 5))        # UV lamp -y

    lens = [(np.linalg.norm(q[1]), np.linalg.norm(q[2])) for q in quads]

    def grids(f):
        return [(max(1, int(round(f * lu))), max(1, int(round(f * lv)))) for lu, lv in lens]

    def total(f):
        return 2 * sum(a * b for a, b in grids(f))

    lo, hi = 0.01, 4096.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if total(mid) <= n_patches:
            lo = mid
        else:
            hi = mid
    b = _Builder()
    for q, (nu, nv) in zip(quads, grids(lo)):
        b.quad(q[0], q[1], q[2], nu, nv, q[3])
    v, n, tv, tn, mat = b.arrays()
    v, tv, tn, mat = _bisect_to(v, tv, tn, mat, n_patches)
    names, rho, emit, M = _materials(S, fluorescent)
    return Scene(v.astype(np.float32), n.astype(np.float32), tv.astype(np.int32), tn.astype(np.int32),
                 mat.astype(np.int32), names, rho, emit, M)


def closed_box(cells=1, S=3):
    """Axis-aligned closed unit box, inward normals, 12*cells^2 triangles; face 0 emits."""
    b = _Builder()
    faces = [((0, 0, 0), (0, 0, 1), (1, 0, 0)), ((0, 1, 0), (1, 0, 0), (0, 0, 1)),
             ((0, 0, 0), (1, 0, 0), (0, 1, 0)), ((0, 0, 1), (0, 1, 0), (1, 0, 0)),
             ((0, 0, 0), (0, 1, 0), (0, 0, 1)), ((1, 0, 0), (0, 0, 1), (0, 1, 0))]
    for k, f in enumerate(faces):
        b.quad(f[0], f[1], f[2], cells, cells, 3 if k == 1 else (1 if k == 4 else (2 if k == 5 else 0)))
    v, n, tv, tn, mat = b.arrays()
    names, rho, emit, M = _materials(S, False)
    return Scene(v.astype(np.float32), n.astype(np.float32), tv.astype(np.int32), tn.astype(np.int32),
                 mat.astype(np.int32), names, rho, emit, M)


def facing_squares(cells=2, gap=1.0, S=1, occluder=False):
    """Two parallel unit squares `gap` apart facing each other; optionally a smaller
    square between them (blocks part of the exchange)."""
    b = _Builder()
    b.quad((0, 0, 0), (0, 0, 1), (1, 0, 0), cells, cells, 3)        # y=0, facing +y, emits
    b.quad((0, gap, 0), (1, 0, 0), (0, 0, 1), cells, cells, 0)      # y=gap, facing -y
    if occluder:
        b.quad((0.25, gap / 2, 0.25), (0.5, 0, 0), (0, 0, 0.5), max(1, cells // 2), max(1, cells // 2), 0)
    v, n, tv, tn, mat = b.arrays()
    names, rho, emit, M = _materials(S, False)
    return Scene(v.astype(np.float32), n.astype(np.float32), tv.astype(np.int32), tn.astype(np.int32),
                 mat.astype(np.int32), names, rho, emit, M)


def write_obj(scene, obj_path, mtl_name="scene.mtl"):
    """Emit the scene as v / vn / `f a//n` with usemtl groups (the form the reference's
    example scenes use), so it can go through the loader surface."""
    with open(obj_path, "w") as f:
        f.write("mtllib %s\n" % mtl_name)
        for p in scene.vertices:
            f.write("v %.9g %.9g %.9g\n" % tuple(p))
        for p in scene.normals:
            f.write("vn %.9g %.9g %.9g\n" % tuple(p))
        cur = -1
        for t in range(scene.N):
            m = int(scene.mat_of_patch[t])
            if m != cur:
                f.write("usemtl %s\n" % scene.material_names[m])
                cur = m
            a, b, c = scene.tri_v[t] + 1
            na, nb, nc = scene.tri_n[t] + 1
            f.write("f %d//%d %d//%d %d//%d\n" % (a, na, b, nb, c, nc))


def write_mtl(scene, mtl_path, rgb=None, emit_rgb=None, blacklight=None):
    """MTL with the three keys the reference reads (vs/MeshS.cpp:37-39): Kd diffuse, Ke emission,
    Ks = the "blacklight" colour that marks a fluorescent material.  rgb / emit_rgb default to the
    scene's S=3 reflectance / emission."""
    rgb = scene.rho if rgb is None else rgb
    emit_rgb = scene.emit if emit_rgb is None else emit_rgb
    with open(mtl_path, "w") as f:
        for m, name in enumerate(scene.material_names):
            f.write("newmtl %s\n" % name)
            f.write("Kd %.9g %.9g %.9g\n" % tuple(rgb[m][:3]))
            f.write("Ke %.9g %.9g %.9g\n" % tuple(emit_rgb[m][:3]))
            ks = (0, 0, 0) if blacklight is None else blacklight[m]
            f.write("Ks %.9g %.9g %.9g\n\n" % tuple(ks))
