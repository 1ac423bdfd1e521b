"""ctypes binding of the CPU oracle (oracle/liboracle.so) and, when present, of
oracle/_ref/libref_check.so (the reference's vendored glm/Eigen arithmetic).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  daisyriot_amd/ never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_check.so")

RULE_INTEGRAND = 0
RULE_RECIPROCITY = 1


class _Mesh(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("V", C.c_int),
                ("normals", C.c_void_p), ("Nn", C.c_int),
                ("tri_v", C.c_void_p), ("tri_n", C.c_void_p), ("N", C.c_int)]


def build(force=False):
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        fp, ip, u8p = C.c_void_p, C.c_void_p, C.c_void_p
        L.orc_surface.restype = C.c_float
        L.orc_surface.argtypes = [fp, fp, fp]
        L.orc_p2p_integrand_literal.restype = C.c_float
        L.orc_p2p_integrand_literal.argtypes = [C.POINTER(_Mesh), C.c_int, C.c_int]
        L.orc_patch_records.argtypes = [C.POINTER(_Mesh), fp, fp, fp, fp]
        L.orc_integrand_rows.argtypes = [C.POINTER(_Mesh), C.c_int, C.c_int, fp]
        L.orc_integrand_rows_cuda_twin.argtypes = [C.POINTER(_Mesh), C.c_int, C.c_int, fp]
        L.orc_uv2xyz.argtypes = [C.POINTER(_Mesh), C.c_int, C.c_float, C.c_float, fp]
        L.orc_closest_hit.restype = C.c_int
        L.orc_closest_hit.argtypes = [C.POINTER(_Mesh), fp, fp, fp]
        L.orc_visibility_count.restype = C.c_int
        L.orc_visibility_count.argtypes = [C.POINTER(_Mesh), C.c_int, C.c_int, fp, C.c_int, C.c_float]
        for f in (L.orc_assemble_rows, L.orc_assemble_rows_bvh):
            f.restype = C.c_int
            f.argtypes = [C.POINTER(_Mesh), fp, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, fp, u8p, C.c_int]
        L.orc_assemble_row_list_bvh.restype = C.c_int
        L.orc_assemble_row_list_bvh.argtypes = [C.POINTER(_Mesh), fp, C.c_int, C.c_float, C.c_int, ip, C.c_int, fp, u8p, C.c_int]
        L.orc_sweep_rows.argtypes = [C.c_int, C.c_int, fp, C.c_long, C.c_int, C.c_int, fp, ip, fp, fp, fp, C.c_int]
        L.orc_residual_sums.argtypes = [C.c_int, C.c_int, fp, fp]
        L.orc_xyz_fit.argtypes = [C.c_double, fp]
        L.orc_patch_colors.argtypes = [C.c_int, C.c_int, fp, C.c_int, fp, fp]
        L.orc_vertex_colors.argtypes = [C.c_int, ip, ip, fp, fp]
        L.orc_converge.restype = C.c_int
        L.orc_converge.argtypes = [C.c_int, C.c_int, fp, fp, ip, fp, fp, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Mesh:
    """Holds contiguous copies of the MeshS arrays and the C struct over them."""

    def __init__(self, vertices, normals, tri_v, tri_n):
        self.vertices = _f32(vertices).reshape(-1, 3)
        self.normals = _f32(normals).reshape(-1, 3)
        self.tri_v = _i32(tri_v).reshape(-1, 3)
        self.tri_n = _i32(tri_n).reshape(-1, 3)
        self.N = self.tri_v.shape[0]
        self.c = _Mesh(_p(self.vertices), self.vertices.shape[0], _p(self.normals),
                       self.normals.shape[0], _p(self.tri_v), _p(self.tri_n), self.N)

    @property
    def ref(self):
        return C.byref(self.c)


def surface(a, b, c):
    a, b, c = _f32(a), _f32(b), _f32(c)
    return float(lib().orc_surface(_p(a), _p(b), _p(c)))


def p2p_integrand_literal(mesh, i, j):
    return np.float32(lib().orc_p2p_integrand_literal(mesh.ref, int(i), int(j)))


def patch_records(mesh):
    N = mesh.N
    cen = np.empty((N, 4, 3), np.float32)
    sa = np.empty((N, 4), np.float32)
    nrm = np.empty((N, 3), np.float32)
    area = np.empty((N,), np.float32)
    lib().orc_patch_records(mesh.ref, _p(cen), _p(sa), _p(nrm), _p(area))
    return cen, sa, nrm, area


def integrand_rows(mesh, row0=0, nrows=None, cuda_twin=False):
    """stored unoccluded integrand rows; cuda_twin: with the mixed float/double arithmetic of the reference's CUDA
    kernel (parallellism.cu:197-207) instead of triangle_math.cpp's all-float form"""
    nrows = mesh.N - row0 if nrows is None else nrows
    out = np.empty((nrows, mesh.N), np.float32)
    (lib().orc_integrand_rows_cuda_twin if cuda_twin else lib().orc_integrand_rows)(mesh.ref, row0, nrows, _p(out))
    return out


def uv2xyz(mesh, tri, u, v):
    out = np.empty(3, np.float32)
    lib().orc_uv2xyz(mesh.ref, int(tri), C.c_float(u), C.c_float(v), _p(out))
    return out


def closest_hit(mesh, org, direction):
    o, d = _f32(org), _f32(direction)
    t = np.empty(1, np.float32)
    tid = lib().orc_closest_hit(mesh.ref, _p(o), _p(d), _p(t))
    return tid, float(t[0])


def visibility_count(mesh, lo, hi, uv, eps=1e-6):
    uv = _f32(uv).reshape(-1, 2)
    return lib().orc_visibility_count(mesh.ref, int(lo), int(hi), _p(uv), uv.shape[0], C.c_float(eps))


def assemble_rows(mesh, uv, eps=1e-6, rule=RULE_INTEGRAND, row0=0, nrows=None,
                  want_vis=True, threads=0, bvh=False):
    uv = _f32(uv).reshape(-1, 2)
    nrows = mesh.N - row0 if nrows is None else nrows
    F = np.empty((nrows, mesh.N), np.float32)
    vis = np.empty((nrows, mesh.N), np.uint8) if want_vis else None
    fn = lib().orc_assemble_rows_bvh if bvh else lib().orc_assemble_rows
    used = fn(mesh.ref, _p(uv), uv.shape[0], C.c_float(eps), rule, row0, nrows, _p(F),
              _p(vis) if want_vis else None, threads)
    return (F, vis, used) if want_vis else (F, None, used)


def assemble_row_list(mesh, uv, rows, eps=1e-6, rule=RULE_INTEGRAND, want_vis=False, threads=0):
    """F (and ray counts) of an arbitrary list of rows through the oracle's BVH, records and BVH built once."""
    uv = _f32(uv).reshape(-1, 2)
    rows = _i32(rows)
    F = np.empty((rows.shape[0], mesh.N), np.float32)
    vis = np.empty((rows.shape[0], mesh.N), np.uint8) if want_vis else None
    used = lib().orc_assemble_row_list_bvh(mesh.ref, _p(uv), uv.shape[0], C.c_float(eps), rule, _p(rows), rows.shape[0], _p(F),
                                           _p(vis) if want_vis else None, threads)
    if used < 0:
        raise ValueError("row outside the mesh")
    return F, vis, used


def sweep_rows(F, M, mat, Rin, B, row0=0, threads=0):
    """One light pass for the rows held in F (nrows x N). Returns Rout; B updated in place."""
    F = _f32(F)
    nrows, N = F.shape
    Rin = _f32(Rin)
    S = Rin.shape[1]
    M = _f32(M)
    mat = _i32(mat)
    assert B.dtype == np.float32 and B.flags.c_contiguous and B.shape == (nrows, S)
    Rout = np.empty((nrows, S), np.float32)
    lib().orc_sweep_rows(N, S, _p(F), F.strides[0] // 4, row0, nrows, _p(M), _p(mat),
                         _p(Rin), _p(Rout), _p(B), threads)
    return Rout


def residual_sums(R):
    R = _f32(R)
    sums = np.empty(R.shape[1], np.float64)
    lib().orc_residual_sums(R.shape[0], R.shape[1], _p(R), _p(sums))
    return sums


def converge(F, M, mat, E, threshold, per_bin, max_iters, threads=0):
    F = _f32(F)
    N = F.shape[0]
    R = _f32(E).copy()
    B = _f32(E).copy()
    S = R.shape[1]
    M, mat = _f32(M), _i32(mat)
    it = lib().orc_converge(N, S, _p(F), _p(M), _p(mat), _p(R), _p(B), C.c_float(threshold),
                            int(per_bin), int(max_iters), threads)
    return it, R, B


def xyz_fit(wavelengths):
    """(S, 3) CIE 1931 fit values the spectral display uses (vs/color.h:14-45)."""
    out = np.empty((len(wavelengths), 3), np.float32)
    for k, w in enumerate(wavelengths):
        o = np.empty(3, np.float32)
        lib().orc_xyz_fit(C.c_double(float(w)), _p(o))
        out[k] = o
    return out


def patch_colors(B, mode, xyz=None):
    B = _f32(B)
    N, S = B.shape
    rgb = np.empty((N, 3), np.float32)
    xyz = _f32(xyz) if xyz is not None else np.zeros((S, 3), np.float32)
    lib().orc_patch_colors(N, S, _p(B), int(mode), _p(xyz), _p(rgb))
    return rgb


def vertex_colors(vtx_off, vtx_tri, rgb):
    off, adj, rgb = _i32(vtx_off), _i32(vtx_tri), _f32(rgb)
    V = off.shape[0] - 1
    out = np.empty((V, 3), np.float32)
    if adj.size == 0:
        adj = np.zeros(1, np.int32)
    lib().orc_vertex_colors(V, _p(off), _p(adj), _p(rgb), _p(out))
    return out


def num_threads():
    return lib().orc_num_threads()


# ---- the reference's vendored glm/Eigen arithmetic (container-built, optional) ----
_ref = None


def ref_available():
    return os.path.exists(_REF)


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(_REF)
        fp, ip = C.c_void_p, C.c_void_p
        R.ref_surface.restype = C.c_float
        R.ref_surface.argtypes = [fp, fp, fp]
        R.ref_p2p_integrand.restype = C.c_float
        R.ref_p2p_integrand.argtypes = [fp, fp, ip, ip, C.c_int, C.c_int]
        R.ref_uv2xyz.argtypes = [fp, ip, C.c_int, C.c_float, C.c_float, fp]
        R.ref_light_pass.argtypes = [C.c_int, C.c_int, fp, fp, ip, fp, fp, C.c_int]
        R.ref_sum.restype = C.c_float
        R.ref_sum.argtypes = [C.c_int, fp]
        R.ref_ini_open.restype = C.c_void_p
        R.ref_ini_open.argtypes = [C.c_char_p]
        R.ref_ini_free.argtypes = [C.c_void_p]
        R.ref_ini_error.argtypes = [C.c_void_p]
        R.ref_ini_get.restype = C.c_char_p
        R.ref_ini_get.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
        R.ref_ini_integer.restype = C.c_long
        R.ref_ini_integer.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_long]
        R.ref_ini_real.restype = C.c_double
        R.ref_ini_real.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_double]
        R.ref_ini_boolean.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
        R.ref_rgb2spec_spectrum.argtypes = [C.c_char_p, fp, fp, C.c_int, fp]
        R.ref_fcache_write.argtypes = [C.c_char_p, C.c_int, fp]
        R.ref_fcache_read.argtypes = [C.c_char_p, C.c_int, fp]
        R.ref_obj_load.restype = C.c_void_p
        R.ref_obj_load.argtypes = [C.c_char_p, C.c_char_p]
        R.ref_obj_free.argtypes = [C.c_void_p]
        R.ref_obj_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 5
        R.ref_obj_copy.argtypes = [C.c_void_p] + [fp] * 8
        R.ref_obj_material_name.restype = C.c_char_p
        R.ref_obj_material_name.argtypes = [C.c_void_p, C.c_int]
        R.ref_xyz_fit.argtypes = [C.c_double, fp]
        R.ref_patch_color_spectral.argtypes = [C.c_int, fp, fp, fp]
        R.ref_vertex_color.argtypes = [C.c_int, ip, fp, fp]
        _ref = R
    return _ref


def ref_surface(a, b, c):
    a, b, c = _f32(a), _f32(b), _f32(c)
    return float(ref().ref_surface(_p(a), _p(b), _p(c)))


def ref_p2p_integrand(mesh, i, j):
    return np.float32(ref().ref_p2p_integrand(_p(mesh.vertices), _p(mesh.normals), _p(mesh.tri_v),
                                              _p(mesh.tri_n), int(i), int(j)))


def ref_uv2xyz(mesh, tri, u, v):
    out = np.empty(3, np.float32)
    ref().ref_uv2xyz(_p(mesh.vertices), _p(mesh.tri_v), int(tri), C.c_float(u), C.c_float(v), _p(out))
    return out


def ref_light_pass(F, M, mat, R, B, mode):
    F, M, mat = _f32(F), _f32(M), _i32(mat)
    R, B = _f32(R).copy(), _f32(B).copy()
    ref().ref_light_pass(F.shape[0], R.shape[1], _p(F), _p(M), _p(mat), _p(R), _p(B), int(mode))
    return R, B


def ref_sum(x):
    x = _f32(x)
    return float(ref().ref_sum(x.size, _p(x)))


def ref_xyz_fit(wavelengths):
    out = np.empty((len(wavelengths), 3), np.float32)
    for k, w in enumerate(wavelengths):
        o = np.empty(3, np.float32)
        ref().ref_xyz_fit(C.c_double(float(w)), _p(o))
        out[k] = o
    return out


def ref_patch_colors_spectral(B, xyz):
    B, xyz = _f32(B), _f32(xyz)
    out = np.empty((B.shape[0], 3), np.float32)
    for i in range(B.shape[0]):
        o = np.empty(3, np.float32)
        b = np.ascontiguousarray(B[i])
        ref().ref_patch_color_spectral(B.shape[1], _p(xyz), _p(b), _p(o))
        out[i] = o
    return out


def ref_vertex_colors(vtx_off, vtx_tri, rgb):
    off, adj, rgb = _i32(vtx_off), _i32(vtx_tri), _f32(rgb)
    V = off.shape[0] - 1
    out = np.zeros((V, 3), np.float32)
    for v in range(V):
        n = int(off[v + 1] - off[v])
        if n == 0:
            continue
        a = np.ascontiguousarray(adj[off[v]:off[v + 1]])
        o = np.empty(3, np.float32)
        ref().ref_vertex_color(n, _p(a), _p(rgb), _p(o))
        out[v] = o
    return out


def ref_ini_query(path, queries):
    """the reference's own INIReader on the file at `path`: queries = [(kind, section, name, default)], kind in
    get / integer / real / boolean; returns (ParseError(), [values])"""
    R = ref()
    h = R.ref_ini_open(path.encode())
    out = []
    for kind, sec, name, default in queries:
        sec, name = sec.encode(), name.encode()
        if kind == "get":
            out.append(R.ref_ini_get(h, sec, name, default.encode()).decode())
        elif kind == "integer":
            out.append(int(R.ref_ini_integer(h, sec, name, int(default))))
        elif kind == "real":
            out.append(float(R.ref_ini_real(h, sec, name, float(default))))
        else:
            out.append(int(R.ref_ini_boolean(h, sec, name, int(default))))
    err = int(R.ref_ini_error(h))
    R.ref_ini_free(h)
    return err, out


def ref_rgb2spec_spectrum(path, rgb, wavelengths):
    """Material::rgb_to_spectrum through the reference's own rgb2spec.cpp on the coefficient table at `path`"""
    rgb, wl = _f32(rgb), _f32(wavelengths)
    out = np.zeros(wl.size, np.float32)
    ok = ref().ref_rgb2spec_spectrum(path.encode(), _p(rgb), _p(wl), wl.size, _p(out))
    return int(ok), out


def ref_obj_load(obj, mtl_dir):
    """the vendored tinyobjloader as MeshS::loadFromFile calls it; dict of the arrays MeshS takes from it"""
    R = ref()
    h = R.ref_obj_load(obj.encode(), mtl_dir.encode())
    ok, V, Nn, N, nm = (C.c_int() for _ in range(5))
    R.ref_obj_counts(h, C.byref(ok), C.byref(V), C.byref(Nn), C.byref(N), C.byref(nm))
    out = dict(vertices=np.zeros((V.value, 3), np.float32), normals=np.zeros((Nn.value, 3), np.float32),
               tri_v=np.zeros((N.value, 3), np.int32), tri_n=np.zeros((N.value, 3), np.int32), mat=np.zeros(N.value, np.int32),
               Kd=np.zeros((nm.value, 3), np.float32), Ke=np.zeros((nm.value, 3), np.float32), Ks=np.zeros((nm.value, 3), np.float32))
    R.ref_obj_copy(h, *[_p(out[k]) for k in ("vertices", "normals", "tri_v", "tri_n", "mat", "Kd", "Ke", "Ks")])
    out["names"] = [R.ref_obj_material_name(h, m).decode() for m in range(nm.value)]
    out["ok"] = ok.value
    R.ref_obj_free(h)
    return out


def ref_fcache_write(path, dense):
    """SerializeMat (vs/Lightning.h:21-45) of the dense N x N matrix through the reference's Eigen"""
    dense = _f32(dense)
    return int(ref().ref_fcache_write(path.encode(), dense.shape[0], _p(dense)))


def ref_fcache_read(path, N):
    """DeserializeMat (vs/Lightning.h:46-74) through the reference's Eigen; dense N x N or None"""
    out = np.zeros((N, N), np.float32)
    return out if ref().ref_fcache_read(path.encode(), int(N), _p(out)) else None
