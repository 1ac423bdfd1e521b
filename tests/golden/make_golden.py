"""Generates tests/golden/*.npz in THIS container from the reference's vendored glm 0.9.8.4
and Eigen 3.2.10 (oracle/_ref/libref_check.so, built by oracle/Makefile from
/root/reference/libraries where they lie).  Fixtures are data only: inputs + the values the
reference's own math libraries produce at the hot path's call sites.

    python tests/golden/make_golden.py [core display host obj fcache scenes]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from daisyriot_amd import scenes  # noqa: E402
from oracle import binding as ob  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    assert ob.ref_available(), "build oracle/_ref first (make -C oracle)"
    rs = np.random.RandomState(7)
    uv = scenes.visibility_samples(50)

    # --- integrand + uv2xyz through glm ---------------------------------------------------
    sc = scenes.cornell_box(96, S=3)
    # perturb vertex normals so that avgNormal's average/normalise path is exercised
    nrm = sc.normals + rs.normal(scale=0.05, size=sc.normals.shape).astype(np.float32)
    tn = sc.tri_n.copy()
    tn[:, 1] = (tn[:, 1] + 1) % nrm.shape[0]
    m = ob.Mesh(sc.vertices, nrm, sc.tri_v, tn)
    pairs = rs.randint(0, sc.N, size=(600, 2)).astype(np.int32)
    pairs[:8] = [[k, k] for k in range(8)]                    # diagonal: NaN path -> stored 0
    ff = np.array([ob.ref_p2p_integrand(m, i, j) for i, j in pairs], np.float32)
    tri = rs.randint(0, sc.N, size=64).astype(np.int32)
    pts = np.array([ob.ref_uv2xyz(m, t, uv[k % 50, 0], uv[k % 50, 1]) for k, t in enumerate(tri)], np.float32)
    kat = np.float32(ob.ref_surface([0, 0, 0], [1, 0, 0], [0, 1, 0]))
    np.savez_compressed(os.path.join(OUT, "integrand_glm.npz"), vertices=m.vertices, normals=m.normals,
                        tri_v=m.tri_v, tri_n=m.tri_n, pairs=pairs, integrand_bits=ff.view(np.uint32),
                        uv=uv, uv_tri=tri, uv_points_bits=pts.view(np.uint32), unit_triangle_area=kat)

    # --- light passes through Eigen --------------------------------------------------------
    for name, S, fluor, mode in (("spectral9", 9, True, 0), ("rgb3", 3, False, 1), ("bw1", 1, False, 0)):
        sc = scenes.cornell_box(128, S=S, fluorescent=fluor)
        mm = ob.Mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        F, _, _ = ob.assemble_rows(mm, uv, bvh=True, want_vis=False)
        M = sc.M.copy()
        if name == "bw1":
            M[:] = 1.0                                            # BWLightning: no reflectance (Lightning.h:419-424)
        E = sc.emission(7.0)
        R, B = E.copy(), E.copy()
        snaps = {}
        done = 0
        for upto in (1, 2, 5, 20):
            for _ in range(upto - done):
                R, B = ob.ref_light_pass(F, M, sc.mat_of_patch, R, B, mode)
            done = upto
            snaps["R%d" % upto] = R.view(np.uint32).copy()
            snaps["B%d" % upto] = B.view(np.uint32).copy()
        sums = np.array([ob.ref_sum(R[:, s].copy()) for s in range(S)], np.float32)
        np.savez_compressed(os.path.join(OUT, "lightpass_%s_eigen.npz" % name), F=F, M=M, E=E,
                            mat=sc.mat_of_patch, mode=np.int32(mode), eigen_sums_R20=sums, **snaps)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))


def display():
    """Display colours through the reference's own color.h + glm (oracle/_ref): the spectral colour cache
    (vs/Lightning.h:168-183) and the vertex means of Drawer::interpolate (vs/Drawer.cpp:161-186)."""
    assert ob.ref_available()
    g = np.load(os.path.join(OUT, "lightpass_spectral9_eigen.npz"))
    sc = scenes.cornell_box(128, S=9, fluorescent=True)
    B = g["B20"].view(np.float32)
    # three exposure levels: all below 1, mixed, all normalised by their maximum
    B = np.concatenate([B * np.float32(0.02), B, B * np.float32(40.0)], axis=0)
    wl = scenes.WAVELENGTHS_9
    xyz = ob.ref_xyz_fit(wl)
    rgb = ob.ref_patch_colors_spectral(B, xyz)
    off, adj = scenes.vertex_adjacency(sc.tri_v, sc.vertices.shape[0])
    vtx = ob.ref_vertex_colors(off, adj, rgb[sc.N:2 * sc.N])
    np.savez_compressed(os.path.join(OUT, "display_color_h.npz"), wavelengths=wl, xyz_bits=xyz.view(np.uint32), B=B,
                        rgb_bits=rgb.view(np.uint32), tri_v=sc.tri_v, n_vertices=np.int32(sc.vertices.shape[0]),
                        vtx_off=off, vtx_tri=adj, vertex_rgb_bits=vtx.view(np.uint32))
    print("wrote display_color_h.npz: %d patches, %d above 1 before normalisation" % (B.shape[0], int((rgb.max(axis=1) >= 1).sum())))


INI_TEXT = b"""; a config in the shape of the reference's config_example.ini, plus the parser's corner cases

[window]             ; Window configuration
width = 800
height = 600

[filepaths]\t\t\t; Necesarry directories
scene = example_scenes/cornellbox_blacklight.obj
mtl_dir = example_scenes/

[drawing]
radiosityRendering = true
antiAliasing = TRUE
supersampling = 4

[lightning]\t\t\t; Lightning configuration
emission_value = 7.0 ; Best to adjust this value per scene
method = 2\t\t\t; 0 = BW, 1 = RGB, 2 = Spectral
method = 1
[acceleration]
cuda_on = yes
hexval : 0x10
negative = -12
float_as_int = 3.9
spaced key   =   spaced value   
# hash comment
off_switch = off
  indented = 5
empty =
"""

INI_QUERIES = [("integer", "window", "width", -1), ("integer", "window", "height", -1), ("integer", "window", "depth", -7),
               ("get", "filepaths", "scene", "UNKNOWN"), ("get", "filepaths", "mtl_dir", "UNKNOWN"), ("get", "filepaths", "nope", "UNKNOWN"),
               ("boolean", "drawing", "radiosityRendering", 0), ("boolean", "drawing", "antiAliasing", 0),
               ("boolean", "drawing", "supersampling", 1), ("integer", "drawing", "supersampling", 0),
               ("real", "lightning", "emission_value", -1.0), ("integer", "lightning", "method", 0), ("integer", "LIGHTNING", "Method", 0),
               ("get", "lightning", "method", ""), ("boolean", "acceleration", "cuda_on", 0), ("integer", "acceleration", "hexval", 0),
               ("integer", "acceleration", "negative", 0), ("integer", "acceleration", "float_as_int", 0), ("real", "acceleration", "float_as_int", 0.0),
               ("get", "acceleration", "spaced key", "?"), ("boolean", "acceleration", "off_switch", 1), ("integer", "acceleration", "indented", 0),
               ("get", "acceleration", "empty", "default"), ("real", "window", "width", 0.5)]


def host_surface():
    """config.ini parsing and RGB -> spectrum table lookup through the reference's own INIReader.h and rgb2spec.cpp
    (oracle/_ref), on a config of this file's own and on a small coefficient table made by lib/rgb2spec_opt"""
    import json
    import subprocess
    import tempfile
    assert ob.ref_available()
    with tempfile.TemporaryDirectory() as d:
        ini = os.path.join(d, "config.ini")
        open(ini, "wb").write(INI_TEXT)
        err, vals = ob.ref_ini_query(ini, INI_QUERIES)
        bad = os.path.join(d, "bad.ini")
        open(bad, "wb").write(b"[a]\nok = 1\nthis line has no separator\nx = 2\n")
        err_bad, vals_bad = ob.ref_ini_query(bad, [("integer", "a", "x", 0), ("integer", "a", "ok", 0)])
        err_missing, _ = ob.ref_ini_query(os.path.join(d, "missing.ini"), [])
        json.dump({"ini_text": INI_TEXT.decode(), "queries": INI_QUERIES, "parse_error": err, "values": vals,
                   "bad_text": "[a]\nok = 1\nthis line has no separator\nx = 2\n", "bad_parse_error": err_bad, "bad_values": vals_bad,
                   "missing_parse_error": err_missing}, open(os.path.join(OUT, "ini_inireader.json"), "w"), indent=1)
        table = os.path.join(d, "srgb8.coeff")
        subprocess.run([os.path.join(ROOT, "daisyriot_amd", "lib", "rgb2spec_opt"), "8", table], check=True, capture_output=True)
        rs = np.random.RandomState(11)
        rgb = rs.uniform(0, 1, size=(60, 3)).astype(np.float32)
        rgb[:6] = [[1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.63, 0.065, 0.05]]
        wl = np.concatenate([np.arange(200.0, 601.0, 50.0), np.arange(380.0, 781.0, 25.0)]).astype(np.float32)
        spectra = np.zeros((60, wl.size), np.float32)
        for k in range(60):
            ok, spectra[k] = ob.ref_rgb2spec_spectrum(table, rgb[k], wl)
            assert ok == 1
        np.savez_compressed(os.path.join(OUT, "rgb2spec_lookup.npz"), table=np.frombuffer(open(table, "rb").read(), np.uint8),
                            rgb=rgb, wavelengths=wl, spectra_bits=spectra.view(np.uint32))
    print("wrote ini_inireader.json (parse error %d) and rgb2spec_lookup.npz" % err)


OBJ_TEXT = b"""# corner cases of the OBJ dialect the reference's scenes use (Blender export) and a few more
mtllib case.mtl
o Cube
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
v 1 1 1
v 0 1 1
vn 0 0 -1
vn 0 0 1
vn 1 0 0
vt 0.5 0.5
usemtl wall
s off
f 1//1 3//1 2//1
f 1//1 4//1 3//1
g top   
usemtl lamp
f 5/1/2 6/1/2 7/1/2
f 5/1/2 7/1/2 8/1/2

# relative indices, a trailing comment and tabs
usemtl wall
f -8//-3 -7//-3\t-3//-1
o Second\r
v 2 0 0
v 3 0 0
v 2 1 0
vn 0 1 0
usemtl uv
f 9//4 10//4 11//4\r
usemtl glow
f 11//4 10//4 9//4
"""

MTL_TEXT = b"""# materials in the reference's conventions (MeshS.cpp:41-63): Ks marks fluorescent, the name Blacklight the UV lamp
newmtl wall
Ns 96.078431
Ka 1.000000 1.000000 1.000000
Kd 0.640000 0.050000 0.050000
Ks 0.000000 0.000000 0.000000
Ke 0.0 0.0 0.0
Ni 1.000000
d 1.000000
illum 2

newmtl lamp
Kd 0.78 0.78 0.78
Ke 1.0 0.9 0.8
Ks 0 0 0

newmtl uv
Kd 0.1 0.1 0.1
Ke 0 0 0
Ks 0.2 0.3 0.9
map_Kd unused.png

newmtl glow
Kd 0.5 0.5 0.5
Ks 0 0 0
"""


def obj_loader():
    """OBJ/MTL through the vendored tinyobjloader as MeshS::loadFromFile calls it (oracle/_ref): expected arrays for a
    synthetic file with the dialect's corner cases, plus the reference's two example scenes as a check on the run"""
    import tempfile
    assert ob.ref_available()
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "case.obj"), "wb").write(OBJ_TEXT)
        open(os.path.join(d, "case.mtl"), "wb").write(MTL_TEXT)
        r = ob.ref_obj_load(os.path.join(d, "case.obj"), d + "/")
    assert r["ok"] == 1
    np.savez_compressed(os.path.join(OUT, "obj_tinyobj.npz"), obj_text=np.frombuffer(OBJ_TEXT, np.uint8), mtl_text=np.frombuffer(MTL_TEXT, np.uint8),
                        vertices=r["vertices"], normals=r["normals"], tri_v=r["tri_v"], tri_n=r["tri_n"], mat=r["mat"],
                        Kd=r["Kd"], Ke=r["Ke"], Ks=r["Ks"], names=np.array(r["names"]))
    print("wrote obj_tinyobj.npz:", r["tri_v"].shape[0], "triangles,", r["names"])
    ex = "/root/reference/example_scenes"
    for name in ("cornellbox_blacklight", "colorballs"):
        t = ob.ref_obj_load("%s/%s.obj" % (ex, name), ex + "/")
        g = np.load(os.path.join(OUT, "scene_%s.npz" % name))
        for k in ("vertices", "normals", "tri_v", "tri_n", "mat", "Kd"):
            assert np.array_equal(t[k], g[k]), (name, k)        # the committed scene fixtures (made by the host loader) agree
    print("example scenes: host-loader fixtures equal tinyobj's arrays")


def reference_scenes():
    """The reference's two example scenes (data files under /root/reference/example_scenes) parsed by
    the C++ host loader and stored as arrays, so that BASELINE.json's configs 0 and 1 can be run
    where the reference tree is absent (the GPU box)."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_host_loader_cpu as hl
    C.CDLL(os.path.join(ROOT, "daisyriot_amd", "lib", "libdaisyriot_hip.so"), mode=C.RTLD_GLOBAL)
    L = C.CDLL(hl.HOST)
    L.drh_mesh_load.restype = C.c_void_p
    L.drh_mesh_load.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int]
    L.drh_mesh_free.argtypes = [C.c_void_p]
    L.drh_mesh_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
    L.drh_mesh_warnings.restype = C.c_char_p
    L.drh_mesh_warnings.argtypes = [C.c_void_p]
    L.drh_mesh_copy.argtypes = [C.c_void_p] * 6
    L.drh_mesh_materials.argtypes = [C.c_void_p] * 7
    L.drh_vertex_fanout.argtypes = [C.c_void_p, C.c_int]
    for name in ("cornellbox_blacklight", "colorballs"):
        got = hl.load(L, "/root/reference/example_scenes/%s.obj" % name, "/root/reference/example_scenes/", hl.WL9)
        np.savez_compressed(os.path.join(OUT, "scene_%s.npz" % name), vertices=got["vertices"], normals=got["normals"],
                            tri_v=got["tri_v"], tri_n=got["tri_n"], mat=got["mat"], kind=got["kind"],
                            Kd=got["rgb"], Ke=got["emission"])
        print(name, got["tri_v"].shape[0], "triangles")




def fcache():
    """the F-matrix disk cache written by SerializeMat's call sequence on the reference's Eigen (oracle/_ref)"""
    import tempfile
    assert ob.ref_available()
    rs = np.random.RandomState(23)
    N = 37
    dense = np.where(rs.random_sample((N, N)) < 0.3, rs.random_sample((N, N)), 0).astype(np.float32)
    dense[:, 5] = 0          # an empty column, an empty row, the last column empty: the outer-index edge cases
    dense[11, :] = 0
    dense[:, N - 1] = 0
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "mat.bin")
        assert ob.ref_fcache_write(p, dense) == 1
        raw = np.frombuffer(open(p, "rb").read(), np.uint8)
        assert np.array_equal(ob.ref_fcache_read(p, N), dense)
    np.savez_compressed(os.path.join(OUT, "fcache_eigen.npz"), dense=dense, file_bytes=raw)
    print("wrote fcache_eigen.npz:", raw.size, "bytes,", int((dense != 0).sum()), "non-zeros")


if __name__ == "__main__":
    # python tests/golden/make_golden.py [core display host obj fcache scenes]   (default: all)
    parts = {"core": main, "display": display, "host": host_surface, "obj": obj_loader, "fcache": fcache, "scenes": reference_scenes}
    for part in (sys.argv[1:] or list(parts)):
        parts[part]()
