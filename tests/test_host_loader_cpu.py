"""The C++ host above the C ABI (daisyriot_amd/host): config.ini reader and MeshS / Material
loader keep the reference's loading surface.  CPU only; the library is
reached through its small C view (host_capi.cpp)."""
import ctypes as C
import os

import numpy as np
import pytest

from daisyriot_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "daisyriot_amd", "lib", "libdaisyriot_host.so")
REF_SCENES = "/root/reference/example_scenes"
WL9 = np.arange(200.0, 601.0, 50.0, dtype=np.float32)


@pytest.fixture(scope="module")
def host():
    C.CDLL(os.path.join(ROOT, "daisyriot_amd", "lib", "libdaisyriot_hip.so"), mode=C.RTLD_GLOBAL)
    L = C.CDLL(HOST)
    L.drh_mesh_load.restype = C.c_void_p
    L.drh_mesh_load.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int]
    L.drh_mesh_free.argtypes = [C.c_void_p]
    L.drh_mesh_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
    L.drh_mesh_warnings.restype = C.c_char_p
    L.drh_mesh_warnings.argtypes = [C.c_void_p]
    L.drh_mesh_copy.argtypes = [C.c_void_p] * 6
    L.drh_mesh_materials.argtypes = [C.c_void_p] * 7
    L.drh_vertex_fanout.argtypes = [C.c_void_p, C.c_int]
    for f in ("drh_ini_open", "drh_ini_parse"):
        getattr(L, f).restype = C.c_void_p
        getattr(L, f).argtypes = [C.c_char_p]
    L.drh_ini_free.argtypes = [C.c_void_p]
    L.drh_ini_error.argtypes = [C.c_void_p]
    L.drh_ini_get.restype = C.c_char_p
    L.drh_ini_get.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
    L.drh_ini_integer.restype = C.c_long
    L.drh_ini_integer.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_long]
    L.drh_ini_real.restype = C.c_double
    L.drh_ini_real.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_double]
    L.drh_ini_boolean.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
    L.drh_visibility_samples.argtypes = [C.c_int, C.c_uint, C.c_void_p]
    return L


def load(host, obj, mtl_dir, wl=WL9):
    wl = np.ascontiguousarray(wl, np.float32)
    h = host.drh_mesh_load(obj.encode(), mtl_dir.encode(), wl.ctypes.data_as(C.c_void_p), wl.size)
    V, Nn, N, nm = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    host.drh_mesh_counts(h, C.byref(V), C.byref(Nn), C.byref(N), C.byref(nm))
    V, Nn, N, nm, S = V.value, Nn.value, N.value, nm.value, wl.size
    out = dict(vertices=np.zeros((V, 3), np.float32), normals=np.zeros((Nn, 3), np.float32),
               tri_v=np.zeros((N, 3), np.int32), tri_n=np.zeros((N, 3), np.int32), mat=np.zeros(N, np.int32))
    host.drh_mesh_copy(h, *[out[k].ctypes.data_as(C.c_void_p) for k in ("vertices", "normals", "tri_v", "tri_n", "mat")])
    m = dict(kind=np.zeros(nm, np.int32), rgb=np.zeros((nm, 3), np.float32), emission=np.zeros((nm, 3), np.float32),
             spectral_values=np.zeros((nm, S), np.float32), spectral_emission=np.zeros((nm, S), np.float32),
             M=np.zeros((nm, S, S), np.float32))
    host.drh_mesh_materials(h, *[m[k].ctypes.data_as(C.c_void_p) for k in
                                 ("kind", "rgb", "emission", "spectral_values", "spectral_emission", "M")])
    out.update(m)
    out["warnings"] = host.drh_mesh_warnings(h).decode()
    out["fanout0"] = host.drh_vertex_fanout(h, int(out["tri_v"][0, 0])) if N else 0
    host.drh_mesh_free(h)
    return out


INI_TEXT = b"""; Example config file for DaisyRiot application

[window]             ; Window configuration
width = 800
height = 600

[filepaths]\t\t\t; Necesarry directories
scene = example_scenes/cornellbox_blacklight.obj
mtl_dir = example_scenes/

[drawing]
radiosityRendering = true
antiAliasing = true
supersampling = 4

[lightning]\t\t\t; Lightning configuration
emission_value = 7.0 ; Best to adjust this value per scene
method = 2\t\t\t; 0 = BW, 1 = RGB, 2 = Spectral

[acceleration]
cuda_on = true
hexval : 0x10
"""


def test_ini_reader_reads_the_reference_keys(host, tmp_path):
    p = tmp_path / "config.ini"
    p.write_bytes(INI_TEXT)
    for h in (host.drh_ini_open(str(p).encode()), host.drh_ini_parse(INI_TEXT)):
        assert host.drh_ini_error(h) == 0
        assert host.drh_ini_integer(h, b"window", b"width", -1) == 800                    # main.cpp:68
        assert host.drh_ini_real(h, b"lightning", b"emission_value", -1) == 7.0          # inline comment cut
        assert host.drh_ini_integer(h, b"LIGHTNING", b"Method", 0) == 2                  # case-insensitive
        assert host.drh_ini_boolean(h, b"acceleration", b"cuda_on", 0) == 1
        assert host.drh_ini_get(h, b"filepaths", b"scene", b"UNKNOWN") == b"example_scenes/cornellbox_blacklight.obj"
        assert host.drh_ini_get(h, b"filepaths", b"nope", b"UNKNOWN") == b"UNKNOWN"
        assert host.drh_ini_integer(h, b"acceleration", b"hexval", 0) == 16
        assert host.drh_ini_boolean(h, b"drawing", b"supersampling", 1) == 1              # "4" is not a boolean -> default
        host.drh_ini_free(h)
    h = host.drh_ini_open(b"/nonexistent/config.ini")
    assert host.drh_ini_error(h) == -1                                                   # main.cpp:64-67 prints and exits
    host.drh_ini_free(h)
    h = host.drh_ini_parse(b"[a]\nok = 1\nthis line has no separator\nx = 2\n")
    assert host.drh_ini_error(h) == 3 and host.drh_ini_integer(h, b"a", b"x", 0) == 2
    host.drh_ini_free(h)


def test_obj_round_trip(host, tmp_path):
    sc = scenes.cornell_box(500, S=3)
    scenes.write_obj(sc, str(tmp_path / "box.obj"), "box.mtl")
    scenes.write_mtl(sc, str(tmp_path / "box.mtl"))
    got = load(host, str(tmp_path / "box.obj"), str(tmp_path) + "/")
    assert got["warnings"] == ""
    assert np.array_equal(got["vertices"], sc.vertices) and np.array_equal(got["normals"], sc.normals)
    assert np.array_equal(got["tri_v"], sc.tri_v) and np.array_equal(got["tri_n"], sc.tri_n)
    assert np.array_equal(got["mat"], sc.mat_of_patch)
    assert np.array_equal(got["rgb"], sc.rho) and np.array_equal(got["emission"], sc.emit)
    assert got["fanout0"] >= 1
    assert (got["kind"] == 0).all()


def test_obj_dialects_and_repairs(host, tmp_path):
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1 0 0\nKe 0 0 0\nnewmtl lamp\nKd 0 0 0\nKe 1 1 1\n"
                                    "newmtl glow\nKd 0.5 0.5 0.5\nKs 0 0.8 1\n")
    (tmp_path / "t.obj").write_text(
        "mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\n"
        "f 1//1 2//1 3//1\n"                  # no material yet -> default grey
        "usemtl a\nf 1/1/1 2/1/1 3/1/1 4/1/1\n"  # quad -> two triangles
        "usemtl lamp\nf -4//-1 -3//-1 -2//-1\n"  # negative (relative) indices
        "usemtl glow\nf 1 2 4\n")              # no normals -> geometric normal
    got = load(host, str(tmp_path / "t.obj"), str(tmp_path))
    assert got["tri_v"].tolist() == [[0, 1, 2], [0, 1, 2], [0, 2, 3], [0, 1, 2], [0, 1, 3]]
    assert got["mat"].tolist() == [3, 0, 0, 1, 2]
    assert got["kind"].tolist() == [0, 1, 2, 0]                 # plain, UV light, fluorescent, default
    assert np.allclose(got["normals"][got["tri_n"][4, 0]], [0, 0, 1])
    for w in ("fan-triangulated", "geometric normal", "default grey"):
        assert w in got["warnings"]
    # UV lamp (Material.cpp:47-78): bell curve around 350 nm, reflects nothing
    assert got["spectral_emission"][1].argmax() == 3 and np.all(got["M"][1] == 0)
    # fluorescent (Material.cpp:90-100): identity with the 350 nm column replaced
    Mf = got["M"][2]
    assert np.allclose(np.diag(Mf)[[0, 1, 2, 4, 5, 6, 7, 8]], 1.0)
    assert np.all(Mf[:, 3] <= 0.98 + 1e-6) and Mf[:, 3].max() > 0.1
    off = Mf.copy()
    off[:, 3] = 0
    np.fill_diagonal(off, 0)
    assert np.all(off == 0)
    # plain (Material.cpp:17-20): M = diag(diffuse spectrum)
    assert np.allclose(got["M"][0], np.diag(got["spectral_values"][0]))


@pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="reference example scenes not present on this box")
def test_reference_example_scenes_load_unchanged(host):
    got = load(host, REF_SCENES + "/cornellbox_blacklight.obj", REF_SCENES + "/")
    assert got["tri_v"].shape[0] == 7712 and got["vertices"].shape[0] == 4360 and got["normals"].shape[0] == 304
    assert got["kind"].tolist() == [1, 2, 2, 0]          # Blacklight, Blacklight_Pink, Blacklight_blue, white (MeshS.cpp:41-63)
    assert got["warnings"] == ""
    got = load(host, REF_SCENES + "/colorballs.obj", REF_SCENES + "/")
    assert got["tri_v"].shape[0] == 6400 and got["vertices"].shape[0] == 3373 and got["normals"].shape[0] == 2551
    assert got["rgb"].shape[0] == 5 and (got["emission"].sum(axis=1) > 0).sum() == 4      # 4 coloured emitters + white


def test_visibility_samples_follow_the_reference_formula(host):
    uv = np.zeros((50, 2), np.float32)
    host.drh_visibility_samples(50, 20191, uv.ctypes.data_as(C.c_void_p))
    assert (uv >= 0).all() and (uv[:, 0] < 1).all() and (uv.sum(axis=1) <= 1 + 1e-6).all()
    uv2 = np.zeros((50, 2), np.float32)
    host.drh_visibility_samples(50, 20191, uv2.ctypes.data_as(C.c_void_p))
    assert np.array_equal(uv, uv2)


def test_xyz_fit_table_equals_the_reference_header(host):
    # the host adapter's xyz_per_wavelength (color.h:14-45) against the values the reference's own header gave
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "display_color_h.npz"))
    host.drh_xyz_fit.argtypes = [C.c_double, C.c_void_p]
    got = np.zeros((len(g["wavelengths"]), 3), np.float32)
    for k, w in enumerate(g["wavelengths"]):
        o = np.zeros(3, np.float32)
        host.drh_xyz_fit(float(w), o.ctypes.data_as(C.c_void_p))
        got[k] = o
    assert np.array_equal(got.view(np.uint32), g["xyz_bits"])


def _host_ini_query(host, path, queries):
    h = host.drh_ini_open(path.encode())
    out = []
    for kind, sec, name, default in queries:
        sec, name = sec.encode(), name.encode()
        if kind == "get":
            out.append(host.drh_ini_get(h, sec, name, default.encode()).decode())
        elif kind == "integer":
            out.append(int(host.drh_ini_integer(h, sec, name, int(default))))
        elif kind == "real":
            out.append(float(host.drh_ini_real(h, sec, name, float(default))))
        else:
            out.append(int(host.drh_ini_boolean(h, sec, name, int(default))))
    err = int(host.drh_ini_error(h))
    host.drh_ini_free(h)
    return err, out


def test_ini_reader_equals_the_reference_inireader(host, tmp_path):
    """golden values produced by the reference's own INIReader.h (tests/golden/make_golden.py host): duplicate keys,
    continuation lines, inline comments, ':' separators, hex and negative integers, boolean spellings, defaults"""
    import json
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ini_inireader.json")))
    p = tmp_path / "config.ini"
    p.write_bytes(g["ini_text"].encode())
    queries = [tuple(q) for q in g["queries"]]
    err, vals = _host_ini_query(host, str(p), queries)
    assert err == g["parse_error"]
    for q, got, want in zip(queries, vals, g["values"]):
        assert got == want, (q, got, want)
    b = tmp_path / "bad.ini"
    b.write_bytes(g["bad_text"].encode())
    err, vals = _host_ini_query(host, str(b), [("integer", "a", "x", 0), ("integer", "a", "ok", 0)])
    assert err == g["bad_parse_error"] and vals == g["bad_values"]
    err, _ = _host_ini_query(host, str(tmp_path / "missing.ini"), [])
    assert err == g["missing_parse_error"]
    from oracle import binding as ob
    if ob.ref_available():                                  # and live, when the checker library is there
        assert ob.ref_ini_query(str(p), queries) == (g["parse_error"], g["values"])


def test_spectrum_lookup_equals_the_reference_rgb2spec(host, tmp_path):
    """rgb2spec_fetch + rgb2spec_eval_precise of the reference (vs/rgb2spec.cpp:78-134, as Material.cpp:35-45 calls
    them) on a small coefficient table: the host's SpectralUpsampler gives the same bits"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rgb2spec_lookup.npz"))
    table = tmp_path / "srgb8.coeff"
    table.write_bytes(g["table"].tobytes())
    host.drh_upsample.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    wl = np.ascontiguousarray(g["wavelengths"], np.float32)
    want = g["spectra_bits"]
    for k, rgb in enumerate(np.ascontiguousarray(g["rgb"], np.float32)):
        out = np.zeros(wl.size, np.float32)
        assert host.drh_upsample(str(table).encode(), rgb.ctypes.data_as(C.c_void_p), wl.ctypes.data_as(C.c_void_p), wl.size,
                                 out.ctypes.data_as(C.c_void_p)) == 1
        assert np.array_equal(out.view(np.uint32), want[k]), (k, rgb)
    from oracle import binding as ob
    if ob.ref_available():
        ok, sp = ob.ref_rgb2spec_spectrum(str(table), g["rgb"][7], wl)
        assert ok == 1 and np.array_equal(sp.view(np.uint32), want[7])


def test_obj_mtl_parsing_equals_the_vendored_tinyobjloader(host, tmp_path):
    """expected arrays produced by the reference's vendored tinyobjloader, called as vs/MeshS.cpp:25-31 calls it
    (tests/golden/make_golden.py obj): v//vn and v/vt/vn corners, relative indices, groups, CR line ends, tabs,
    several usemtl switches, unknown .mtl keys"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "obj_tinyobj.npz"))
    (tmp_path / "case.obj").write_bytes(g["obj_text"].tobytes())
    (tmp_path / "case.mtl").write_bytes(g["mtl_text"].tobytes())
    got = load(host, str(tmp_path / "case.obj"), str(tmp_path) + "/")
    for k in ("vertices", "normals", "tri_v", "tri_n", "mat"):
        assert np.array_equal(got[k], g[k]), k
    assert np.array_equal(got["rgb"], g["Kd"])                         # Material::rgbcolor = Kd (MeshS.cpp:37)
    # classification (MeshS.cpp:41-63): Ks > 0 -> fluorescent; emission of plain materials = Ke
    assert got["kind"].tolist() == [0, 0, 2, 0]
    assert np.array_equal(got["emission"][[0, 1, 3]], g["Ke"][[0, 1, 3]])
    from oracle import binding as ob
    if ob.ref_available():
        live = ob.ref_obj_load(str(tmp_path / "case.obj"), str(tmp_path) + "/")
        for k in ("vertices", "normals", "tri_v", "tri_n", "mat", "Kd", "Ke", "Ks"):
            assert np.array_equal(live[k], g[k]), k
        if os.path.isdir(REF_SCENES):
            for name in ("cornellbox_blacklight", "colorballs"):
                t = ob.ref_obj_load("%s/%s.obj" % (REF_SCENES, name), REF_SCENES + "/")
                mine = load(host, "%s/%s.obj" % (REF_SCENES, name), REF_SCENES + "/")
                for k in ("vertices", "normals", "tri_v", "tri_n", "mat"):
                    assert np.array_equal(t[k], mine[k]), (name, k)
                assert np.array_equal(t["Kd"], mine["rgb"])


def test_f_matrix_cache_format_equals_eigens(host, tmp_path):
    """the file SerializeMat writes (vs/Lightning.h:21-45; bytes produced by that call sequence on the reference's
    Eigen, tests/golden/make_golden.py fcache): the host reads it, and writes the same bytes for the same matrix"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fcache_eigen.npz"))
    dense = np.ascontiguousarray(g["dense"], np.float32)
    N = dense.shape[0]
    host.drh_fcache_write.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    host.drh_fcache_read.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    ref_file = tmp_path / "eigen.bin"
    ref_file.write_bytes(g["file_bytes"].tobytes())
    got = np.zeros((N, N), np.float32)
    assert host.drh_fcache_read(str(ref_file).encode(), N, got.ctypes.data_as(C.c_void_p)) == 1
    assert np.array_equal(got, dense)
    mine = tmp_path / "mine.bin"
    assert host.drh_fcache_write(str(mine).encode(), N, dense.ctypes.data_as(C.c_void_p)) == 1
    assert mine.read_bytes() == ref_file.read_bytes()
    assert host.drh_fcache_read(str(ref_file).encode(), N + 1, got.ctypes.data_as(C.c_void_p)) == 0      # wrong size: refused
    from oracle import binding as ob
    if ob.ref_available():
        assert np.array_equal(ob.ref_fcache_read(str(mine), N), dense)      # and Eigen reads the host's file


def test_ini_reader_fuzz_against_the_reference_inireader(host, tmp_path):
    """random ini texts (sections, ':' and '=' separators, inline and full-line comments, continuation lines, empty
    values, duplicate keys, malformed lines, CR line ends): every lookup and the error line number equal the reference's
    own INIReader.h (live; needs the checker library oracle/_ref)"""
    import random
    from oracle import binding as ob
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rnd = random.Random(7)
    secs = ["a", "B", "window", "Sec Tion", "x.y"]
    names = ["k", "Key", "width", "v a l", "n1", "on", "x"]
    vals = ["1", "0", "true", "False", "yes", "no", "on", "off", "0x1F", "-3", "4.5e2", "abc", "a b c", "  padded  ", "", "7 ; c",
            "8 # c", "\"q\"", "1,2", "TRUE", "On"]

    def line():
        r = rnd.random()
        if r < 0.18:
            return "[%s]%s" % (rnd.choice(secs), rnd.choice(["", " ", "  ; c", "\t# c"]))
        if r < 0.26:
            return rnd.choice(["; comment", "# comment", "", "   ", "\t"])
        if r < 0.32:
            return rnd.choice(["no separator here", "[unclosed", "  continuation text", "= novalue name", " [a]", "[a ;c]", "k ;c = 1"])
        sep = rnd.choice(["=", ":", " = ", " : ", "\t=\t"])
        lead = rnd.choice(["", " ", "\t"]) if rnd.random() < 0.15 else ""
        return "%s%s%s%s%s" % (lead, rnd.choice(names), sep, rnd.choice(vals), rnd.choice(["", " ", "\r", " ; tail"]))

    queries = [(k, s, n, d) for s in secs + ["a ", "A"] for n in names + ["KEY"]
               for k, d in (("get", "D"), ("integer", -9), ("real", -1.5), ("boolean", 1), ("boolean", 0))]
    for it in range(300):
        text = "\n".join(line() for _ in range(rnd.randint(3, 25))) + rnd.choice(["", "\n"])
        p = tmp_path / ("f%d.ini" % it)
        p.write_text(text)
        assert _host_ini_query(host, str(p), queries) == ob.ref_ini_query(str(p), queries), text


def _random_obj(rnd):
    """triangles whose corners all carry a normal and whose materials exist (nothing for the loader to repair)"""
    def fnum():
        return rnd.choice(["%d" % rnd.randint(-3, 3), "%.3f" % rnd.uniform(-2, 2), "%.2e" % rnd.uniform(-2, 2), "-.5", ".25", "1.", "+2.5",
                           "%.9f" % rnd.uniform(-1, 1), "%.7g" % rnd.uniform(-1e3, 1e3), "1e", "3e-2x", "12.5000001", "0.1", "-0.30000001"])
    nv, nn, nt = rnd.randint(3, 9), rnd.randint(1, 4), rnd.randint(0, 3)
    lines = ["mtllib m.mtl", "# c", "usemtl a"]
    for _ in range(nv):
        lines.append("v%s%s %s %s%s" % (rnd.choice([" ", "  ", "\t"]), fnum(), fnum(), fnum(), rnd.choice(["", " ", " 1.0"])))
    for _ in range(nn):
        lines.append("vn %s %s %s" % (fnum(), fnum(), fnum()))
    for _ in range(nt):
        lines.append("vt %s %s" % (fnum(), fnum()))
    for _ in range(rnd.randint(1, 8)):
        r = rnd.random()
        if r < 0.15:
            lines.append("usemtl " + rnd.choice(["a", "b"]))
        elif r < 0.25:
            lines.append(rnd.choice(["g grp", "o obj", "s off", "s 1", "", "# x", "g"]))
        else:
            def corner():
                vi = rnd.randint(1, nv)
                vi = vi if rnd.random() < 0.8 else vi - nv - 1
                ni = rnd.randint(1, nn)
                ni = ni if rnd.random() < 0.8 else ni - nn - 1
                if nt and rnd.random() < 0.4:
                    return "%d/%d/%d" % (vi, rnd.randint(1, nt), ni)
                return "%d//%d" % (vi, ni)
            lines.append("f " + rnd.choice([" ", "  ", "\t"]).join(corner() for _ in range(3)) + rnd.choice(["", " ", "\r"]))
    return "\n".join(lines) + "\n"


def test_obj_reader_fuzz_against_the_vendored_tinyobjloader(host, tmp_path):
    """random OBJ/MTL files (corner forms with normals, relative indices, tabs, CR, groups, odd number spellings that
    the parser reads its own way: "-.5", "1.", "1e", nine-digit decimals): the arrays MeshS takes from the parser are
    bit for bit tinyobjloader's (live; needs oracle/_ref)"""
    import random
    from oracle import binding as ob
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rnd = random.Random(3)
    for it in range(300):
        kd = " ".join("%.6f" % rnd.uniform(0, 1) for _ in range(rnd.choice([1, 2, 3])))
        (tmp_path / "m.mtl").write_text("newmtl a\nKd %s\nKe 1 1 1\nnewmtl b\nKd 0.5 0.5 0.5\nKs 0.1 0 0\nnewmtl c\nKe .5 0.25\n" % kd)
        text = _random_obj(rnd)
        (tmp_path / "x.obj").write_text(text)
        want = ob.ref_obj_load(str(tmp_path / "x.obj"), str(tmp_path) + "/")
        got = load(host, str(tmp_path / "x.obj"), str(tmp_path) + "/")
        assert got["warnings"] == "", (got["warnings"], text)
        for k in ("vertices", "normals", "tri_v", "tri_n", "mat"):
            assert np.array_equal(want[k].view(np.uint32) if want[k].dtype == np.float32 else want[k],
                                  got[k].view(np.uint32) if got[k].dtype == np.float32 else got[k]), (k, want[k].tolist(), got[k].tolist(), text)
        assert np.array_equal(want["Kd"], got["rgb"]), (kd, want["Kd"].tolist(), got["rgb"].tolist())
