"""End to end through the C++ host: config.ini + .obj/.mtl -> MeshS -> Lightning (factory,
converge, F cache) -> per-patch radiance, on the GPU, against the Python/ctypes route and the
oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from daisyriot_amd import api, scenes
from oracle import binding as ob

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "daisyriot_amd", "lib", "daisyriot_cli")
HOST = os.path.join(ROOT, "daisyriot_amd", "lib", "libdaisyriot_host.so")


def _host_uv():
    C.CDLL(api.LIB_PATH, mode=C.RTLD_GLOBAL)
    L = C.CDLL(HOST)
    uv = np.zeros((50, 2), np.float32)
    L.drh_visibility_samples(50, C.c_uint(20191), uv.ctypes.data_as(C.c_void_p))
    return uv


def _write_scene(tmp_path, n, method, cuda_on):
    sc = scenes.cornell_box(n, S=3)
    scenes.write_obj(sc, str(tmp_path / "box.obj"), "box.mtl")
    scenes.write_mtl(sc, str(tmp_path / "box.mtl"))
    (tmp_path / "config.ini").write_text(
        "[window]\nwidth = 800\nheight = 600\n[filepaths]\nscene = %s\nmtl_dir = %s/\n"
        "[drawing]\nradiosityRendering = true\n[lightning]\nemission_value = 7.0 ; per scene\nmethod = %d\n"
        "[acceleration]\ncuda_on = %s\n" % (tmp_path / "box.obj", tmp_path, method, "true" if cuda_on else "false"))
    return sc


def _run(tmp_path, *extra):
    out = tmp_path / "out.csv"
    r = subprocess.run([CLI, str(tmp_path / "config.ini"), "--out", str(out)] + list(extra),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    data = np.loadtxt(out, delimiter=",", skiprows=1)
    return r.stdout, data


@pytest.mark.parametrize("cuda_on", [True, False])
def test_rgb_scene_through_the_cli(tmp_path, cuda_on):
    sc = _write_scene(tmp_path, 420, 1, cuda_on)
    stdout, data = _run(tmp_path, "--no-matfile")
    assert "Number of triangles: 420" in stdout
    B_cli = data[:, 4:7].astype(np.float32)
    assert np.array_equal(data[:, 1:4].astype(np.float32), B_cli)          # RGB colour = B (Lightning.h:332-334)
    # same thing through ctypes, and through the oracle
    uv = _host_uv()
    E = sc.emission(7.0)
    rule = api.RULE_INTEGRAND if cuda_on else api.RULE_RECIPROCITY
    with api.Context(0) as c:
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        c.assemble(uv, rule=rule)
        F = c.read_rows(0, sc.N)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        it = c.converge(1e-4, per_bin=True, max_iters=100000)
        B_py, _ = c.read()
    assert ("Number of light passes %d." % it) in stdout
    assert np.allclose(B_cli, B_py, rtol=1e-6, atol=1e-9)
    Fo, _, _ = ob.assemble_rows(ob.Mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n), uv, rule=rule, bvh=True, want_vis=False)
    assert np.array_equal(F.view(np.uint32), Fo.view(np.uint32))
    it_o, _, B_o = ob.converge(Fo, sc.M, sc.mat_of_patch, E, 1e-4, True, 100000)
    assert it_o == it
    assert (np.abs(B_cli - B_o) / np.abs(B_o).max(axis=0)).max() < 1e-4      # north_star's bar


def test_f_cache_round_trip_and_extra_passes(tmp_path):
    _write_scene(tmp_path, 300, 1, True)
    out1, d1 = _run(tmp_path)
    assert "Loaded & Serialized matrix" in out1 and os.path.exists(tmp_path / "box")
    # header of the reference's format (Lightning.h:38-46): rows, cols, nnz, outerSize, innerSize
    hdr = np.fromfile(tmp_path / "box", dtype=np.int32, count=5)
    assert hdr[0] == hdr[1] == hdr[3] == hdr[4] == 300 and 0 < hdr[2] < 300 * 300
    out2, d2 = _run(tmp_path)
    assert "Deserialized matrix" in out2
    assert np.array_equal(d1, d2)
    out3, d3 = _run(tmp_path, "--passes", "3", "--ply", str(tmp_path / "view.ply"))   # three presses of 'L'
    assert (d3[:, 4:] >= d2[:, 4:]).all() and (d3[:, 4:] > d2[:, 4:]).any()
    # vertex-averaged display colours (the corner values of Drawer::interpolate) as a viewable mesh
    ply = (tmp_path / "view.ply").read_text().splitlines()
    nv = int([l for l in ply if l.startswith("element vertex")][0].split()[-1])
    body = ply[ply.index("end_header") + 1:]
    assert len(body) == nv + 300 and all(0 <= int(x) <= 255 for x in body[0].split()[3:6])
    cols = np.array([[int(x) for x in l.split()[3:6]] for l in body[:nv]])
    assert cols.max() > 50                                                     # the lamp lights the room


def test_spectral_and_bw_methods_run(tmp_path):
    _write_scene(tmp_path, 300, 2, True)
    out, d = _run(tmp_path, "--no-matfile", "--passes", "4")
    assert d.shape == (300, 4 + 9)                                            # nine bins (main.cpp:94)
    assert np.isfinite(d).all() and (d[:, 1:4] <= 1 + 1e-6).all()             # display colour normalised (Lightning.h:177-180)
    assert "Number of light passes 4." in out                                 # 63 < 200: converge takes no pass
    # BW has no reflectance (Lightning.h:419-424) and never converges in a closed room: cap it
    ini = (tmp_path / "config.ini").read_text().replace("method = 2", "method = 0\nmax_passes = 12")
    (tmp_path / "config.ini").write_text(ini)
    out, d = _run(tmp_path, "--no-matfile")
    assert d.shape == (300, 5) and np.isfinite(d).all() and "Number of light passes 12." in out
    r = subprocess.run([CLI, str(tmp_path / "missing.ini")], capture_output=True, text=True)
    assert r.returncode == 1 and "Can't load" in r.stdout


def test_spectral_scene_with_a_generated_coefficient_table(tmp_path):
    """color_tables/srgb.coeff (absent from the reference tree) made by lib/rgb2spec_opt and picked up from the
    working directory exactly where the reference looks for it (Material.cpp:11)"""
    _write_scene(tmp_path, 300, 2, True)
    _, d0 = _run(tmp_path, "--no-matfile", "--passes", "4")
    (tmp_path / "color_tables").mkdir()
    tool = os.path.join(ROOT, "daisyriot_amd", "lib", "rgb2spec_opt")
    r = subprocess.run([tool, "16", str(tmp_path / "color_tables" / "srgb.coeff")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = tmp_path / "out.csv"
    r = subprocess.run([CLI, str(tmp_path / "config.ini"), "--out", str(out), "--no-matfile", "--passes", "4"],
                       capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert 'Loading "color_tables/srgb.coeff"' in r.stdout
    d1 = np.loadtxt(out, delimiter=",", skiprows=1)
    assert d1.shape == d0.shape and np.isfinite(d1).all() and (d1[:, 4:] >= 0).all()
    assert not np.allclose(d1[:, 4:], d0[:, 4:])          # the table's spectra, not the stand-in's
    # the upsampled reflectances stay below 1: four passes lose energy in every bin that carries any
    assert d1[:, 4:].sum() > 0


def test_cli_shards_over_several_devices_and_reads_the_extra_ini_keys(tmp_path):
    """[acceleration] devices = 0,0,0 : one process, three ranks (here all on GPU 0: the one-GPU rehearsal of the dr_group path,
    exchange by peer copies) -- same passes, same radiance as one device; rays_per_patch / seed / tolerance / bins are honoured"""
    _write_scene(tmp_path, 700, 1, True)
    out1, d1 = _run(tmp_path, "--no-matfile")
    ini = (tmp_path / "config.ini").read_text()
    (tmp_path / "config.ini").write_text(ini + "devices = 0,0,0\n")
    out3, d3 = _run(tmp_path, "--no-matfile", "--ply", str(tmp_path / "v.ply"))
    assert "sharded over 3 GPUs" in out3
    p1 = [l for l in out1.splitlines() if l.startswith("Number of light passes")][0].split(".")[0]
    assert p1 in out3                                                     # the same pass count
    assert np.allclose(d1, d3, rtol=2e-6, atol=1e-9)
    assert np.array_equal(d1[:, 1:4], d1[:, 4:7]) and np.array_equal(d3[:, 1:4], d3[:, 4:7])
    # --devices on the command line overrides the ini
    out2, d2 = _run(tmp_path, "--no-matfile", "--devices", "0,0")
    assert "sharded over 2 GPUs" in out2 and np.allclose(d1, d2, rtol=2e-6, atol=1e-9)
    # [acceleration] exchange / tree / walk name dr_options: the residual exchanged inside the pass, the SAH tree, the threaded walk
    (tmp_path / "config.ini").write_text(ini + "devices = 0,0,0\nexchange = inpass\ntree = sah\nwalk = threaded\n")
    out5, d5 = _run(tmp_path, "--no-matfile")
    assert "sharded over 3 GPUs" in out5 and p1 in out5 and np.array_equal(d3, d5)          # same bits as the peer-copy run
    (tmp_path / "config.ini").write_text(ini + "exchange = carrier-pigeon\n")
    r = subprocess.run([CLI, str(tmp_path / "config.ini"), "--no-matfile"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "unknown value 'carrier-pigeon'" in r.stderr
    # fewer rays, another seed: another (coarser) visibility estimate; a loose tolerance: fewer passes
    (tmp_path / "config.ini").write_text(ini + "rays_per_patch = 8\nseed = 7\n[lightning]\ntolerance = 0.05\n")
    out4, d4 = _run(tmp_path, "--no-matfile")
    n1 = int(p1.split()[-1])
    n4 = int([l for l in out4.splitlines() if l.startswith("Number of light passes")][0].split(".")[0].split()[-1])
    assert 0 < n4 < n1 and not np.allclose(d1[:, 4:], d4[:, 4:], rtol=1e-3)
    uv8 = np.zeros((8, 2), np.float32)
    C.CDLL(api.LIB_PATH, mode=C.RTLD_GLOBAL)
    C.CDLL(HOST).drh_visibility_samples(8, C.c_uint(7), uv8.ctypes.data_as(C.c_void_p))
    sc = scenes.cornell_box(700, S=3)
    with api.Context(0) as c:
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        c.assemble(uv8)
        c.solver_init(sc.emission(7.0), sc.M, sc.mat_of_patch)
        assert c.converge(0.05, per_bin=True, max_iters=1000) == n4
        B, _ = c.read()
    assert np.allclose(d4[:, 4:7], B, rtol=1e-6, atol=1e-9)


def test_spectral_bins_key_and_unreadable_cache(tmp_path):
    _write_scene(tmp_path, 300, 2, True)
    ini = (tmp_path / "config.ini").read_text()
    (tmp_path / "config.ini").write_text(ini.replace("method = 2", "method = 2\nbins = 5"))
    out, d = _run(tmp_path, "--no-matfile", "--passes", "2")
    assert d.shape == (300, 4 + 5) and np.isfinite(d).all()
    # a file at the cache path that is not this scene's matrix is neither trusted nor overwritten (the reference would
    # load whatever is there: Lightning.h:84-96)
    junk = np.arange(64, dtype=np.int32).tobytes()
    (tmp_path / "box").write_bytes(junk)
    r = subprocess.run([CLI, str(tmp_path / "config.ini"), "--out", str(tmp_path / "o.csv")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "not a form-factor cache of this scene" in r.stderr
    assert (tmp_path / "box").read_bytes() == junk
