"""the generator of color_tables/srgb.coeff (daisyriot_amd/host/rgb2spec_opt.cpp) and the table
reader/evaluator of the host (rgb2spec_fetch / rgb2spec_eval_precise semantics, vs/rgb2spec.cpp:78-134).
The reference's own srgb.coeff is absent from its tree (.MISSING_LARGE_BLOBS), so the numbers cannot be
compared with it -- parity unpinned; pinned here: the file format the reference's loader reads, and the property
the table exists for (spectra in [0,1] whose colour under D65 through the reference's observer fit is the RGB)."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "daisyriot_amd", "lib", "rgb2spec_opt")
HOST = os.path.join(ROOT, "daisyriot_amd", "lib", "libdaisyriot_host.so")
RES = 16

D65 = np.array([0.0341, 3.2945, 20.236, 37.0535, 39.9488, 44.9117, 46.6383, 52.0891, 49.9755, 54.6482, 82.7549, 91.486,
                93.4318, 86.6823, 104.865, 117.008, 117.812, 114.861, 115.923, 108.811, 109.354, 107.802, 104.79, 107.689,
                104.405, 104.046, 100.0, 96.3342, 95.788, 88.6856, 90.0062, 89.5991, 87.6987, 83.2886, 83.6992, 80.0268,
                80.2146, 82.2778, 78.2842, 69.7213, 71.6091, 74.349, 61.604, 69.8856, 75.087, 63.5927, 46.4182, 66.8054,
                63.3828, 64.304, 59.4519, 51.959, 57.4406, 60.3125])      # CIE D65, 300..830 nm, 10 nm
XYZ_TO_RGB = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
RGB_TO_XYZ = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])


def _lobe(x, mu, a, b):
    t = (x - mu) * np.where(x < mu, a, b)
    return np.exp(-0.5 * t * t)


WL = np.arange(360.0, 831.0, 1.0)
OBS = np.stack([0.362 * _lobe(WL, 442.0, 0.0624, 0.0374) + 1.056 * _lobe(WL, 599.8, 0.0264, 0.0323) - 0.065 * _lobe(WL, 501.1, 0.0490, 0.0382),
                0.821 * _lobe(WL, 568.8, 0.0213, 0.0247) + 0.286 * _lobe(WL, 530.9, 0.0613, 0.0322),
                1.217 * _lobe(WL, 437.0, 0.0845, 0.0278) + 0.681 * _lobe(WL, 459.0, 0.0385, 0.0725)])     # vs/color.h:14-45
ILL = np.interp(WL, np.arange(300.0, 831.0, 10.0), D65)
GAIN = RGB_TO_XYZ.sum(axis=1) / (OBS * ILL).sum(axis=1)      # white reflector under D65 = RGB (1,1,1)


def colour_of(spectrum):
    return XYZ_TO_RGB @ ((OBS * ILL * spectrum).sum(axis=1) * GAIN)


@pytest.fixture(scope="module")
def table(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("ct") / "srgb.coeff")
    r = subprocess.run([TOOL, str(RES), path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return path


@pytest.fixture(scope="module")
def up():
    C.CDLL(os.path.join(ROOT, "daisyriot_amd", "lib", "libdaisyriot_hip.so"), mode=C.RTLD_GLOBAL)
    L = C.CDLL(HOST)
    L.drh_upsample.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]

    def f(path, rgb, wl=WL):
        rgb = np.ascontiguousarray(rgb, np.float32)
        wl = np.ascontiguousarray(wl, np.float32)
        out = np.zeros(wl.size, np.float32)
        found = L.drh_upsample(path.encode(), rgb.ctypes.data_as(C.c_void_p), wl.ctypes.data_as(C.c_void_p), wl.size,
                               out.ctypes.data_as(C.c_void_p))
        return found, out
    return f


def test_file_format_is_what_the_reference_loader_reads(table):
    raw = open(table, "rb").read()
    assert raw[:4] == b"SPEC"                                                   # rgb2spec.cpp:16-20
    res = struct.unpack("<I", raw[4:8])[0]
    assert res == RES
    assert len(raw) == 8 + 4 * res + 4 * 3 * res ** 3 * 3                       # scale[res] + data[3][res]^3[3], rgb2spec.cpp:30-33
    scale = np.frombuffer(raw[8:8 + 4 * res], np.float32)
    k = np.arange(res) / (res - 1)
    s1 = k * k * (3 - 2 * k)
    assert np.allclose(scale, s1 * s1 * (3 - 2 * s1), atol=1e-7) and scale[0] == 0 and scale[-1] == 1
    data = np.frombuffer(raw[8 + 4 * res:], np.float32)
    assert np.isfinite(data).all()


def test_grid_colours_round_trip(table, up):
    scale = np.frombuffer(open(table, "rb").read()[8:8 + 4 * RES], np.float32)
    rs = np.random.RandomState(5)
    worst = 0.0
    for _ in range(150):
        l, k, j, i = rs.randint(0, 3), rs.randint(3, RES), rs.randint(0, RES), rs.randint(0, RES)
        rgb = np.zeros(3)
        rgb[l], rgb[(l + 1) % 3], rgb[(l + 2) % 3] = scale[k], scale[k] * i / (RES - 1), scale[k] * j / (RES - 1)
        found, sp = up(table, rgb)
        assert found == 1 and sp.min() >= 0.0 and sp.max() <= 1.0
        worst = max(worst, np.abs(colour_of(sp) - rgb).max())
    assert worst < 2e-3, worst


def test_greys_are_flat_and_white_is_one(table, up):
    for g in (0.18, 0.5, 0.75, 1.0):
        _, sp = up(table, [g, g, g])
        assert sp.max() - sp.min() < 1e-3, g                   # flat ...
        assert abs(sp.mean() - g) < 1e-2, g                    # ... at the grey's level (coarse test table: 1 %)
    _, sp = up(table, [0, 0, 0])                       # black: no lookup (the reference's divides by zero), zero spectrum
    assert np.all(sp == 0)


def test_interpolated_colours_are_close_and_bins_outside_the_visible_are_bounded(table, up):
    # off-grid colours at this coarse resolution: interpolation of the coefficients, a few percent
    for rgb in ((0.75, 0.75, 0.75), (0.14, 0.45, 0.091), (0.2, 0.3, 0.9), (0.63, 0.3, 0.25)):
        _, sp = up(table, rgb)
        assert np.abs(colour_of(sp) - np.array(rgb)).max() < 0.05, rgb
    # the reference evaluates the model at its bins 200..600 nm (main.cpp:94): the sigmoid keeps them in [0,1]
    _, sp = up(table, (0.63, 0.065, 0.05), wl=np.arange(200.0, 601.0, 50.0))
    assert sp.shape == (9,) and sp.min() >= 0 and sp.max() <= 1 and sp[-1] > sp[4]      # a red: more at 600 than at 400 nm


def test_missing_table_falls_back_to_the_stand_in(up, tmp_path):
    found, sp = up(str(tmp_path / "nothing.coeff"), (0.63, 0.065, 0.05))
    assert found == 0 and sp.min() >= 0 and sp.max() <= 1
