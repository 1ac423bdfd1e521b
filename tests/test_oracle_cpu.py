"""CPU tests of the oracle (the checker): the reference's one known answer, the committed
golden vectors (reference's vendored glm / Eigen arithmetic, tests/golden/make_golden.py),
internal consistency (literal form == per-patch-record form, BVH == brute force) and the
analytic identities SURVEY.md 8(c) lists."""
import os

import numpy as np
import pytest

from daisyriot_amd import scenes
from oracle import binding as ob

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _mesh(sc):
    return ob.Mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)


def test_reference_known_answer_unit_triangle_area():
    # vs/unittest1.cpp:15 -- the only known-answer test in the reference
    assert ob.surface([0, 0, 0], [1, 0, 0], [0, 1, 0]) == 0.5


def test_golden_integrand_through_glm():
    g = np.load(os.path.join(GOLD, "integrand_glm.npz"))
    assert float(g["unit_triangle_area"]) == 0.5
    m = ob.Mesh(g["vertices"], g["normals"], g["tri_v"], g["tri_n"])
    got = np.array([ob.p2p_integrand_literal(m, i, j) for i, j in g["pairs"]], np.float32)
    want = g["integrand_bits"].view(np.float32)
    assert np.all(want[:8] == 0)                           # i == j: normalize(0) is NaN, the > 0 tests fail -> 0
    assert not np.isnan(want).any()
    assert np.array_equal(_bits(got), _bits(want))
    rows = ob.integrand_rows(m)
    stored = np.where(want > 0, want, 0).astype(np.float32)
    assert np.array_equal(_bits(rows[g["pairs"][:, 0], g["pairs"][:, 1]]), _bits(stored))
    pts = np.array([ob.uv2xyz(m, t, g["uv"][k % 50, 0], g["uv"][k % 50, 1]) for k, t in enumerate(g["uv_tri"])])
    assert np.array_equal(_bits(pts), g["uv_points_bits"])


@pytest.mark.parametrize("name", ["spectral9", "rgb3", "bw1"])
def test_golden_light_passes_through_eigen(name):
    g = np.load(os.path.join(GOLD, "lightpass_%s_eigen.npz" % name))
    F, M, E, mat = g["F"], g["M"], g["E"], g["mat"]
    R, B = E.copy(), E.copy()
    done = 0
    for upto in (1, 2, 5, 20):
        for _ in range(upto - done):
            R = ob.sweep_rows(F, M, mat, R, B)
        done = upto
        assert np.array_equal(_bits(R), g["R%d" % upto]), (name, upto)
        assert np.array_equal(_bits(B), g["B%d" % upto]), (name, upto)
    # Eigen's vectorised .sum() orders additions differently: equal to rounding only
    assert np.allclose(ob.residual_sums(R), g["eigen_sums_R20"], rtol=1e-5, atol=1e-12)


def test_golden_display_colours_through_color_h():
    # vs/color.h:14-52 + vs/Lightning.h:168-183 + vs/Drawer.cpp:161-186, evaluated by the reference's own header
    g = np.load(os.path.join(GOLD, "display_color_h.npz"))
    xyz = ob.xyz_fit(g["wavelengths"])
    assert np.array_equal(_bits(xyz), g["xyz_bits"])
    rgb = ob.patch_colors(g["B"], 2, xyz)
    assert np.array_equal(_bits(rgb), g["rgb_bits"])
    mx = rgb.max(axis=1)
    assert (mx < 1).any() and (mx == 1).any()              # both branches of the max-normalisation
    N = g["tri_v"].shape[0]
    vtx = ob.vertex_colors(g["vtx_off"], g["vtx_tri"], rgb[N:2 * N])
    assert np.array_equal(_bits(vtx), g["vertex_rgb_bits"])
    # RGB / BW pass the radiosity through (Lightning.h:332-334, 406-408)
    B3 = np.ascontiguousarray(g["B"][:, :3])
    assert np.array_equal(ob.patch_colors(B3, 1), B3)
    B1 = np.ascontiguousarray(g["B"][:, :1])
    assert np.array_equal(ob.patch_colors(B1, 0), np.repeat(B1, 3, axis=1))


@pytest.mark.skipif(not ob.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_against_vendored_glm_and_eigen_live():
    sc = scenes.cornell_box(150, S=8, fluorescent=True)
    m = _mesh(sc)
    rs = np.random.RandomState(3)
    rows = ob.integrand_rows(m)
    for i, j in rs.randint(0, sc.N, size=(800, 2)):
        r = ob.ref_p2p_integrand(m, i, j)
        r = r if r > 0 else np.float32(0)
        assert _bits(r) == _bits(rows[i, j])
    F, _, _ = ob.assemble_rows(m, scenes.visibility_samples(50), bvh=True, want_vis=False)
    E = sc.emission(7.0)
    R, B = E.copy(), E.copy()
    Rr, Br = E.copy(), E.copy()
    for _ in range(4):
        R = ob.sweep_rows(F, sc.M, sc.mat_of_patch, R, B)
        Rr, Br = ob.ref_light_pass(F, sc.M, sc.mat_of_patch, Rr, Br, 0)
    assert np.array_equal(_bits(R), _bits(Rr)) and np.array_equal(_bits(B), _bits(Br))


def test_literal_form_equals_record_form():
    sc = scenes.cornell_box(120, S=1)
    m = _mesh(sc)
    rows = ob.integrand_rows(m)
    rs = np.random.RandomState(1)
    for i, j in rs.randint(0, sc.N, size=(500, 2)):
        f = ob.p2p_integrand_literal(m, i, j)
        f = f if f > 0 else np.float32(0)
        assert _bits(f) == _bits(rows[i, j])
    assert np.all(np.diag(rows) == 0)


def test_bvh_visibility_equals_brute_force(uv50):
    for sc in (scenes.cornell_box(260, S=1), scenes.facing_squares(cells=3, occluder=True)):
        m = _mesh(sc)
        F1, v1, _ = ob.assemble_rows(m, uv50, bvh=False)
        F2, v2, _ = ob.assemble_rows(m, uv50, bvh=True)
        assert np.array_equal(v1, v2) and np.array_equal(_bits(F1), _bits(F2))
        assert np.array_equal(v1, v1.T)                         # one shared count per unordered pair


def test_visibility_semantics(uv50):
    sc = scenes.facing_squares(cells=1, gap=1.0, S=1, occluder=False)
    m = _mesh(sc)
    # open pair: every ray reaches its target; closest hit of a hand-made ray
    assert ob.visibility_count(m, 0, 2, uv50) == 50
    tid, t = ob.closest_hit(m, [0.25, 0.0, 0.25], [0, 1, 0])
    assert tid in (2, 3) and abs(t - 1.0) < 1e-6
    assert ob.closest_hit(m, [0.25, 0.5, 0.25], [1, 0, 0])[0] == -1
    # with the occluder some pairs are fully blocked, and F stores nothing for them
    sc = scenes.facing_squares(cells=2, gap=1.0, S=1, occluder=True)
    F, vis, _ = ob.assemble_rows(_mesh(sc), uv50)
    traced = vis != 255
    assert (vis[traced] == 0).any()
    assert np.all(F[traced & (vis == 0)] == 0)
    assert np.all(F[~traced] == 0)


def test_rules_differ_only_in_the_reverse_entry(uv50):
    sc = scenes.cornell_box(200, S=1)
    m = _mesh(sc)
    Fi, _, _ = ob.assemble_rows(m, uv50, rule=ob.RULE_INTEGRAND, bvh=True)
    Fr, _, _ = ob.assemble_rows(m, uv50, rule=ob.RULE_RECIPROCITY, bvh=True)
    iu = np.triu_indices(sc.N, 1)
    assert np.array_equal(_bits(Fi[iu]), _bits(Fr[iu]))           # F[row<col] identical
    _, _, _, area = ob.patch_records(m)
    lhs = area[:, None] * Fr                                      # reciprocity A_i F_ij = A_j F_ji
    assert np.allclose(lhs, lhs.T, rtol=2e-6, atol=1e-12)
    nz = (Fi > 0) & (Fi.T > 0)
    assert np.allclose((area[:, None] * Fi)[nz], (area[:, None] * Fi).T[nz], rtol=0.35)   # quadrature-level only


def test_analytic_identities(uv50):
    # two parallel unit squares one unit apart: F = 0.19982 (closed form)
    sc = scenes.facing_squares(cells=6, gap=1.0, S=1)
    m = _mesh(sc)
    F, _, _ = ob.assemble_rows(m, uv50, bvh=True)
    _, _, _, area = ob.patch_records(m)
    n1 = sc.N // 2
    F12 = (area[:n1, None] * F[:n1, n1:]).sum() / area[:n1].sum()
    assert abs(F12 - 0.19982) < 0.004
    assert np.all(F[:n1, :n1] == 0)                               # coplanar patches exchange nothing
    # closed convex box: every patch sees exactly the rest of the box
    sc = scenes.closed_box(cells=4, S=1)
    F, vis, _ = ob.assemble_rows(_mesh(sc), uv50, bvh=True)
    assert abs(F.sum(axis=1).mean() - 1.0) < 0.06             # 4x4-point rule, 192 patches: 1.039
    assert np.all(vis[vis != 255] == 50)


def test_solver_semantics():
    rs = np.random.RandomState(2)
    N, S = 40, 3
    F = (rs.random_sample((N, N)) / N).astype(np.float32)
    np.fill_diagonal(F, 0)
    rho = rs.random_sample((2, S)).astype(np.float32) * 0.8
    M = np.stack([np.diag(r) for r in rho]).astype(np.float32)
    mat = rs.randint(0, 2, N).astype(np.int32)
    E = np.zeros((N, S), np.float32)
    E[:3] = 9.0
    # RGB rule (vs/Lightning.h:342-349): residual = (F*residual) .* rho
    R, B = E.copy(), E.copy()
    R1 = ob.sweep_rows(F, M, mat, R, B)
    want = np.zeros_like(R1)
    for s in range(S):
        y = np.zeros(N, np.float32)
        for j in range(N):
            y = (y + F[:, j] * E[j, s]).astype(np.float32)
        want[:, s] = y * rho[mat, s]
    assert np.array_equal(_bits(R1), _bits(want))
    assert np.array_equal(_bits(B), _bits(E + R1))
    # converge_lightning: a scene whose residual starts below the threshold takes zero passes
    # (vs/Lightning.h:145-151; SURVEY 8c: 63 < 200)
    it, Rc, Bc = ob.converge(F, M, mat, E, 200.0, False, 100)
    assert it == 0 and np.array_equal(Bc, E)
    it, Rc, Bc = ob.converge(F, M, mat, E, 1e-4, True, 1000)
    assert 3 < it < 1000 and ob.residual_sums(Rc).max() <= 1e-4
    assert ob.converge(F, M, mat, E, 0.0, False, 5)[0] == 5           # cap


def test_sample_set_and_scenes():
    uv = scenes.visibility_samples(50)
    assert uv.shape == (50, 2) and (uv >= 0).all() and (uv.sum(axis=1) <= 1.0 + 1e-6).all()
    assert np.array_equal(uv, scenes.visibility_samples(50))        # deterministic
    for n in (64, 1000, 16384):
        sc = scenes.cornell_box(n, S=8)
        assert sc.N == n and sc.tri_v.max() < sc.vertices.shape[0] and sc.tri_n.max() < sc.normals.shape[0]
        _, sa, nrm, area = ob.patch_records(_mesh(sc))
        assert (area > 0).all() and np.allclose(np.linalg.norm(nrm, axis=1), 1, atol=1e-6)
        assert np.allclose(sa.sum(axis=1), area, rtol=1e-5)
    sc = scenes.cornell_box(640, S=9, fluorescent=True)
    f = sc.material_names.index("fluorescent")
    assert np.allclose(np.diag(sc.M[f])[[0, 1, 2, 4]], 1.0)          # unit diagonal outside the UV bins
    assert sc.emit[sc.material_names.index("uvlamp")].argmax() == 3   # 350 nm


@pytest.mark.parametrize("name,nnz,rs_mean,rs_max", [("cornellbox_blacklight", 0.193, 0.863, 1.52),
                                                    ("colorballs", 0.294, 0.691, 1.09)])
def test_integrand_statistics_of_the_reference_run(name, nnz, rs_mean, rs_max):
    """the reference's unmodified triangle_math.cpp, driven over the first 256
    rows of its own example scenes, gave these nonzero fractions and row sums; the oracle must too."""
    g = np.load(os.path.join(GOLD, "scene_%s.npz" % name))
    m = ob.Mesh(g["vertices"], g["normals"], g["tri_v"], g["tri_n"])
    Fu = ob.integrand_rows(m, 0, 256)
    rs = Fu.sum(axis=1, dtype=np.float64)
    assert abs((Fu > 0).mean() - nnz) < 0.0005
    assert abs(rs.mean() - rs_mean) < 0.0005 and abs(rs.max() - rs_max) < 0.005


@pytest.mark.parametrize("name", ["cornellbox_blacklight", "colorballs"])
def test_cuda_twin_arithmetic_deviation(name):
    """The build evaluates the integrand as the reference's CPU file does (triangle_math.cpp:49-58, all float, M_PIf) under
    BOTH assembly rules; with cuda_on = true the reference itself runs the CUDA twin, which divides by a double pi and
    multiplies in double (parallellism.cu:197-207).  How far apart are the two?  Measured on the reference's own scenes:
    a few float ulps per form factor, 1e-6 of the converged radiance -- two orders below north_star's 1e-4 bar."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "scene_%s.npz" % name))
    m = ob.Mesh(g["vertices"], g["normals"], g["tri_v"], g["tri_n"])
    N = m.N
    Fa = ob.integrand_rows(m)
    Fb = ob.integrand_rows(m, cuda_twin=True)
    assert np.array_equal(Fa > 0, Fb > 0)                       # the same pairs see each other
    nz = Fa > 0
    rel = np.abs(Fa[nz].astype(np.float64) - Fb[nz]) / Fa[nz]
    assert rel.max() < 1.5e-6 and rel.mean() < 2e-7             # 16 terms of <= 1.5 ulp each, summed in float
    # converged radiance (RGB rule, unoccluded F scaled to a contraction so that the iteration converges like the real one)
    Kd, Ke, mat = g["Kd"], g["Ke"], g["mat"]
    M = np.zeros((Kd.shape[0], 3, 3), np.float32)
    for k in range(Kd.shape[0]):
        M[k] = np.diag(np.clip(Kd[k], 0, 0.8))
    E = np.where(Ke[mat] > 0, Ke[mat] * np.float32(7.0), 0).astype(np.float32)
    if not (E > 0).any():
        E[: max(1, N // 200)] = 7.0
    scale = np.float32(0.9 / max(Fa.sum(axis=1).max(), 1e-9))
    out = []
    for F in (Fa, Fb):
        Fs = (F * scale).astype(np.float32)
        it, R, B = ob.converge(Fs, M, mat, E, 1e-4, True, 400)
        out.append((it, B))
    assert out[0][0] == out[1][0]
    dB = np.abs(out[0][1] - out[1][1]) / (np.abs(out[0][1]).max(axis=0) + 1e-30)
    assert dB.max() < 2e-6
    print("%s: max rel dF %.2e, mean %.2e, max rel dB %.2e after %d passes" % (name, rel.max(), rel.mean(), dB.max(), out[0][0]))
