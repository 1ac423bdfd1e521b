"""BASELINE.json configs 0 and 1: the reference's own example scenes (stored as arrays in
tests/golden/scene_*.npz by make_golden.py) through the HIP path, against the oracle on row
samples and against the statistics the survey measured with the reference's unmodified
triangle_math.cpp."""
import os

import numpy as np
import pytest

from daisyriot_amd import api, scenes
from oracle import binding as ob

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# (name, S, nonzero fraction, row-sum mean, row-sum max) of the unoccluded integrand, SURVEY.md 6
CASES = [("cornellbox_blacklight", 3, 0.193, 0.863, 1.52), ("colorballs", 8, 0.294, 0.691, 1.09)]


def _inputs(g, S):
    Kd, Ke = g["Kd"], g["Ke"]
    n_mat = Kd.shape[0]
    if S == 3:
        rho, emit = Kd.copy(), Ke.copy()
    else:
        rho = np.stack([scenes._smooth_spectrum(Kd[m], scenes.WAVELENGTHS_8) if len(set(Kd[m])) > 1
                        else np.full(S, min(Kd[m][0], 0.98), np.float32) for m in range(n_mat)])
        emit = np.stack([scenes._smooth_spectrum(Ke[m], scenes.WAVELENGTHS_8) if len(set(Ke[m])) > 1
                         else np.full(S, Ke[m][0], np.float32) for m in range(n_mat)])
    M = np.stack([np.diag(r) for r in rho]).astype(np.float32)
    E = (emit[g["mat"]] * np.float32(7.0)).astype(np.float32)
    return E, M


@pytest.mark.parametrize("name,S,nnz,rs_mean,rs_max", CASES)
def test_reference_scene(name, S, nnz, rs_mean, rs_max, uv50):
    g = np.load(os.path.join(GOLD, "scene_%s.npz" % name))
    N = g["tri_v"].shape[0]
    m = ob.Mesh(g["vertices"], g["normals"], g["tri_v"], g["tri_n"])
    E, M = _inputs(g, S)
    with api.Context(0) as c:
        c.set_mesh(g["vertices"], g["normals"], g["tri_v"], g["tri_n"])
        # the integrand alone: the numbers the reference's own CPU code gave in the survey probe
        c.integrand_only()
        Fu = c.read_rows(0, N)
        rs = Fu[:256].sum(axis=1, dtype=np.float64)          # the probe covered the first 256 rows
        assert abs((Fu[:256] > 0).mean() - nnz) < 0.0005
        assert abs(rs.mean() - rs_mean) < 0.0005 and abs(rs.max() - rs_max) < 0.005
        rows = np.linspace(0, N - 8, 6).astype(int)
        for r in rows:
            assert np.array_equal(Fu[r:r + 2].view(np.uint32), ob.integrand_rows(m, int(r), 2).view(np.uint32))
        # with visibility: row samples against the oracle (its own BVH), exact
        c.assemble(uv50, keep_visibility=True)
        F = c.read_rows(0, N)
        for r in rows:
            Fo, viso, _ = ob.assemble_rows(m, uv50, row0=int(r), nrows=2, bvh=True)
            assert np.array_equal(c.read_visibility(int(r), 2), viso), (name, r)
            assert np.array_equal(F[r:r + 2].view(np.uint32), Fo.view(np.uint32)), (name, r)
        assert np.isfinite(F).all() and (F >= 0).all() and np.all(np.diag(F) == 0)
        assert (F <= Fu).all()                                   # visibility only ever removes energy
        # light passes on the scene's own materials, against the oracle on the same F
        c.solver_init(E, M, g["mat"])
        c.step(6)
        B, R = c.read()
    Ro, Bo = E.copy(), E.copy()
    for _ in range(6):
        Ro = ob.sweep_rows(F, M, g["mat"], Ro, Bo)
    for got, want in ((R, Ro), (B, Bo)):
        assert (np.abs(got - want) / (np.abs(want).max(axis=0) + 1e-30)).max() < 2e-5


# north_star's parity sentence is about CONVERGED radiance: the reference's own stopping rules on its own scenes.
#   RGB      vs/Lightning.h:336-340   go on while any channel's sum of the residual exceeds 1e-4
#   spectral vs/Lightning.h:145-151   go on while the sum over all bins exceeds a threshold (200 in the reference; with
#                                     colorballs' lamps at emission_value 7 the sums start at 1.3e5 and the reference's
#                                     200 is reached after a handful of passes, so the test also runs 1e-3 of the
#                                     emitted total: same rule, more than 10 passes)
# cornellbox_blacklight in RGB mode: MeshS classifies its only emitter as a UV lamp (vs/MeshS.cpp:41-63), whose RGB
# emission is zero -- the reference's literal inputs give E = 0 and zero passes (checked: same here).  To have light in
# the scene the second case keeps the lamp's `Ke 1 1 1` of the .mtl file as RGB emission; reflectances are the file's Kd.
def _case_inputs(name, variant):
    g = np.load(os.path.join(GOLD, "scene_%s.npz" % name))
    if name == "colorballs":
        E, M = _inputs(g, 8)
        return g, E, M, False, (200.0 if variant == "ref200" else float(E.sum(dtype=np.float64)) * 1e-3)
    Kd, Ke = g["Kd"].copy(), g["Ke"].copy()
    if variant == "lamp_lit":
        Ke[g["kind"] == 1] = 1.0
    M = np.stack([np.diag(r) for r in Kd]).astype(np.float32)
    E = (Ke[g["mat"]] * np.float32(7.0)).astype(np.float32)
    return g, E, M, True, 1e-4


@pytest.mark.parametrize("name,variant,min_passes", [("cornellbox_blacklight", "literal", 0), ("cornellbox_blacklight", "lamp_lit", 11),
                                                     ("colorballs", "ref200", 1), ("colorballs", "total_1e-3", 11)])
def test_converged_radiance_on_the_reference_scenes(name, variant, min_passes, uv50):
    g, E, M, per_bin, thr = _case_inputs(name, variant)
    N = g["tri_v"].shape[0]
    with api.Context(0) as c:
        c.set_mesh(g["vertices"], g["normals"], g["tri_v"], g["tri_n"])
        c.assemble(uv50)
        F = c.read_rows(0, N)
        c.solver_init(E, M, g["mat"])
        it = c.converge(thr, per_bin=per_bin, max_iters=5000)
        B, R = c.read()
        sums = c.residual_sums()
    it_o, Ro, Bo = ob.converge(F, M, g["mat"], E, thr, per_bin, 5000)
    assert min_passes <= it < 5000, it
    assert it == it_o, (it, it_o)
    # the rule really stopped it: the residual's sums are at or below the threshold now
    assert (sums <= thr).all() if per_bin else sums.sum() <= thr
    # converged radiance per bin within north_star's 1e-4 (relative to the bin's maximum)
    assert (np.abs(B - Bo) / (np.abs(Bo).max(axis=0) + 1e-30)).max() < 1e-4
    assert (np.abs(R - Ro).max(axis=0) <= 1e-4 * np.abs(Bo).max(axis=0) + 1e-12).all()
