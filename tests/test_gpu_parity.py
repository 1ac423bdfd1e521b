"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on
the same inputs.  Bars: visibility ray counts and the traced-pair set are integers ->
exact; the stored integrand is computed with the same individually rounded fp32
operations as the oracle -> bit-exact; F = V*Fu -> bit-exact; light passes use fused
multiply-adds in a different summation order -> relative 2e-5 of the per-bin maximum
(north_star's bar is 1e-4 relative on converged radiance)."""
import os

import numpy as np
import pytest

from daisyriot_amd import api, scenes
from oracle import binding as ob

pytestmark = pytest.mark.gpu

SWEEP_RTOL = 2e-5


def _ctx(sc):
    c = api.Context(0)
    c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
    return c


def _mesh(sc):
    return ob.Mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("n", [64, 200, 1000])
def test_integrand_bit_exact(n):
    sc = scenes.cornell_box(n, S=3)
    with _ctx(sc) as c:
        c.integrand_only()
        F = c.read_rows(0, sc.N)
    ref = ob.integrand_rows(_mesh(sc))
    assert np.array_equal(_bits(F), _bits(ref))
    assert np.all(np.diag(F) == 0)


@pytest.mark.parametrize("rule", [api.RULE_INTEGRAND, api.RULE_RECIPROCITY])
@pytest.mark.parametrize("n", [64, 333, 1000])
def test_formfactors_and_visibility_exact(n, rule, uv50):
    sc = scenes.cornell_box(n, S=3)
    with _ctx(sc) as c:
        c.assemble(uv50, rule=rule, keep_visibility=True)
        F = c.read_rows(0, sc.N)
        vis = c.read_visibility(0, sc.N)
        info = c.info()
    Fo, viso, _ = ob.assemble_rows(_mesh(sc), uv50, rule=rule, bvh=True)
    assert np.array_equal(vis, viso), "ray counts differ in %d entries" % int((vis != viso).sum())
    assert np.array_equal(_bits(F), _bits(Fo))
    assert info.pairs_traced == int((viso != 255).sum()) // 2


def test_visibility_against_brute_force_closest_hit(uv50):
    """the oracle's brute-force closest hit (no BVH on either side of the comparison)"""
    sc = scenes.cornell_box(300, S=1)
    with _ctx(sc) as c:
        c.assemble(uv50, keep_visibility=True)
        vis = c.read_visibility(0, sc.N)
    _, viso, _ = ob.assemble_rows(_mesh(sc), uv50, bvh=False)
    assert np.array_equal(vis, viso)


def test_occluder_blocks_and_K_other_than_50():
    sc = scenes.facing_squares(cells=4, gap=1.0, S=1, occluder=True)
    for K in (1, 7, 64, 65, 130):
        uv = scenes.visibility_samples(K, seed=3)
        with _ctx(sc) as c:
            c.assemble(uv, keep_visibility=True)
            vis = c.read_visibility(0, sc.N)
            F = c.read_rows(0, sc.N)
        Fo, viso, _ = ob.assemble_rows(_mesh(sc), uv, bvh=False)
        assert np.array_equal(vis, viso), K
        assert np.array_equal(_bits(F), _bits(Fo)), K
    traced = viso != 255
    assert (viso[traced] == 0).any() and (viso[traced] == 130).any()   # some pairs blocked, some open


def test_ragged_sizes_and_single_triangle(uv50):
    for n in (65, 127, 129, 257):
        sc = scenes.cornell_box(n, S=1)
        with _ctx(sc) as c:
            c.assemble(uv50, keep_visibility=True)
            F = c.read_rows(0, sc.N)
        Fo, _, _ = ob.assemble_rows(_mesh(sc), uv50, bvh=True)
        assert np.array_equal(_bits(F), _bits(Fo)), n
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    nn = np.array([[0, 0, 1]], np.float32)
    with api.Context(0) as c:
        c.set_mesh(v, nn, [[0, 1, 2]], [[0, 0, 0]])
        c.assemble(uv50)
        assert c.read_rows(0, 1)[0, 0] == 0.0


def test_degenerate_triangle_is_harmless(uv50):
    sc = scenes.closed_box(cells=2, S=1)
    tv = sc.tri_v.copy()
    tv[5] = [tv[5][0], tv[5][0], tv[5][1]]        # zero-area patch: area 0 -> NaN integrand -> stored 0
    with api.Context(0) as c:
        c.set_mesh(sc.vertices, sc.normals, tv, sc.tri_n)
        c.assemble(uv50, keep_visibility=True)
        F = c.read_rows(0, sc.N)
    Fo, _, _ = ob.assemble_rows(ob.Mesh(sc.vertices, sc.normals, tv, sc.tri_n), uv50, bvh=False)
    assert np.isfinite(F).all()
    assert np.array_equal(_bits(F), _bits(Fo))


def test_errors_are_reported():
    sc = scenes.closed_box(cells=1, S=1)
    with api.Context(0) as c:
        with pytest.raises(api.DaisyRiotError):
            c.assemble(scenes.visibility_samples(5))           # no mesh yet
        bad = sc.tri_v.copy()
        bad[0, 0] = 10 ** 6
        with pytest.raises(api.DaisyRiotError):
            c.set_mesh(sc.vertices, sc.normals, bad, sc.tri_n)
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        with pytest.raises(api.DaisyRiotError):
            c.step(1)                                          # solver not initialised
        with pytest.raises(api.DaisyRiotError):
            c.solver_init(np.zeros((sc.N, 17), np.float32), np.zeros((1, 17, 17), np.float32), np.zeros(sc.N, np.int32))
    with pytest.raises(api.DaisyRiotError):
        api.Context(10 ** 6)


@pytest.mark.parametrize("S,fluor,n", [(1, False, 300), (3, False, 520), (8, True, 700), (9, True, 300), (16, False, 260)])
def test_light_passes_match_oracle(S, fluor, n, uv50):
    sc = scenes.cornell_box(n, S=S, fluorescent=fluor)
    E = sc.emission(7.0)
    with _ctx(sc) as c:
        c.assemble(uv50)
        F = c.read_rows(0, sc.N)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        R = E.copy()
        B = E.copy()
        for passes in (1, 1, 3, 15):
            tot = c.step(passes, want_sum=True)
            for _ in range(passes):
                R = ob.sweep_rows(F, sc.M, sc.mat_of_patch, R, B)
            Bg, Rg = c.read()
            for got, want in ((Rg, R), (Bg, B)):
                scale = np.abs(want).max(axis=0) + 1e-30
                assert (np.abs(got - want) / scale).max() < SWEEP_RTOL
            assert abs(tot - ob.residual_sums(R).sum()) <= 1e-5 * max(1.0, abs(tot))
        c.reset()
        Bg, Rg = c.read()
        assert np.array_equal(Bg, E) and np.array_equal(Rg, E)


def test_converge_stops_at_the_same_pass(uv50):
    sc = scenes.cornell_box(400, S=3)
    E = sc.emission(7.0)
    with _ctx(sc) as c:
        c.assemble(uv50)
        F = c.read_rows(0, sc.N)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        it_rgb = c.converge(1e-4, per_bin=True, max_iters=500)          # Lightning.h:336-340
        Bg, _ = c.read()
        it_o, Ro, Bo = ob.converge(F, sc.M, sc.mat_of_patch, E, 1e-4, True, 500)
        assert it_rgb == it_o and it_rgb > 5
        assert (np.abs(Bg - Bo) / (np.abs(Bo).max(axis=0) + 1e-30)).max() < 1e-4
        c.reset()
        it_cap = c.converge(0.0, per_bin=False, max_iters=7)            # cap honoured
        assert it_cap == 7
        c.reset()
        assert c.converge(200.0, per_bin=False, max_iters=50) == ob.converge(F, sc.M, sc.mat_of_patch, E, 200.0, False, 50)[0]
        # the convergence test runs on the device; the host looks once per batch of queued passes: whatever the batch,
        # the same pass count and the same state (passes queued behind the converged one do nothing)
        for every in (1, 3, 64):
            c.set_check_interval(every)
            c.reset()
            assert c.converge(1e-4, per_bin=True, max_iters=500) == it_rgb
            Bk, Rk = c.read()
            assert np.array_equal(_bits(Bk), _bits(Bg))
            assert abs(c.residual_sums().sum() - Rk.astype(np.float64).sum()) <= 1e-9 * max(1.0, Rk.sum())
            c.step(1)                                                    # and the state is consistent: stepping on works
            c.reset()
            assert c.converge(0.0, per_bin=False, max_iters=7) == 7
        c.set_check_interval(8)


def test_sweep_on_loaded_rows_and_padding():
    """externally supplied F (the DeserializeMat route) with N not a multiple of the tile"""
    rs = np.random.RandomState(5)
    sc = scenes.cornell_box(300, S=8)
    N = sc.N
    F = (rs.random_sample((N, N)) * (rs.random_sample((N, N)) < 0.3) / N).astype(np.float32)
    E = rs.random_sample((N, 8)).astype(np.float32)
    with _ctx(sc) as c:
        c.load_rows(0, F)
        assert np.array_equal(c.read_rows(0, N), F)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        c.step(4)
        Bg, Rg = c.read()
    R, B = E.copy(), E.copy()
    for _ in range(4):
        R = ob.sweep_rows(F, sc.M, sc.mat_of_patch, R, B)
    assert (np.abs(Rg - R) / np.abs(R).max(axis=0)).max() < SWEEP_RTOL
    assert (np.abs(Bg - B) / np.abs(B).max(axis=0)).max() < SWEEP_RTOL


def test_sharded_rows_assemble_identically(uv50):
    """a rank of a 2- or 3-way shard produces exactly its rows of the single-GPU matrix
    (the multi-rank exchange itself is covered by the CPU gloo tests)"""
    sc = scenes.cornell_box(700, S=3)
    with _ctx(sc) as c:
        c.assemble(uv50)
        F = c.read_rows(0, sc.N)
    for world in (2, 3):
        for rank in range(world):
            with api.Context(0) as c:
                c.set_shard(rank, world)
                c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
                row0, nrows, rpr = c.shard()
                c.assemble(uv50)
                if nrows:
                    assert np.array_equal(_bits(c.read_rows(row0, nrows)), _bits(F[row0:row0 + nrows]))


def test_full_size_properties(uv50):
    """BASELINE.json config 3 size (16k patches): size-independent properties instead of the oracle."""
    sc = scenes.cornell_box(16384, S=8)
    E = sc.emission(7.0)
    with _ctx(sc) as c:
        c.assemble(uv50, keep_visibility=True)
        info = c.info()
        rows = np.r_[0:64, 8000:8064, 16320:16384]
        F = np.concatenate([c.read_rows(r, 64) for r in (0, 8000, 16320)])
        vis = np.concatenate([c.read_visibility(r, 64) for r in (0, 8000, 16320)])
        # spot rows against the oracle (BVH visibility)
        Fo, viso, _ = ob.assemble_rows(_mesh(sc), uv50, row0=8000, nrows=8, bvh=True)
        assert np.array_equal(vis[64:72], viso)
        assert np.array_equal(_bits(F[64:72]), _bits(Fo))
        assert np.isfinite(F).all() and (F >= 0).all()
        assert np.all(F[np.arange(rows.size), rows] == 0)              # F_ii = 0
        # closed room: row sums of a closed environment stay near 1 (4x4-point rule: loose band)
        rs_ = F.sum(axis=1)
        assert 0.5 < rs_.mean() < 1.3
        # linearity of a pass: step(E1+E2) = step(E1)+step(E2)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        c.step(3)
        B1, R1 = c.read()
        c.solver_init(2 * E, sc.M, sc.mat_of_patch)
        c.step(3)
        B2, R2 = c.read()
        assert np.allclose(R2, 2 * R1, rtol=1e-5, atol=1e-9)
        # energy decays: the residual sum shrinks every pass in a scene with rho < 1
        c.reset()
        sums = [c.step(1, want_sum=True) for _ in range(6)]
        assert all(b < a for a, b in zip(sums, sums[1:]))
    assert info.pairs_traced > 0


def test_bvh_invariants():
    """threaded LBVH: pre-order skips, every patch in exactly one leaf, leaves inside their boxes"""
    sc = scenes.cornell_box(3000, S=1)
    with _ctx(sc) as c:
        b = c.read_bvh()
    n = len(b)
    assert np.all(b["skip"] > np.arange(n)) and np.all(b["skip"] <= n) and b["skip"][0] == n
    leaf = b["tri"] >= 0
    first, cnt = b["tri"][leaf] >> 3, (b["tri"][leaf] & 3) + 1          # (bit 2: the leaf's two gate boxes are equal)
    assert cnt.sum() == sc.N and cnt.max() <= 4
    order = np.argsort(first)
    assert np.array_equal(first[order], np.r_[0, np.cumsum(cnt[order])[:-1]])     # leaves tile the sorted array
    assert np.all(b["skip"][leaf] == np.nonzero(leaf)[0] + 1)                     # a leaf's subtree is itself
    internal = np.nonzero(~leaf)[0]
    for i in internal[:: max(1, len(internal) // 300)]:                           # children inside the parent
        l, r = i + 1, b["skip"][i + 1]
        for ch in (l, r):
            # (boxes are kept as centre/half-extent, each grown by the same pad: equal up to rounding of the centre)
            assert np.all(b["lo"][ch] >= b["lo"][i] - 1e-6) and np.all(b["hi"][ch] <= b["hi"][i] + 1e-6)
        assert b["skip"][r] == b["skip"][i]
    lo, hi = sc.vertices.min(0), sc.vertices.max(0)
    assert np.all(b["lo"][0] <= lo) and np.all(b["hi"][0] >= hi)


@pytest.mark.parametrize("fenced", [0, 1])
def test_column_split_sweep_matches_oracle(fenced):
    """small row shards cut the columns into ranges (k_sweep<..., SPLIT>: the last range of a row block adds the partial sums);
    forced here through the context's options (dr_options::sweep_ksplit) -- with the hand-offs between the blocks of a pass in
    their shipped form (write-through stores + ticket) and in the memory-model form (release / acquire fences): same bits"""
    rs = np.random.RandomState(11)
    for S in (3, 8, 12):
        sc = scenes.cornell_box(1000, S=S)
        N = sc.N
        F = (rs.random_sample((N, N)) * (rs.random_sample((N, N)) < 0.4) / N).astype(np.float32)
        E = rs.random_sample((N, S)).astype(np.float32)
        got = {}
        for fz in sorted({0, fenced}):
            with api.Context(0) as c:
                o = c.set_options(sweep_ksplit=4, sweep_fenced=fz)
                assert o.sweep_ksplit == 4 and c.options().sweep_fenced == fz
                c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
                c.load_rows(0, F)
                c.solver_init(E, sc.M, sc.mat_of_patch)
                assert c.info().sweep_ksplit == 4
                c.step(3)
                got[fz] = c.read()
        Bg, Rg = got[fenced]
        assert np.array_equal(_bits(got[0][0]), _bits(Bg)) and np.array_equal(_bits(got[0][1]), _bits(Rg))
        R, B = E.copy(), E.copy()
        for _ in range(3):
            R = ob.sweep_rows(F, sc.M, sc.mat_of_patch, R, B)
        assert (np.abs(Rg - R) / np.abs(R).max(axis=0)).max() < 2e-5, S
        assert (np.abs(Bg - B) / np.abs(B).max(axis=0)).max() < 2e-5, S


def test_options_round_trip_and_validation():
    """dr_options: what is set is what is read back; nonsense is refused; the pass layout cannot change under a live solver"""
    d = api.options_defaults()
    with api.Context(0) as c:
        assert c.options().as_dict() == d.as_dict()
        o = c.set_options(tree=api.TREE_LBVH, walk=api.WALK_THREADED, sah_bins=16, sweep_taper=2)
        assert (o.tree, o.walk, o.sah_bins, o.sweep_taper) == (api.TREE_LBVH, api.WALK_THREADED, 16, 2)
        assert c.options().as_dict() == o.as_dict()
        for bad in (dict(tree=7), dict(walk=-1), dict(sah_bins=1), dict(sweep_ksplit=65), dict(sweep_rows_per_wave=3),
                    dict(group_exchange=9), dict(sah_dilate=float("nan"))):
            with pytest.raises(api.DaisyRiotError):
                c.set_options(**bad)
        assert c.options().as_dict() == o.as_dict()             # a refused set changes nothing
        sc = scenes.cornell_box(300, S=3)
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        i = c.info()
        assert i.tree_used == api.TREE_LBVH and i.tree_depth > 3
        c.integrand_only()
        c.solver_init(sc.emission(7.0), sc.M, sc.mat_of_patch)
        with pytest.raises(api.DaisyRiotError, match="cannot change under an initialised solver"):
            c.set_options(sweep_ksplit=2)
        c.set_options(sweep_taper=0)                             # (not a layout option: allowed)


def test_rccl_binding_single_rank(uv50):
    """world = 1 through the real exchange path: RCCL found at run time, unique id, communicator,
    in-place all-gather after every pass -- results unchanged"""
    sc = scenes.cornell_box(300, S=8)
    E = sc.emission(7.0)
    with _ctx(sc) as c:
        c.assemble(uv50)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        c.step(4)
        B0, R0 = c.read()
    with api.Context(0) as c:
        c.set_shard(0, 1)
        c.comm_init(api.comm_unique_id(), 0, 1)
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        c.assemble(uv50)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        c.step(4)
        B1, R1 = c.read()
    assert np.array_equal(B0, B1) and np.array_equal(R0, R1)


@pytest.mark.parametrize("world,S,n", [(2, 8, 700), (3, 8, 700), (3, 9, 700), (4, 3, 2500), (5, 16, 1300)])
def test_row_sharded_solver_on_one_gpu(world, S, n, uv50):
    """every rank's kernels (rank > 0: row offset, short last shard, chunk placement, local B and
    material indices; the two launches of a multi-rank pass -- own chunk, then the other columns -- on the
    VALU and the MFMA kernel) with the exchange staged through the host: identical to the unsharded run"""
    sc = scenes.cornell_box(n, S=S, fluorescent=(S >= 8))
    E = sc.emission(7.0)
    passes = 5
    with _ctx(sc) as c:
        c.assemble(uv50)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        c.step(passes)
        B1, R1 = c.read()
    ranks = []
    for r in range(world):
        c = api.Context(0)
        c.set_shard(r, world)
        c.comm_manual()
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        c.assemble(uv50)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        ranks.append(c)
    for _ in range(passes):
        for c in ranks:
            c.step(1)
        chunks = [c.exchange_export() for c in ranks]
        for d, c in enumerate(ranks):
            for s_, ch in enumerate(chunks):
                if s_ != d:
                    c.exchange_import(s_, ch)
    B = np.zeros_like(B1)
    for c in ranks:
        row0, nrows, _ = c.shard()
        Bc, Rc = c.read()
        B[row0:row0 + nrows] = Bc[row0:row0 + nrows]
        # same kernel, same per-row arithmetic -> the gathered residual is bitwise the unsharded one
        # unless the column split differs between shard sizes (then equal to rounding)
        assert np.allclose(Rc, R1, rtol=2e-6, atol=1e-12)
        c.close()
    assert np.allclose(B, B1, rtol=2e-6, atol=1e-12)


def test_context_reuse_and_limits(uv50):
    """one context through several scenes / rules / bin counts; argument limits"""
    with api.Context(0) as c:
        for n, S in ((130, 3), (64, 8), (257, 1)):
            sc = scenes.cornell_box(n, S=S)
            m = _mesh(sc)
            c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
            for rule in (api.RULE_RECIPROCITY, api.RULE_INTEGRAND):
                c.assemble(uv50, rule=rule, keep_visibility=(rule == api.RULE_INTEGRAND))
                Fo, _, _ = ob.assemble_rows(m, uv50, rule=rule, bvh=True, want_vis=False)
                assert np.array_equal(_bits(c.read_rows(0, sc.N)), _bits(Fo))
            E = sc.emission(3.0)
            for _ in range(2):
                c.solver_init(E, sc.M, sc.mat_of_patch)
                c.step(2)
            B, R = c.read()
            Ro, Bo = E.copy(), E.copy()
            for _ in range(2):
                Ro = ob.sweep_rows(Fo, sc.M, sc.mat_of_patch, Ro, Bo)
            assert np.allclose(B, Bo, rtol=2e-5, atol=1e-9)
            with pytest.raises(api.DaisyRiotError):
                c.read_rows(sc.N - 1, 2)                           # past the last row
            with pytest.raises(api.DaisyRiotError):
                c.solver_init(E, sc.M, np.full(sc.N, 99, np.int32))  # material index out of range
        # K limits: counts are bytes with 255 as the "not traced" mark
        sc = scenes.facing_squares(cells=1, S=1)
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        uv = scenes.visibility_samples(254, seed=9)
        c.assemble(uv, keep_visibility=True)
        vis = c.read_visibility(0, sc.N)
        _, viso, _ = ob.assemble_rows(_mesh(sc), uv, bvh=False)
        assert np.array_equal(vis, viso) and vis[vis != 255].max() == 254
        with pytest.raises(api.DaisyRiotError):
            c.assemble(scenes.visibility_samples(255))
        bad = sc.vertices.copy()
        bad[0, 0] = np.nan
        with pytest.raises(api.DaisyRiotError):
            c.set_mesh(bad, sc.normals, sc.tri_v, sc.tri_n)
        with pytest.raises(api.DaisyRiotError):
            c.set_mesh(np.zeros((0, 3), np.float32), sc.normals, np.zeros((0, 3), np.int32), np.zeros((0, 3), np.int32))
    # two triangles only
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1]], np.float32)
    nn = np.array([[0, 0, 1], [0, 0, -1]], np.float32)
    tv, tn = np.array([[0, 1, 2], [3, 5, 4]], np.int32), np.array([[0, 0, 0], [1, 1, 1]], np.int32)
    with api.Context(0) as c:
        c.set_mesh(v, nn, tv, tn)
        c.assemble(uv50, keep_visibility=True)
        F, vis = c.read_rows(0, 2), c.read_visibility(0, 2)
    Fo, viso, _ = ob.assemble_rows(ob.Mesh(v, nn, tv, tn), uv50, bvh=False)
    assert np.array_equal(_bits(F), _bits(Fo)) and np.array_equal(vis, viso) and F[0, 1] > 0


def _bowl_scene():
    """two facing sheets; the lower one is a very shallow bowl with per-triangle geometric normals, so
    its own patches face each other at grazing angles; everything sits far from the origin"""
    sc = scenes.facing_squares(cells=6, gap=1.0, S=1)
    v = sc.vertices.astype(np.float64)
    lower = np.isclose(v[:, 1], 0.0)
    v[lower, 1] += 0.01 * ((v[lower, 0] - 0.5) ** 2 + (v[lower, 2] - 0.5) ** 2)
    v += 3.7
    v = v.astype(np.float32)
    a, b, c = (v[sc.tri_v[:, k]].astype(np.float64) for k in range(3))
    n = np.cross(b - a, c - a)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    tn = np.repeat(np.arange(sc.N, dtype=np.int32)[:, None], 3, axis=1)
    return v, n.astype(np.float32), sc.tri_v, tn, sc.N


def test_grazing_rays_along_a_shallow_bowl(uv50):
    """nearly coplanar patches that do face each other (as on the slightly tilted walls of the
    reference's colorballs scene): rays run almost parallel to thin BVH boxes far from the origin --
    the slab test must stay conservative (an fma-folded slab test loses the whole t range here)"""
    v, nrm, tv, tn, N = _bowl_scene()
    with api.Context(0) as c:
        c.set_mesh(v, nrm, tv, tn)
        c.assemble(uv50, keep_visibility=True)
        vis, F = c.read_visibility(0, N), c.read_rows(0, N)
    Fo, viso, _ = ob.assemble_rows(ob.Mesh(v, nrm, tv, tn), uv50, bvh=False)
    n1 = N // 2
    sub = viso[:n1, :n1]
    assert (sub != 255).sum() > 100 and (sub[sub != 255] < 50).any()   # in-sheet pairs are traced, some blocked
    assert np.array_equal(vis, viso)
    assert np.array_equal(_bits(F), _bits(Fo))


@pytest.mark.parametrize("seed,n,scale,offset", [(1, 300, 1.0, 0.0), (2, 500, 40.0, 0.0), (3, 200, 1e-3, 0.0),
                                                 (4, 400, 1.0, 700.0), (5, 300, 1.0, -3.0e4), (6, 1500, 5.0, 0.0)])
def test_random_triangle_soup(seed, n, scale, offset, uv50):
    """general position: intersecting, overlapping, sliver and tiny triangles with arbitrary vertex
    normals, at three scene scales, far from the origin (coordinates much larger than the scene), and a
    denser one -- ray counts and F against the brute-force oracle, exact"""
    rs = np.random.RandomState(seed)
    c0 = rs.random_sample((n, 1, 3)) * 2 - 1
    size = rs.random_sample((n, 1, 1)) ** 3 * 0.6 + 1e-3
    tri = (c0 + (rs.random_sample((n, 3, 3)) - 0.5) * size) * scale + 0.37 * scale + offset * np.array([1.0, -0.4, 0.25])
    tri[::17, 2] = tri[::17, 1] + (tri[::17, 1] - tri[::17, 0]) * 1e-4       # slivers
    v = tri.reshape(-1, 3).astype(np.float32)
    tv = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    nrm = rs.normal(size=(64, 3)).astype(np.float32)
    tn = rs.randint(0, 64, size=(n, 3)).astype(np.int32)
    with api.Context(0) as c:
        c.set_mesh(v, nrm, tv, tn)
        c.assemble(uv50, keep_visibility=True)
        vis, F = c.read_visibility(0, n), c.read_rows(0, n)
    Fo, viso, _ = ob.assemble_rows(ob.Mesh(v, nrm, tv, tn), uv50, bvh=False)
    traced = viso != 255
    assert traced.sum() > n and (viso[traced] == 0).any() and (viso[traced] == 50).any()
    assert np.array_equal(vis, viso), int((vis != viso).sum())
    assert np.array_equal(_bits(F), _bits(Fo))


def test_bench_size_64k_properties(uv50):
    """BASELINE.json's metric size (65 536 patches, 8 bins) on one GPU: spot rows against the oracle
    (exact) and the size-independent properties of the matrix and of the passes"""
    sc = scenes.cornell_box(65536, S=8)
    N = sc.N
    E = sc.emission(7.0)
    m = _mesh(sc)
    with _ctx(sc) as c:
        c.assemble(uv50, keep_visibility=True)
        info = c.info()
        assert info.bytes_F == 4 * N * N
        for r in (0, 31111, N - 2):
            Fo, viso, _ = ob.assemble_rows(m, uv50, row0=r, nrows=2, bvh=True)
            assert np.array_equal(c.read_visibility(r, 2), viso), r
            assert np.array_equal(_bits(c.read_rows(r, 2)), _bits(Fo)), r
        # transpose symmetry of the ray counts on a block far from the diagonal
        va, vb = c.read_visibility(1000, 64)[:, 40000:40064], c.read_visibility(40000, 64)[:, 1000:1064]
        assert np.array_equal(va, vb.T)
        rows = c.read_rows(20000, 256)
        assert np.isfinite(rows).all() and (rows >= 0).all()
        assert np.all(rows[np.arange(256), 20000 + np.arange(256)] == 0)
        assert 0.9 < rows.sum(axis=1).mean() < 1.1                      # closed room
        c.solver_init(E, sc.M, sc.mat_of_patch)
        sums = [c.step(1, want_sum=True) for _ in range(5)]
        assert all(b < a for a, b in zip(sums, sums[1:]))               # energy decays, rho < 1
        B, R = c.read()
        c.solver_init(3 * E, sc.M, sc.mat_of_patch)
        c.step(5)
        B3, R3 = c.read()
        assert np.allclose(R3, 3 * R, rtol=1e-5, atol=1e-12) and np.allclose(B3, 3 * B, rtol=1e-5, atol=1e-12)
        # one pass against the oracle on a row sample of the same F
        c.solver_init(E, sc.M, sc.mat_of_patch)
        c.step(1)
        _, R1 = c.read()
        Fs = c.read_rows(50000, 64)
        Bo = E[50000:50064].copy()
        Ro = ob.sweep_rows(Fs, sc.M, sc.mat_of_patch, E, Bo, row0=50000)
        assert (np.abs(R1[50000:50064] - Ro) / (np.abs(Ro).max(axis=0) + 1e-30)).max() < 2e-5


def test_one_card_holds_196k_patches(uv50):
    """A scene three times the bench's on ONE card: 196 608 patches, dense F = 154.6 GB of the 288 (+ 38.7 GB of ray counts kept for
    the test).  Spot rows against the oracle (exact), transpose symmetry of the ray counts far from the diagonal, closed-room row
    sums, and one light pass of sampled rows against the oracle."""
    sc = scenes.cornell_box(196608, S=8)
    N = sc.N
    assert N == 196608
    E = sc.emission(7.0)
    m = _mesh(sc)
    with _ctx(sc) as c:
        c.assemble(uv50, keep_visibility=True)
        info = c.info()
        assert info.bytes_F == 4 * N * N
        for r in (1, 98765, N - 2):
            Fo, viso, _ = ob.assemble_rows(m, uv50, row0=r, nrows=2, bvh=True)
            assert np.array_equal(c.read_visibility(r, 2), viso), r
            assert np.array_equal(_bits(c.read_rows(r, 2)), _bits(Fo)), r
        va, vb = c.read_visibility(3000, 64)[:, 150016:150080], c.read_visibility(150016, 64)[:, 3000:3064]
        assert np.array_equal(va, vb.T)
        rows = c.read_rows(60000, 128)
        assert np.isfinite(rows).all() and (rows >= 0).all() and 0.9 < rows.sum(axis=1).mean() < 1.1
        c.solver_init(E, sc.M, sc.mat_of_patch)
        c.step(1)
        _, R1 = c.read()
        Fs = c.read_rows(170000, 64)
        Bo = E[170000:170064].copy()
        Ro = ob.sweep_rows(Fs, sc.M, sc.mat_of_patch, E, Bo, row0=170000)
        assert (np.abs(R1[170000:170064] - Ro) / (np.abs(Ro).max(axis=0) + 1e-30)).max() < 2e-5


def test_c5_shard_256k_fluorescent(uv50):
    """BASELINE.json's largest configuration on one card: 262 144 patches, 8 bins with fluorescent cross-bin
    transfer, rank 5 of an 8-way row shard (32 768 x 262 144 floats = 34.4 GB of F).  Spot rows against the
    oracle (exact), matrix properties, one light pass of sampled rows against the oracle."""
    sc = scenes.cornell_box(262144, S=8, fluorescent=True)
    N = sc.N
    assert N == 262144 and np.abs(sc.M - sc.M * np.eye(8, dtype=np.float32)).max() > 0      # a cross-bin column exists
    E = sc.emission(7.0)
    m = _mesh(sc)
    rank, world = 5, 8
    with api.Context(0) as c:
        c.set_shard(rank, world)
        c.comm_manual()
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        row0, nrows, rpr = c.shard()
        assert (row0, nrows, rpr) == (5 * 32768, 32768, 32768)
        c.assemble(uv50, keep_visibility=True)
        info = c.info()
        assert info.bytes_F == 4 * nrows * N
        r = row0 + nrows - 4                                 # the last rows of the shard (one oracle call: ~40 s of CPU)
        Fo, viso, _ = ob.assemble_rows(m, uv50, row0=r, nrows=4, bvh=True)
        assert np.array_equal(c.read_visibility(r, 4), viso)
        assert np.array_equal(_bits(c.read_rows(r, 4)), _bits(Fo))
        rows = c.read_rows(row0 + 9000, 128)
        assert np.isfinite(rows).all() and (rows >= 0).all()
        assert np.all(rows[np.arange(128), row0 + 9000 + np.arange(128)] == 0)
        assert 0.9 < rows.sum(axis=1).mean() < 1.1
        # the block on the diagonal of this shard is symmetric in its ray counts
        va = c.read_visibility(row0 + 100, 64)[:, row0 + 20000:row0 + 20064]
        vb = c.read_visibility(row0 + 20000, 64)[:, row0 + 100:row0 + 164]
        assert np.array_equal(va, vb.T)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        c.step(1)
        B1, _ = c.read(R=False)
        r = row0 + 30000
        Fs = c.read_rows(r, 48)
        Bo = E[r:r + 48].copy()
        Ro = ob.sweep_rows(Fs, sc.M, sc.mat_of_patch, E, Bo, row0=r)
        assert (np.abs(B1[r:r + 48] - Bo) / (np.abs(Bo).max(axis=0) + 1e-30)).max() < 2e-5
        own = c.exchange_export()[:8 * rpr].reshape(8, rpr)           # this rank's new residual chunk, bin-major (+ its sums)
        assert (np.abs(own[:, 30000:30048].T - Ro) / (np.abs(Ro).max(axis=0) + 1e-30)).max() < 2e-5


@pytest.mark.parametrize("world,n,rule", [(2, 700, api.RULE_INTEGRAND), (3, 1000, api.RULE_RECIPROCITY), (5, 1500, api.RULE_INTEGRAND)])
def test_multi_rank_assembly_with_ray_count_exchange(world, n, rule, uv50):
    """a pair between two ranks' rows is traced by one of them only and its ray counts handed over (what
    dr_formfactors_assemble does over RCCL, here with the slot buffers staged through the host): every rank ends
    with exactly its rows of the single-GPU matrix, and together the ranks trace every pair once"""
    sc = scenes.cornell_box(n, S=3)
    with _ctx(sc) as c:
        c.assemble(uv50, rule=rule, keep_visibility=True)
        F, V = c.read_rows(0, sc.N), c.read_visibility(0, sc.N)
        traced_once = c.info().pairs_traced
    ranks = []
    for r in range(world):
        c = api.Context(0)
        c.set_shard(r, world)
        c.comm_manual()
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        c.assemble_split(uv50, rule=rule, keep_visibility=True)
        ranks.append(c)
    total = sum(c.info().pairs_traced for c in ranks)
    assert total == traced_once                                    # nothing traced twice, nothing left out
    # the all-to-all: the block rank a traced for rank b goes to b (and only to b)
    blocks = {(a, b): ranks[a].vis_exchange_export(b) for a in range(world) for b in range(world) if a != b}
    for d, c in enumerate(ranks):
        for s_ in range(world):
            if s_ != d:
                c.vis_exchange_import(s_, blocks[(s_, d)])
        c.assemble_finish()
    for c in ranks:
        row0, nrows, _ = c.shard()
        if nrows:
            assert np.array_equal(c.read_visibility(row0, nrows), V[row0:row0 + nrows])
            assert np.array_equal(_bits(c.read_rows(row0, nrows)), _bits(F[row0:row0 + nrows]))
        c.close()
    # a rank on its own (no exchange) still gets its rows: every pair that touches them is traced locally
    with api.Context(0) as c:
        c.set_shard(world - 1, world)
        c.comm_manual()
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        c.assemble(uv50, rule=rule)
        row0, nrows, _ = c.shard()
        if nrows:
            assert np.array_equal(_bits(c.read_rows(row0, nrows)), _bits(F[row0:row0 + nrows]))
        with pytest.raises(api.DaisyRiotError):
            c.assemble_finish()                                    # nothing pending
        with pytest.raises(api.DaisyRiotError):
            c.vis_exchange_export(0)                               # no split assembly in flight


@pytest.mark.parametrize("S,n,world,rank", [(8, 8192, 1, 0), (3, 6000, 1, 0), (9, 8192, 1, 0), (16, 6000, 1, 0),
                                             (8, 16384, 1, 0), (8, 8192, 2, 1), (12, 8192, 3, 2)])
def test_zero_block_skipping_is_bit_identical(S, n, world, rank, uv50):
    """optional dr_solver_skip_zero_blocks: passes that do not read the all-zero 32 x 256 blocks of F give the same
    bits (the skipped products are exact zeros), on the VALU and MFMA kernels, fused and column-split, on a shard"""
    sc = scenes.cornell_box(n, S=S, fluorescent=(S >= 8))
    E = sc.emission(7.0)
    with api.Context(0) as c:
        if world > 1:
            c.set_shard(rank, world)
            c.comm_manual()
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        c.assemble(uv50)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        c.step(3)
        B0, R0 = c.read()
        own0 = c.exchange_export() if world > 1 else None
        assert c.info().blocks_total == 0                                  # not built unless asked for
        c.skip_zero_blocks(True)
        c.reset()
        c.step(3)
        B1, R1 = c.read()
        i = c.info()
        assert 0 < i.blocks_nonzero < i.blocks_total, (i.blocks_nonzero, i.blocks_total)     # something is skipped
        row0, nrows, _ = c.shard()
        assert np.array_equal(_bits(B1[row0:row0 + nrows]), _bits(B0[row0:row0 + nrows]))
        if world == 1:
            assert np.array_equal(_bits(R1), _bits(R0))
        else:
            assert np.array_equal(_bits(c.exchange_export()), _bits(own0))
        # a new matrix invalidates the block map: loaded rows that fill a formerly empty block must be seen
        rows = c.read_rows(row0, 64)
        rows[:, :] = 1e-3
        c.load_rows(row0, rows)
        c.reset()
        c.step(1)
        Bs, _ = c.read(R=False)
        c.skip_zero_blocks(False)
        c.reset()
        c.step(1)
        Bd, _ = c.read(R=False)
        assert np.array_equal(_bits(Bs[row0:row0 + nrows]), _bits(Bd[row0:row0 + nrows]))


def test_work_on_the_callers_stream(uv50):
    """dr_set_stream: launches and copies go to a stream the host owns (here a torch stream), so that the host can
    order its own work with ours; results as on the context's own stream; back to the own stream with NULL"""
    import torch
    sc = scenes.cornell_box(900, S=8, fluorescent=True)
    E = sc.emission(7.0)
    with _ctx(sc) as c:
        c.assemble(uv50)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        c.step(3)
        B0, R0 = c.read()
    st = torch.cuda.Stream()
    with api.Context(0) as c:
        c.set_stream(st.cuda_stream)
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        c.assemble(uv50)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        c.step(2)
        done = torch.cuda.Event()
        done.record(st)                      # the host's own marker behind our passes on its stream
        done.synchronize()
        c.set_stream(0)                      # NULL: back to the context's stream (drains the other one first)
        c.step(1)
        B1, R1 = c.read()
    assert np.array_equal(_bits(B1), _bits(B0)) and np.array_equal(_bits(R1), _bits(R0))


@pytest.mark.parametrize("exchange", ["p2p", "inpass"])
@pytest.mark.parametrize("world,S,n", [(2, 8, 700), (3, 3, 2500), (4, 9, 1300), (3, 3, 400)])
def test_group_one_process_several_ranks(world, S, n, exchange, uv50):
    """dr_group: one process, `world` ranks (here all on GPU 0: the one-GPU rehearsal) -- assembly with ray-count exchange,
    passes and converge give the single-context results.  Exchange of the residual by peer copies behind events (p2p), or
    IN the pass (dr_options::group_exchange = INPASS): the pass kernel stores its chunk and its sums into every rank's buffer
    and publishes its number, the next pass starts behind a one-thread gate kernel -- no copy, no collective, no event; same
    bits either way.  (3, 3, 400) leaves the last rank without rows."""
    sc = scenes.cornell_box(n, S=S, fluorescent=(S >= 8))
    E = sc.emission(7.0)
    with _ctx(sc) as c:
        c.assemble(uv50, keep_visibility=True)
        F, V = c.read_rows(0, sc.N), c.read_visibility(0, sc.N)
        traced_once = c.info().pairs_traced
        c.solver_init(E, sc.M, sc.mat_of_patch)
        c.step(4)
        B1, R1 = c.read()
        c.reset()
        thr, per_bin = (1e-3, True) if S == 3 else (0.5, False)
        it1 = c.converge(thr, per_bin=per_bin, max_iters=300)
        Bc1, _ = c.read()
    with api.Group([0] * world) as g:
        assert not g.uses_rccl()
        if exchange == "inpass":
            assert g.set_options(group_exchange=api.GROUP_EXCHANGE_INPASS).group_exchange == api.GROUP_EXCHANGE_INPASS
        g.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        g.assemble(uv50, keep_visibility=True)
        assert sum(c.info().pairs_traced for c in g.ranks) == traced_once
        for c in g.ranks:
            row0, nrows, _ = c.shard()
            if nrows:
                assert np.array_equal(c.read_visibility(row0, nrows), V[row0:row0 + nrows])
                assert np.array_equal(_bits(c.read_rows(row0, nrows)), _bits(F[row0:row0 + nrows]))
        g.solver_init(E, sc.M, sc.mat_of_patch)
        tot = g.step(4, want_sum=True)
        B, R = g.read()
        assert np.allclose(B, B1, rtol=2e-6, atol=1e-12) and np.allclose(R, R1, rtol=2e-6, atol=1e-12)
        assert abs(tot - R1.astype(np.float64).sum()) <= 1e-5 * max(1.0, abs(tot))
        for c in g.ranks:                                     # every rank holds the same gathered residual
            _, Rc = c.read(B=False)
            assert np.array_equal(_bits(Rc), _bits(R))
        g.reset()
        assert g.converge(thr, per_bin=per_bin, max_iters=300) == it1
        Bc, Rc = g.read()
        assert np.allclose(Bc, Bc1, rtol=2e-6, atol=1e-12)
        KEEP[(world, S, n, exchange)] = (_bits(B).copy(), _bits(R).copy(), _bits(Bc).copy(), _bits(Rc).copy())
    other = KEEP.get((world, S, n, "p2p" if exchange == "inpass" else "inpass"))
    if other is not None:                                     # the two exchanges: bit-identical B and R
        assert all(np.array_equal(a, b) for a, b in zip(other, KEEP[(world, S, n, exchange)]))


KEEP = {}


@pytest.mark.parametrize("on_host", [0, 1])
def test_sah_tree_on_degenerate_layouts(uv50, on_host):
    """The SAH build (on the device, and the host's reference builder) on inputs that defeat binning: triangles strung along a line at exponentially growing distances (every
    cut is lopsided) and a pile of coincident triangles (all centroids in one bin) -- the median fallback keeps the tree
    logarithmic, and the ray counts and F still equal the brute-force oracle's."""
    rs = np.random.RandomState(11)
    base = (rs.random_sample((1, 3, 3)) - 0.5).astype(np.float64)
    line = np.concatenate([base + np.array([[[1.35 ** k, 0.0, 0.3 * (k % 3)]]]) for k in range(60)])
    pile = np.repeat(base + np.array([[[2.0, 1.0, 0.0]]]), 90, axis=0)
    facing = (rs.random_sample((100, 3, 3)) - 0.5) * 0.8 + np.array([[[3.0, 0.5, 2.0]]])
    tri = np.concatenate([line, pile, facing])
    n = tri.shape[0]
    v = tri.reshape(-1, 3).astype(np.float32)
    tv = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    nrm = rs.normal(size=(8, 3)).astype(np.float32)
    tn = rs.randint(0, 8, size=(n, 3)).astype(np.int32)
    with api.Context(0) as c:
        c.set_options(tree=api.TREE_SAH, sah_on_host=on_host)
        c.set_mesh(v, nrm, tv, tn)
        assert c.info().tree_depth < 40                     # logarithmic whatever the layout
        c.assemble(uv50, keep_visibility=True)
        vis, F = c.read_visibility(0, n), c.read_rows(0, n)
    Fo, viso, _ = ob.assemble_rows(ob.Mesh(v, nrm, tv, tn), uv50, bvh=False)
    assert np.array_equal(vis, viso) and np.array_equal(_bits(F), _bits(Fo))


def test_device_sah_tree_is_the_hosts_tree():
    """the SAH topology built on the device (level by level, one workgroup per open node) against the host's reference builder
    on the same scene: the same algorithm in the same float expressions, so -- up to the order inside a leaf pair and tie
    cases -- the same tree: equal node counts and depths within one, and a summed node area (what a walk pays for) within 0.5 %"""
    for n in (7000, 16384):
        sc = scenes.cornell_box(n, S=3)
        got = {}
        for on_host in (0, 1):
            with api.Context(0) as c:
                c.set_options(tree=api.TREE_SAH, sah_on_host=on_host)
                c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
                i = c.info()
                assert i.tree_used == api.TREE_SAH and i.tree_on_host == on_host
                b = c.read_bvh()
                ext = np.maximum(b["hi"].astype(np.float64) - b["lo"], 0.0)
                area = ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]
                # every patch exactly once in the leaves (leaf code = first * 8 + count - 1 [+ 4])
                leaves = b["tri"][b["tri"] >= 0]
                cover = np.zeros(n, np.int32)
                for code in leaves:
                    cover[(code >> 3):(code >> 3) + (code & 3) + 1] += 1
                assert (cover == 1).all()
                got[on_host] = (len(b), i.tree_depth, area.sum(), i.last_bvh_ms)
        assert abs(got[0][0] - got[1][0]) <= 0.002 * got[1][0], got
        assert abs(got[0][1] - got[1][1]) <= 1, got
        assert abs(got[0][2] - got[1][2]) <= 0.005 * got[1][2], got


def test_all_trees_give_the_same_matrix_at_16k(uv50):
    """The tree only decides how many nodes a walk visits, never what is hit: the Morton tree, the SAH topology built on the
    device (the default at this size) and the one from the host's reference builder give the same ray counts and the same F,
    bit for bit, on every row of a 16 384-patch scene (hashes of all rows)."""
    import hashlib
    sc = scenes.cornell_box(16384, S=8)
    got = {}
    for tree, opts in (("lbvh", dict(tree=api.TREE_LBVH)), ("sah", dict(tree=api.TREE_SAH)), ("sah-host", dict(tree=api.TREE_SAH, sah_on_host=1))):
        with api.Context(0) as c:
            c.set_options(**opts)
            c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
            c.assemble(uv50, keep_visibility=True)
            hF, hV = hashlib.sha256(), hashlib.sha256()
            for r0 in range(0, 16384, 2048):
                hF.update(_bits(c.read_rows(r0, 2048)).tobytes())
                hV.update(c.read_visibility(r0, 2048).tobytes())
            got[tree] = (hF.hexdigest(), hV.hexdigest(), c.info().pairs_traced)
    assert got["lbvh"] == got["sah"] == got["sah-host"]


ALT_MODES = {
    "threaded-walk": dict(walk=api.WALK_THREADED),
    "pair-walk": dict(walk=api.WALK_PAIRS),
    "path-records": dict(walk=api.WALK_PATHS),
    "general-node-test": dict(octant_test=0),
    "general-node-test+threaded": dict(octant_test=0, walk=api.WALK_THREADED),
    "sah-tree": dict(tree=api.TREE_SAH),
    "sah-tree-from-host": dict(tree=api.TREE_SAH, sah_on_host=1),
    "morton-tree": dict(tree=api.TREE_LBVH),
    "sah-tree+path-records": dict(tree=api.TREE_SAH, walk=api.WALK_PATHS),
    "morton-tree+threaded": dict(tree=api.TREE_LBVH, walk=api.WALK_THREADED),
}


@pytest.mark.parametrize("mode", sorted(ALT_MODES))
def test_alternative_walks_are_exact(mode, uv50):
    """Every tree and every walk the options offer (dr_options: walk = threaded tree | sibling-pair records | path records in
    front of the pair walk; octant_test = 0: the general node test for every pair; tree = Morton on the device | binned SAH
    on the device | binned SAH from the host's reference builder) gives the same ray counts and F as the brute-force
    oracle: on a soup far from the origin, on the Cornell box, and with an origin offset that throws rays far from their
    patches.  The context reports what it really used."""
    opts = ALT_MODES[mode]
    if True:
        rs = np.random.RandomState(5)
        n = 300
        c0 = rs.random_sample((n, 1, 3)) * 2 - 1
        tri = (c0 + (rs.random_sample((n, 3, 3)) - 0.5) * 0.4) * 3.0 + 250.0
        v = tri.reshape(-1, 3).astype(np.float32)
        tv = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
        nrm = rs.normal(size=(16, 3)).astype(np.float32)
        tn = rs.randint(0, 16, size=(n, 3)).astype(np.int32)
        for eps in (api.ORIGIN_EPS, 0.3):
            with api.Context(0) as c:
                c.set_options(**opts)
                c.set_mesh(v, nrm, tv, tn)
                c.assemble(uv50, eps=eps, keep_visibility=True)
                vis, F = c.read_visibility(0, n), c.read_rows(0, n)
                i = c.info()
            if "walk" in opts:
                assert i.walk_used == opts["walk"]
            if "tree" in opts:
                assert i.tree_used == opts["tree"] and i.tree_on_host == opts.get("sah_on_host", 0)
            Fo, viso, _ = ob.assemble_rows(ob.Mesh(v, nrm, tv, tn), uv50, eps=eps, bvh=False)
            assert np.array_equal(vis, viso) and np.array_equal(_bits(F), _bits(Fo))
        sc = scenes.cornell_box(1500, S=3)
        with api.Context(0) as c:
            c.set_options(**opts)
            c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
            c.assemble(uv50, keep_visibility=True)
            vis, F = c.read_visibility(0, sc.N), c.read_rows(0, sc.N)
        Fo, viso, _ = ob.assemble_rows(ob.Mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n), uv50, bvh=True)
        assert np.array_equal(vis, viso) and np.array_equal(_bits(F), _bits(Fo))
    # a single triangle and a two-triangle scene: the paths have depth 0 / 1
    if mode == "paths":
        sc = scenes.cornell_box(1500, S=3)
        for n in (1, 2, 3):
            tvs, tns = sc.tri_v[:n], sc.tri_n[:n]
            with api.Context(0) as c:
                c.set_mesh(sc.vertices, sc.normals, tvs, tns)
                c.assemble(uv50, keep_visibility=True)
                vis, F = c.read_visibility(0, n), c.read_rows(0, n)
            Fo, viso, _ = ob.assemble_rows(ob.Mesh(sc.vertices, sc.normals, tvs, tns), uv50, bvh=False)
            assert np.array_equal(vis, viso) and np.array_equal(_bits(F), _bits(Fo))


@pytest.mark.parametrize("fenced", [0, 1])
@pytest.mark.parametrize("S", [3, 9])
def test_converge_on_a_column_split_pass(S, fenced):
    """8192 patches on one rank: few enough row blocks that the pass cuts its columns into (tapered) ranges by itself -- the
    in-launch hand-offs (partial sums -> last range of a row block; row blocks' residual sums -> last row block) feed the
    device-side convergence test: same pass count as the oracle, sums in the tails equal to the residual's sums (an
    independent reduction: k_chunk_sums after plain passes) -- in the shipped form of the hand-offs (write-through stores +
    ticket, a gfx950 property) and in the memory-model form (dr_options::sweep_fenced: release / acquire fences): same bits"""
    n = 8192
    sc = scenes.cornell_box(n, S=S, fluorescent=(S >= 8))
    E = sc.emission(7.0)
    with api.Context(0) as c:
        c.set_options(sweep_fenced=fenced)
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        c.integrand_only()                                   # unoccluded F: rows sum to about 1, the materials absorb
        F = c.read_rows(0, sc.N)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        thr, per_bin = (1e-2, True) if S == 3 else (20.0, False)
        it = c.converge(thr, per_bin=per_bin, max_iters=200)
        Bg, Rg = c.read()
        sums = c.residual_sums()
        it_o, Ro, Bo = ob.converge(F, sc.M, sc.mat_of_patch, E, thr, per_bin, 200)
        assert it == it_o and 2 < it < 200
        assert (np.abs(Bg - Bo) / (np.abs(Bo).max(axis=0) + 1e-30)).max() < 1e-4
        assert np.allclose(sums, Rg.astype(np.float64).sum(axis=0), rtol=1e-9)
        c.step(2)                                            # plain passes afterwards: sums on demand
        _, R2 = c.read(B=False)
        assert np.allclose(c.residual_sums(), R2.astype(np.float64).sum(axis=0), rtol=1e-9)
        assert c.info().sweep_ksplit > 1
        SPLIT_KEEP[(S, fenced)] = (it, _bits(Bg).copy(), _bits(Rg).copy(), sums.copy())
    other = SPLIT_KEEP.get((S, 1 - fenced))
    if other is not None:
        assert other[0] == it and np.array_equal(other[1], _bits(Bg)) and np.array_equal(other[2], _bits(Rg)) and np.array_equal(other[3], sums)


SPLIT_KEEP = {}


@pytest.mark.parametrize("n", [4095, 4097, 6144, 8191])
def test_sah_builder_size_classes_give_exact_rows(n, uv50):
    """the device's SAH builder around its size thresholds (a root that is just / just not a BIG node, the default's switch-over
    at 6 144 patches, a ragged last tile) and the pair walk on its trees: sampled rows of the ray counts and of F equal the
    oracle's (its own BVH), bit for bit"""
    sc = scenes.cornell_box(n, S=3)
    m = _mesh(sc)
    with api.Context(0) as c:
        c.set_options(tree=api.TREE_SAH)
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        i = c.info()
        assert i.tree_used == api.TREE_SAH and i.tree_on_host == 0
        c.assemble(uv50, keep_visibility=True)
        assert c.info().walk_used == api.WALK_PAIRS
        for r in np.linspace(0, n - 2, 5).astype(int):
            Fo, viso, _ = ob.assemble_rows(m, uv50, row0=int(r), nrows=2, bvh=True)
            assert np.array_equal(c.read_visibility(int(r), 2), viso), (n, r)
            assert np.array_equal(_bits(c.read_rows(int(r), 2)), _bits(Fo)), (n, r)


def test_group_in_pass_exchange_through_resets_and_mixed_calls(uv50):
    """the in-pass exchange keeps its pass numbers straight through every call order a viewer produces: steps, a reset, a converge
    that stops early (passes queued behind the converged one do nothing but still publish), more steps, a second converge --
    the same bits as one context doing the same calls"""
    sc = scenes.cornell_box(1300, S=8, fluorescent=True)
    E = sc.emission(7.0)

    def script(x):
        out = []
        x.step(3); out.append(x.read())
        x.reset()
        out.append(x.converge(5.0, per_bin=False, max_iters=40)); out.append(x.read())
        x.step(2); out.append(x.read())
        out.append(x.converge(0.5, per_bin=False, max_iters=300)); out.append(x.read())
        x.reset(); x.step(1); out.append(x.read())
        return out

    with _ctx(sc) as c:
        c.assemble(uv50)
        c.solver_init(E, sc.M, sc.mat_of_patch)
        want = script(c)
    with api.Group([0, 0, 0]) as g:
        g.set_options(group_exchange=api.GROUP_EXCHANGE_INPASS)
        g.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        g.assemble(uv50)
        g.solver_init(E, sc.M, sc.mat_of_patch)
        got = script(g)
    assert 3 < want[1] < 40 and 0 < want[4] < 300
    for a, b in zip(want, got):
        if isinstance(a, tuple):
            assert np.allclose(a[0], b[0], rtol=2e-6, atol=1e-12) and np.allclose(a[1], b[1], rtol=2e-6, atol=1e-12)
        else:
            assert a == b
