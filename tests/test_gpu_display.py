"""Display colours on the MI355X against the golden vectors the reference's own color.h + glm
produced (tests/golden/display_color_h.npz) and against the oracle: bit-exact."""
import os

import numpy as np
import pytest

from daisyriot_amd import api, scenes
from oracle import binding as ob

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _ctx_with_B(sc, B, world=1, rank=0):
    ctx = api.Context(0)
    if world > 1:
        ctx.set_shard(rank, world)
        ctx.comm_manual()
    ctx.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
    M = np.zeros((1, B.shape[1], B.shape[1]), np.float32)
    ctx.solver_init(B, M, np.zeros(sc.N, np.int32))      # B = E before the first pass
    return ctx


def test_spectral_colour_cache_and_vertex_means_match_the_reference_header():
    g = np.load(os.path.join(GOLD, "display_color_h.npz"))
    sc = scenes.cornell_box(128, S=9, fluorescent=True)
    N = sc.N
    assert np.array_equal(sc.tri_v, g["tri_v"])
    xyz = g["xyz_bits"].view(np.float32)
    want = g["rgb_bits"].view(np.float32)
    for k in range(3):                                   # three exposure levels: below 1, mixed, all normalised
        B = np.ascontiguousarray(g["B"][k * N:(k + 1) * N])
        ctx = _ctx_with_B(sc, B)
        got = ctx.patch_colors(api.DISPLAY_SPECTRAL, xyz)
        assert np.array_equal(_bits(got), _bits(want[k * N:(k + 1) * N])), k
        if k == 1:
            vtx = ctx.vertex_colors(g["vtx_off"], g["vtx_tri"])            # from the colours left on the device
            assert np.array_equal(_bits(vtx), g["vertex_rgb_bits"])
            vtx2 = ctx.vertex_colors(g["vtx_off"], g["vtx_tri"], rgb_all=want[N:2 * N])
            assert np.array_equal(_bits(vtx2), g["vertex_rgb_bits"])
        ctx.close()


@pytest.mark.parametrize("S,mode", [(1, api.DISPLAY_BW), (3, api.DISPLAY_RGB), (8, api.DISPLAY_SPECTRAL), (16, api.DISPLAY_SPECTRAL)])
def test_display_modes_after_light_passes(S, mode, uv50):
    sc = scenes.cornell_box(700, S=S)
    ctx = api.Context(0)
    ctx.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
    ctx.assemble(uv50)
    ctx.solver_init(sc.emission(7.0), sc.M, sc.mat_of_patch)
    ctx.step(4)
    B, _ = ctx.read()
    xyz = ob.xyz_fit(np.linspace(380.0, 700.0, S)) if mode == api.DISPLAY_SPECTRAL else None
    got = ctx.patch_colors(mode, xyz)
    want = ob.patch_colors(B, mode, xyz)
    assert np.array_equal(_bits(got), _bits(want))
    off, adj = scenes.vertex_adjacency(sc.tri_v, sc.vertices.shape[0])
    assert np.array_equal(_bits(ctx.vertex_colors(off, adj)), _bits(ob.vertex_colors(off, adj, want)))
    ctx.close()


def test_sharded_colours_and_errors():
    sc = scenes.cornell_box(700, S=8)
    B = np.random.RandomState(3).uniform(0, 3, size=(sc.N, 8)).astype(np.float32)
    xyz = ob.xyz_fit(scenes.WAVELENGTHS_8)
    want = ob.patch_colors(B, api.DISPLAY_SPECTRAL, xyz)
    off, adj = scenes.vertex_adjacency(sc.tri_v, sc.vertices.shape[0])
    parts = []
    for rank in range(3):
        ctx = _ctx_with_B(sc, B, world=3, rank=rank)
        info = ctx.info()
        got = ctx.patch_colors(api.DISPLAY_SPECTRAL, xyz)
        assert got.shape == (info.nrows, 3)
        assert np.array_equal(_bits(got), _bits(want[info.row0:info.row0 + info.nrows]))
        parts.append(got)
        with pytest.raises(api.DaisyRiotError):          # a rank holds only its rows: vertex means need all colours
            ctx.vertex_colors(off, adj)
        if rank == 2:
            allc = np.concatenate(parts, axis=0)
            assert np.array_equal(_bits(ctx.vertex_colors(off, adj, rgb_all=allc)), _bits(ob.vertex_colors(off, adj, want)))
        ctx.close()
    ctx = _ctx_with_B(sc, B)
    with pytest.raises(api.DaisyRiotError):
        ctx.patch_colors(api.DISPLAY_RGB)                 # 8 bins are not RGB
    with pytest.raises(api.DaisyRiotError):
        ctx.patch_colors(api.DISPLAY_SPECTRAL)            # no fit values
    with pytest.raises(api.DaisyRiotError):
        ctx.patch_colors(7)
    bad = adj.copy()
    bad[5] = sc.N
    with pytest.raises(api.DaisyRiotError):
        ctx.vertex_colors(off, bad)
    ctx.close()
