"""The SAH tree topology (daisyriot_amd/csrc/geom_kernels.hip): the host's reference builder (sah_hierarchy_host, no GPU needed:
the library's host-only debug entry) and, marked gpu, the DEVICE builder dr_scene_set_mesh runs (sah_hierarchy_device) on the
same inputs -- both must produce a full binary tree over a permutation of the boxes, logarithmic whatever the layout."""
import time

import numpy as np
import pytest

from daisyriot_amd import api, scenes


def _check_tree(t, N):
    order, left, right, first, last, parent = (t[k] for k in ("order", "left", "right", "first", "last", "parent"))
    assert sorted(order.tolist()) == list(range(N))                       # a permutation: every box in exactly one leaf
    if N == 1:
        assert parent[0] == -1
        return 0
    assert parent[0] == -1
    depth = np.zeros(2 * N - 1, np.int64)
    seen = np.zeros(2 * N - 1, bool)
    stack = [0]
    seen[0] = True
    while stack:
        i = stack.pop()
        kids = (int(left[i]), int(right[i]))
        lo = int(first[i])
        for k in kids:
            assert 0 <= k < 2 * N - 1 and not seen[k] and parent[k] == i
            seen[k] = True
            depth[k] = depth[i] + 1
            if k >= N - 1:                                                # a leaf: one position, where the range says
                assert k - (N - 1) == lo
                lo += 1
            else:
                assert first[k] == lo and last[k] >= first[k] + 1         # an interior node: a contiguous range of >= 2
                lo = int(last[k]) + 1
                stack.append(k)
        assert lo == int(last[i]) + 1                                     # the children's ranges tile the parent's
    assert seen.all()                                                     # N - 1 interior nodes and N leaves, all reached
    assert first[0] == 0 and last[0] == N - 1
    return int(depth.max())


def _boxes_of(sc):
    v = sc.vertices[sc.tri_v]                                             # [N][3][3]
    return np.concatenate([v.min(axis=1), v.max(axis=1)], axis=1).astype(np.float32)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 64, 1000])
def test_topology_is_a_full_binary_tree_over_a_permutation(n):
    rs = np.random.RandomState(n)
    c = rs.random_sample((n, 3)).astype(np.float32) * 10
    e = rs.random_sample((n, 3)).astype(np.float32) * 0.3
    _check_tree(api.sah_topology(np.concatenate([c - e, c + e], axis=1)), n)


def test_cornell_box_tree_is_shallow_and_deterministic():
    sc = scenes.cornell_box(16384, S=3)
    b = _boxes_of(sc)
    t0 = time.perf_counter()
    t = api.sah_topology(b)
    dt = time.perf_counter() - t0
    d = _check_tree(t, sc.N)
    assert d <= 2 * int(np.ceil(np.log2(sc.N))) + 4                       # no degenerate chains on a regular scene
    t2 = api.sah_topology(b)
    assert all(np.array_equal(t[k], t2[k]) for k in t)                    # same input, same tree (every rank builds its own)
    assert dt < 2.0                                                       # 16k boxes: milliseconds on one core


def test_lopsided_and_coincident_inputs_stay_logarithmic():
    """boxes strung along a line at geometrically growing distances (every SAH cut wants to peel one box off) and a pile of
    identical boxes (no plane separates them): the median fallback bounds the depth"""
    n = 4096
    x = (1.01 ** np.arange(n)).astype(np.float32)
    line = np.stack([x, np.zeros(n, np.float32), np.zeros(n, np.float32)], axis=1)
    d = _check_tree(api.sah_topology(np.concatenate([line - 0.001, line + 0.001], axis=1)), n)
    assert d <= 6 * int(np.ceil(np.log2(n)))
    pile = np.tile(np.array([[0, 0, 0, 1, 1, 1]], np.float32), (n, 1))
    d = _check_tree(api.sah_topology(pile), n)
    assert d <= int(np.ceil(np.log2(n))) + 1


def test_quad_mates_share_a_leaf_pair():
    """the two triangles of an axis-aligned quad have the same box: the build never separates them before the last split,
    so k_emit can collapse them into one two-triangle leaf"""
    sc = scenes.closed_box(cells=8, S=3)
    b = _boxes_of(sc)
    t = api.sah_topology(b)
    _check_tree(t, sc.N)
    pos = np.empty(sc.N, np.int64)
    pos[t["order"]] = np.arange(sc.N)
    same = [(i, i + 1) for i in range(0, sc.N - 1, 2) if np.array_equal(b[i], b[i + 1])]
    assert len(same) > sc.N // 4
    together = sum(1 for i, j in same if abs(pos[i] - pos[j]) == 1 and t["parent"][sc.N - 1 + pos[i]] == t["parent"][sc.N - 1 + pos[j]])
    assert together == len(same)


def test_denormal_extents_are_degenerate_axes():
    """centroids that differ by a denormal on one axis: bins / extent overflows to inf there and (centroid - lowest) * inf would
    be 0 * inf = NaN for the lowest centroid -- such an axis counts as having no extent (round-2 advisor finding)"""
    n = 200
    rs = np.random.RandomState(3)
    c = np.zeros((n, 3), np.float32)
    c[:, 0] = rs.random_sample(n).astype(np.float32)
    c[1::2, 1] = np.float32(1e-40)                                        # a denormal step along y
    c[:, 2] = (rs.randint(0, 2, n) * np.float32(1e-44)).astype(np.float32)
    e = np.float32(0.01)
    d = _check_tree(api.sah_topology(np.concatenate([c - e, c + e], axis=1)), n)
    assert d <= 4 * int(np.ceil(np.log2(n)))
    only_y = np.zeros((n, 3), np.float32)
    only_y[1::2, 1] = np.float32(1e-40)                                   # nothing but the denormal step: the median fallback
    d = _check_tree(api.sah_topology(np.concatenate([only_y - e, only_y + e], axis=1)), n)
    assert d <= int(np.ceil(np.log2(n))) + 2


def _inputs_for_the_device():
    rs = np.random.RandomState(17)
    out = {}
    for n in (1, 2, 3, 7, 64, 65, 1000, 5000):
        c = rs.random_sample((n, 3)).astype(np.float32) * 10
        e = rs.random_sample((n, 3)).astype(np.float32) * 0.3
        out["random-%d" % n] = np.concatenate([c - e, c + e], axis=1)
    n = 4096
    x = (1.01 ** np.arange(n)).astype(np.float32)
    line = np.stack([x, np.zeros(n, np.float32), np.zeros(n, np.float32)], axis=1)
    out["line"] = np.concatenate([line - 0.001, line + 0.001], axis=1)
    out["pile"] = np.tile(np.array([[0, 0, 0, 1, 1, 1]], np.float32), (n, 1))
    big_pile = np.tile(np.array([[0, 0, 0, 1, 1, 1]], np.float32), (9000, 1))           # a BIG node without a plane: median by chunks
    big_pile[::3, 0] += 5.0
    out["big-pile"] = big_pile
    dn = np.zeros((300, 3), np.float32)
    dn[1::2, 1] = np.float32(1e-40)
    out["denormal"] = np.concatenate([dn - np.float32(0.01), dn + np.float32(0.01)], axis=1)
    sc = scenes.cornell_box(20000, S=3)
    out["cornell-20000"] = _boxes_of(sc)
    return out


@pytest.mark.gpu
def test_device_builder_makes_valid_logarithmic_trees():
    """the device's builder (tiny nodes: one wave each; big ones: chunk-parallel passes; the rest: a workgroup each) on random
    boxes of every size class, on layouts that defeat binning, and on a scene: a full binary tree over a permutation, no deeper
    than the host's reference tree plus a few levels"""
    with api.Context(0) as c:
        for name, boxes in _inputs_for_the_device().items():
            n = boxes.shape[0]
            d_dev = _check_tree(c.sah_topology(boxes), n)
            d_host = _check_tree(api.sah_topology(boxes), n)
            assert d_dev <= d_host + 3, (name, d_dev, d_host)
            assert d_dev <= 6 * int(np.ceil(np.log2(max(n, 2)))), (name, d_dev)
