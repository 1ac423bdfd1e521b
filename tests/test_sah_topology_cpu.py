"""Host logic of the BVH build (no GPU): the SAH tree topology dr_scene_set_mesh hands to the device kernels from 6 144 patches
up (daisyriot_amd/csrc/geom_kernels.hip: sah_hierarchy_host), through the library's host-only debug entry."""
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
