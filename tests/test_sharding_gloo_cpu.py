"""The N>1 path on CPU: two gloo ranks run the row-sharded light-pass protocol of
dr_solver_step (own rows -> own chunk of the gathered residual -> in-place all-gather ->
next pass) using the library's own shard arithmetic (dr_shard_rows / dr_residual_offset,
pure host code of libdaisyriot_hip.so).  The per-row arithmetic is the oracle's (there is
no GPU here); the result must equal the unsharded oracle bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from daisyriot_amd import api, scenes
from oracle import binding as ob


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, N, S, passes, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.cornell_box(N, S=S, fluorescent=True)
    uv = scenes.visibility_samples(50)
    row0, nrows, rpr = api.shard_rows(sc.N, rank, world)
    m = ob.Mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
    # each rank assembles only its own rows (no communication in the assembly)
    F, _, _ = ob.assemble_rows(m, uv, row0=row0, nrows=nrows, bvh=True, want_vis=False)
    E = sc.emission(7.0)
    # gathered residual in the device layout [world][chunk]: a chunk = [S][rpr] + the chunk's per-bin sums (16 doubles)
    cf = api.residual_chunk_floats(S, rpr)
    full = np.zeros(world * cf, np.float32)
    idx = np.array([[api.residual_offset(i, s, S, rpr) for s in range(S)] for i in range(sc.N)])
    full[idx] = E
    B = E[row0:row0 + nrows].copy()
    sums_seen = []
    for _ in range(passes):
        Rin = full[idx]                                       # N x S view of the gathered buffer
        Rout = ob.sweep_rows(F, sc.M, sc.mat_of_patch, Rin, B, row0=row0)
        chunk = np.zeros(cf, np.float32)                      # own chunk, bin-major, zero padded
        for s in range(S):
            chunk[s * rpr:s * rpr + nrows] = Rout[:, s]
        # the rank's share of the convergence sums rides in the same message (no second collective)
        chunk[S * rpr:].view(np.float64)[:S] = Rout.astype(np.float64).sum(axis=0)
        gathered = torch.zeros(world * cf)
        dist.all_gather_into_tensor(gathered, torch.from_numpy(chunk))
        full = gathered.numpy().copy()
        tails = full.reshape(world, cf)[:, S * rpr:].copy().view(np.float64)[:, :S]
        sums_seen.append(tails.sum(axis=0))
    np.save(os.path.join(out_dir, "B_%d.npy" % rank), B)
    np.save(os.path.join(out_dir, "R_%d.npy" % rank), full[idx])
    np.save(os.path.join(out_dir, "sums_%d.npy" % rank), np.array(sums_seen))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_passes_equal_unsharded(world, tmp_path):
    N, S, passes = 300, 8, 4
    mp.spawn(_rank_main, args=(world, _free_port(), N, S, passes, str(tmp_path)), nprocs=world, join=True)
    sc = scenes.cornell_box(N, S=S, fluorescent=True)
    m = ob.Mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
    F, _, _ = ob.assemble_rows(m, scenes.visibility_samples(50), bvh=True, want_vis=False)
    E = sc.emission(7.0)
    R, B = E.copy(), E.copy()
    sums = []
    for _ in range(passes):
        R = ob.sweep_rows(F, sc.M, sc.mat_of_patch, R, B)
        sums.append(R.astype(np.float64).sum(axis=0))
    Bs = []
    for r in range(world):          # every rank saw the same per-bin sums after every pass: they all stop at the same pass
        got = np.load(tmp_path / ("sums_%d.npy" % r))
        assert np.array_equal(got, np.load(tmp_path / "sums_0.npy"))
        assert np.allclose(got, np.array(sums), rtol=1e-12)
    for r in range(world):
        row0, nrows, _ = api.shard_rows(N, r, world)
        Br = np.load(tmp_path / ("B_%d.npy" % r))
        assert Br.shape[0] == nrows
        Bs.append(Br)
        Rr = np.load(tmp_path / ("R_%d.npy" % r))
        assert np.array_equal(Rr.view(np.uint32), R.view(np.uint32)), "rank %d holds a different residual" % r
    assert np.array_equal(np.concatenate(Bs).view(np.uint32), B.view(np.uint32))


def _assembly_rank_main(rank, world, port, N, out_dir):
    """the ray-count exchange of a multi-rank assembly over real messages: every tile pair between two ranks' rows
    is 'traced' (here: taken from the oracle) by the rank dr_vis_exchange_tracer names, its 64 x 64 counts go to slot
    [other rank][own tile][foreign tile] of the rank's send blocks, block p travels to rank p only (all-to-all), and the
    other rank picks slot [tracer rank][foreign tile][own tile] of what it received"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.cornell_box(N, S=3)
    uv = scenes.visibility_samples(50)
    row0, nrows, rpr = api.shard_rows(sc.N, rank, world)
    m = ob.Mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
    T, nT = rpr // 64, (sc.N + 63) // 64
    # what this rank may look at: the counts of its own rows (it could trace any pair that touches them)
    _, Vown, _ = ob.assemble_rows(m, uv, row0=row0, nrows=nrows, bvh=True, want_vis=True)

    def tile(V_rows, o, t):                 # counts of own tile o x tile t, rows = the tile of the lower index
        blk = np.full((64, 64), 255, np.uint8)
        r0, c0 = o * 64 - row0, t * 64
        sub = V_rows[r0:r0 + 64, c0:c0 + 64]
        blk[:sub.shape[0], :sub.shape[1]] = sub
        return blk if o < t else blk.T

    send = np.full((world, T, T, 64, 64), 255, np.uint8)
    own_tiles = range(row0 // 64, (row0 + nrows + 63) // 64)
    traced = 0
    for o in own_tiles:
        for t in range(nT):
            if t // T == rank:
                continue
            if api.vis_exchange_tracer(sc.N, world, o * 64, t * 64) == rank:
                send[t // T, o - rank * T, t % T] = tile(Vown, o, t)
                traced += 1
    recv = np.full((world, T, T, 64, 64), 255, np.uint8)
    reqs, bufs = [], {}
    for p in range(world):
        if p == rank:
            continue
        bufs[p] = torch.zeros(send[p].size, dtype=torch.uint8)
        reqs.append(dist.irecv(bufs[p], src=p))
        reqs.append(dist.isend(torch.from_numpy(send[p].reshape(-1).copy()), dst=p))
    for q in reqs:
        q.wait()
    for p, b in bufs.items():
        recv[p] = b.numpy().reshape(T, T, 64, 64)
    # rebuild the counts of the own rows: own x own and own-traced pairs locally, the rest from the slots
    V = np.full((nrows, sc.N), 255, np.uint8)
    for o in own_tiles:
        for t in range(nT):
            mine = (t // T == rank) or api.vis_exchange_tracer(sc.N, world, o * 64, t * 64) == rank
            blk = tile(Vown, o, t) if mine else recv[t // T, t % T, o - rank * T]
            blk = blk if o < t else blk.T       # back to rows = this rank's tile
            r0, c0 = o * 64 - row0, t * 64
            h, w = min(64, nrows - r0), min(64, sc.N - c0)
            V[r0:r0 + h, c0:c0 + w] = blk[:h, :w]
    np.save(os.path.join(out_dir, "V_%d.npy" % rank), V)
    np.save(os.path.join(out_dir, "n_%d.npy" % rank), np.array([traced]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ray_count_exchange_protocol_over_a_collective(world, tmp_path):
    N = 700
    mp.spawn(_assembly_rank_main, args=(world, _free_port(), N, str(tmp_path)), nprocs=world, join=True)
    sc = scenes.cornell_box(N, S=3)
    m = ob.Mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
    _, V, _ = ob.assemble_rows(m, scenes.visibility_samples(50), bvh=True, want_vis=True)
    n_cross = 0
    for r in range(world):
        row0, nrows, rpr = api.shard_rows(N, r, world)
        assert np.array_equal(np.load(tmp_path / ("V_%d.npy" % r)), V[row0:row0 + nrows]), r
        n_cross += int(np.load(tmp_path / ("n_%d.npy" % r))[0])
    # every tile pair between two different ranks was traced exactly once
    T = api.shard_rows(N, 0, world)[2] // 64
    nT = (N + 63) // 64
    want = sum(1 for a in range(nT) for b in range(a + 1, nT) if a // T != b // T)
    assert n_cross == want
