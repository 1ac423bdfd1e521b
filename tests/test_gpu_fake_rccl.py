"""The multi-PROCESS path on one GPU: `world` processes, one context each, joined by the library's communicator calls --
dr_comm_unique_id / dr_comm_init, the go / no-go agreement and the all-to-all of ray-count slots inside
dr_formfactors_assemble, the all-gather of the residual (with the convergence sums in its tails) after every pass,
dr_solver_converge's device-side test deciding alike on every rank.  Real RCCL refuses several ranks on one device, so the
ranks bind tests/fake_rccl/libfake_rccl.so (DR_RCCL_LIB), a loop-back implementation of the same entry points over shared
memory: what is tested is everything on the library's side of that API.  (Real RCCL with world = 1 is covered by
test_rccl_binding_single_rank; with world > 1 it needs a multi-GPU node.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from daisyriot_amd import api, scenes

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FAKE_DIR = os.path.join(HERE, "fake_rccl")
FAKE = os.path.join(FAKE_DIR, "libfake_rccl.so")


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _run_ranks(world, S, n, rule, tmp_path, **extra_env):
    subprocess.check_call(["make", "-C", FAKE_DIR], stdout=subprocess.DEVNULL)
    env = dict(os.environ, DR_RCCL_LIB=FAKE, FAKE_RCCL_SLOT_MB="16", **extra_env)
    env.pop("DR_NO_VIS_EXCHANGE", None)
    id_file = str(tmp_path / "id.npy")
    procs = [subprocess.Popen([sys.executable, os.path.join(FAKE_DIR, "rank_main.py"), str(r), str(world), str(n), str(S), str(rule),
                               id_file, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    return procs, outs


# (3, 3, 400): 400 patches in shards of 256 rows leave rank 2 WITHOUT rows -- it still has to take every decision of a
# converge run with the others (the reference's 6400-patch scene on 8 GPUs has such a rank)
@pytest.mark.parametrize("world,S,n,rule", [(2, 8, 700, api.RULE_INTEGRAND), (3, 3, 2500, api.RULE_RECIPROCITY),
                                            (3, 3, 400, api.RULE_INTEGRAND)])
def test_ranks_in_separate_processes(world, S, n, rule, tmp_path):
    procs, outs = _run_ranks(world, S, n, rule, tmp_path)
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    # the same scene on one context
    sc = scenes.cornell_box(n, S=S, fluorescent=(S >= 8))
    uv = scenes.visibility_samples(50)
    E = sc.emission(7.0)
    with api.Context(0) as c:
        c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
        c.assemble(uv, rule=rule, keep_visibility=True)
        F, V = c.read_rows(0, sc.N), c.read_visibility(0, sc.N)
        traced_once = c.info().pairs_traced
        c.solver_init(E, sc.M, sc.mat_of_patch)
        c.step(4)
        B4, R4 = c.read()
        c.reset()
        thr, per_bin = (1e-3, True) if S == 3 else (0.5, False)
        it = c.converge(thr, per_bin=per_bin, max_iters=300)
        Bc, Rc = c.read()
    ranks = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    assert sum(int(d["traced"]) for d in ranks) == traced_once          # every pair traced by exactly one rank
    if n == 400:
        assert int(ranks[2]["nrows"]) == 0 and it > 8                   # an empty rank, and more than one batch of queued passes
    for d in ranks:
        row0, nrows = int(d["row0"]), int(d["nrows"])
        assert np.array_equal(d["V"], V[row0:row0 + nrows])
        assert np.array_equal(_bits(d["F"]), _bits(F[row0:row0 + nrows]))
        assert np.allclose(d["B4"], B4[row0:row0 + nrows], rtol=2e-6, atol=1e-12)
        assert np.allclose(d["R4"], R4, rtol=2e-6, atol=1e-12)
        assert np.array_equal(_bits(d["R4"]), _bits(ranks[0]["R4"]))    # every rank holds the same gathered residual
        assert abs(float(d["sum4"]) - R4.astype(np.float64).sum()) <= 1e-5 * max(1.0, R4.sum())
        assert int(d["iters"]) == it                                     # all ranks stop at the same pass
        assert np.allclose(d["Bc"], Bc[row0:row0 + nrows], rtol=2e-6, atol=1e-12)
        assert np.array_equal(_bits(d["Rc"]), _bits(ranks[0]["Rc"]))


def test_a_failing_rank_does_not_strand_its_peers(tmp_path):
    """after the ranks agreed to start, rank 1's first assembly launch "fails" (DR_FAULT_ASSEMBLE_RANK): the second go / no-go
    before the all-to-all turns that into an error on EVERY rank instead of peers waiting in the collective"""
    procs, outs = _run_ranks(2, 3, 700, api.RULE_INTEGRAND, tmp_path, DR_FAULT_ASSEMBLE_RANK="1")
    assert all(p.returncode != 0 for p in procs), [p.returncode for p in procs]
    assert "injected failure" in outs[1] and "another rank failed" in outs[0], outs


def test_bench_launches_its_own_ranks(tmp_path):
    """`bench.py --gpus 2` without a torchrun environment starts its ranks itself and relays rank 0's line; here as the
    shared-GPU rehearsal (two processes on GPU 0, gloo + the loop-back stand-in): the line proves two ranks through the
    communicators' own rank / size, and the rows of the two shards add up"""
    import json
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--shared-gpu-rehearsal",
                        "--patches", "8192", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_world"] == 2 and d["rccl_ranks"] == [0, 1] and d["rehearsal_on_shared_gpu"] is True
    assert sum(d["config"]["rows_per_gpu"]) == 8192 and len(d["roofline"]["per_rank"]) == 2
    assert d["value"] > 0 and d["formfactors"]["pairs_traced"] > 0 and d["converge_mode"]["passes"] == 4
    # on a box with fewer GPUs than ranks the real multi-process run stops with a message instead of hanging
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "this box has" in r.stderr
