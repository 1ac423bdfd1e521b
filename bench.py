#!/usr/bin/env python3
"""bench.py -- radiosity hot path on MI355X: form-factor assembly + light-pass iteration.

Workload (BASELINE.json configs[3], the one the metric is quoted on; it fits one GPU):
synthetic subdivided Cornell box, N = 65 536 patches, S = 8 spectral bins, K = 50
visibility rays per pair, dense fp32 F resident in HBM (17.2 GB), row-sharded over the
ranks (strong scaling: the problem is fixed, each of the --gpus ranks owns N/gpus rows).

A "step" is one light pass (vs/Lightning.h:196-226): every rank streams its F shard once
for all S bins, then the residual vector is all-gathered (RCCL) -- exactly K steps are
timed between barriers.  The form-factor assembly that produces F runs once before the
timed region and is reported in "formfactors" (pairs/s).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--patches P] [--bins S]
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BARE_READ_GBS = 7215.0     # the best a kernel that only reads 17.18 GB reaches on this part (tools/hbm_probe.hip, profiles/r01/hbm_probe.md)


def sweep_bytes(nrows, N, S, n_mat):
    """ALGORITHMIC bytes one sweep launch moves on one rank:
    F shard read once + gathered residual read + residual out + B read/write + M."""
    return 4 * nrows * N + 4 * N * S + 4 * nrows * S * 3 + 4 * n_mat * S * S


def pmc_traffic(N, S, world):
    """HBM bytes per sweep launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    collected in separate runs, gfx950 corrections applied: profiles/r01/pmc_summary.md), when they
    were taken on this exact workload; None otherwise (PMC counters cannot be read live here)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_sweep_64k.json")))
        w = d["workload"]
        if (w["patches"], w["bins"], w["world"]) == (N, S, world):
            return d["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baselines(ctx, sc, uv, n_rows_sweep, n_rows_ff):
    """The oracle (a CPU port of the reference's algorithm) timed on this box's host cores
    on a bounded row sample of the same workload; single thread, like the reference."""
    import numpy as np

    from oracle import binding as ob

    N, S = sc.N, sc.S
    info = ctx.info()
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    r0 = info.row0
    _, R = ctx.read(B=False, R=True)
    # a 256-row probe sizes the sample to about 10 s of single-thread work
    n0 = min(256, info.nrows)
    F = ctx.read_rows(r0, n0)
    B = np.zeros((n0, S), np.float32)
    t = time.perf_counter()
    ob.sweep_rows(F, sc.M, sc.mat_of_patch, R, B, row0=r0, threads=1)
    dt = time.perf_counter() - t
    n1 = min(info.nrows, n_rows_sweep, max(n0, int(n0 * 10.0 / max(dt, 1e-6))))
    if n1 != n0:
        F = ctx.read_rows(r0, n1)
        B = np.zeros((n1, S), np.float32)
        t = time.perf_counter()
        ob.sweep_rows(F, sc.M, sc.mat_of_patch, R, B, row0=r0, threads=1)
        dt = time.perf_counter() - t
    iters = 1.0 / (dt * (N / n1))
    sweep = {"value": iters, "unit": "iters/s", "cores": 1, "kind": "port", "cpu": cpu_model,
             "sample": "%d of %d rows of the same F and residual, all %d bins, oracle/orc_sweep_rows, %.1f s"
                       % (n1, N, S, dt)}
    n2 = min(n_rows_ff, info.nrows)
    m = ob.Mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
    t = time.perf_counter()
    ob.assemble_rows(m, uv, row0=r0, nrows=n2, want_vis=False, threads=1, bvh=True)
    dt2 = time.perf_counter() - t
    ff = {"value": n2 * (N - 1) / dt2, "unit": "pairs/s", "cores": 1, "kind": "port",
          "sample": "%d rows x %d columns, integrand + %d rays per facing pair through the oracle's BVH, %.1f s"
                    % (n2, N, uv.shape[0], dt2)}
    # the same two samples with OpenMP over rows on every host core (the reference itself has no threading)
    # (a one-GPU box grants about 16 host CPUs however many the machine has: more threads only thrash)
    cores = max(1, min(ob.num_threads(), len(os.sched_getaffinity(0)), 16))
    B = np.zeros((n1, S), np.float32)
    t = time.perf_counter()
    ob.sweep_rows(F, sc.M, sc.mat_of_patch, R, B, row0=r0, threads=cores)
    dt3 = time.perf_counter() - t
    n3 = min(info.nrows, cores)
    t = time.perf_counter()
    ob.assemble_rows(m, uv, row0=r0, nrows=n3, want_vis=False, threads=cores, bvh=True)
    dt4 = time.perf_counter() - t
    allcores = {"cores": cores, "kind": "port", "cpu": cpu_model, "iters_per_s": 1.0 / (dt3 * (N / n1)),
                "pairs_per_s": n3 * (N - 1) / dt4,
                "sample": "same row samples (%d sweep rows, %d assembly rows), OpenMP over rows, %.1f + %.1f s" % (n1, n3, dt3, dt4)}
    return sweep, ff, allcores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--patches", type=int, default=65536)
    ap.add_argument("--bins", type=int, default=8)
    ap.add_argument("--rays", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-zero-block-report", action="store_true",
                    help="skip the extra passes with zero-block skipping (keeps a kernel profile of this run to the dense pass)")
    ap.add_argument("--rehearse-comm", action="store_true",
                    help="single rank only: still create the torch process group and the library's RCCL communicator "
                         "(all-gather of one chunk per pass), to rehearse the multi-GPU code path on a one-GPU box")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: libraries that print there on their own (RCCL's version banner at
    # communicator creation, for one) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    from daisyriot_amd import api, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        sys.exit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback for the hot path)")
    torch.cuda.set_device(local_rank)
    use_comm = world > 1 or args.rehearse_comm
    if use_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def barrier():
        if use_comm:
            dist.barrier()

    sc = scenes.cornell_box(args.patches, S=args.bins)
    uv = scenes.visibility_samples(args.rays)
    E = sc.emission(7.0)
    N, S = sc.N, sc.S

    ctx = api.Context(local_rank)
    ctx.set_shard(rank, world)
    if use_comm:
        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt.copy_(torch.from_numpy(api.comm_unique_id()))
        dist.broadcast(idt, 0)
        ctx.comm_init(idt.cpu().numpy(), rank, world)
    ctx.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)

    # ---- form-factor assembly (once; produces the F the passes stream) --------------------
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.assemble(uv, rule=api.RULE_INTEGRAND)
    torch.cuda.synchronize()
    barrier()
    t_asm = time.perf_counter() - t0
    info = ctx.info()
    asm = torch.tensor([t_asm, info.last_assemble_ms / 1e3, float(info.pairs_traced), info.last_bvh_ms / 1e3],
                       dtype=torch.float64, device="cuda")
    if use_comm:
        mx = asm.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = asm.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        t_asm, t_asm_kernel, pairs_traced, t_bvh = float(mx[0]), float(mx[1]), float(sm[2]), float(mx[3])
    else:
        t_asm, t_asm_kernel, pairs_traced, t_bvh = (float(x) for x in asm)

    # ---- light passes --------------------------------------------------------------------------
    ctx.solver_init(E, sc.M, sc.mat_of_patch)
    for _ in range(args.warmup):
        ctx.step(1)
    ctx.synchronize()
    ctx.profile(True)
    ctx.profile_reset()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.step(1)
    ctx.synchronize()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile(False)
    info = ctx.info()
    tt = torch.tensor([dt, info.sweep_ms_total / max(1, info.sweep_launches)], dtype=torch.float64, device="cuda")
    if use_comm:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt, kern_ms = float(tt[0]), float(tt[1])
    resid = ctx.step(0, want_sum=True)

    # ---- the same passes with the optional zero-block skipping (reported beside the headline, never in it) ----
    skip_report = None
    if not args.no_zero_block_report:
        ctx.skip_zero_blocks(True)
        for _ in range(max(2, args.warmup)):
            ctx.step(1)
        ctx.synchronize()
        ctx.profile(True)
        ctx.profile_reset()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.step(1)
        ctx.synchronize()
        torch.cuda.synchronize()
        barrier()
        dt_skip = time.perf_counter() - t0
        ctx.profile(False)
        info_skip = ctx.info()
        ts = torch.tensor([dt_skip, info_skip.sweep_ms_total / max(1, info_skip.sweep_launches)], dtype=torch.float64, device="cuda")
        if use_comm:
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        ctx.skip_zero_blocks(False)
        skip_report = {"iters_per_s": args.steps / float(ts[0]), "kernel_ms_avg": float(ts[1]),
                       "blocks_nonzero": int(info_skip.blocks_nonzero), "blocks_total": int(info_skip.blocks_total),
                       "F_bytes_read_per_launch": int(info_skip.blocks_nonzero) * 32 * 256 * 4,
                       "note": "rank 0's shard; same F, same passes, results bit-identical to the dense pass"}

    if rank == 0:
        n_mat = sc.M.shape[0]
        alg = sweep_bytes(info.nrows, N, S, n_mat)
        achieved = alg / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "radiosity iters/s at N=%dk patches (patch-pair form-factors/s in \"formfactors\")" % (N // 1024),
            "value": args.steps / dt,
            "unit": "iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "subdivided Cornell box, %d patches, %d spectral bins, K=%d rays/pair, dense fp32 F "
                                   "row-sharded over %d GPU(s), one RCCL all-gather of the residual per pass"
                                   % (N, S, args.rays, world),
                       "patches": N, "bins": S, "rays_per_pair": args.rays,
                       "F_bytes_per_gpu": int(info.bytes_F), "rows_per_gpu": int(info.nrows)},
            "roofline": {"bound": "hbm", "kernel": "k_sweep", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(N, S, world),
                         "algorithmic_bytes_per_launch": alg, "kernel_ms_avg": kern_ms,
                         "bare_read_GBs": BARE_READ_GBS, "frac_of_bare_read": achieved / BARE_READ_GBS,
                         "launches_timed": int(info.sweep_launches)},
            "formfactors": {"value": N * (N - 1) / t_asm, "unit": "pairs/s", "seconds": t_asm,
                            "kernel_seconds": t_asm_kernel, "bvh_build_seconds": t_bvh,
                            "pairs_traced": pairs_traced, "rays_per_s": pairs_traced * args.rays / t_asm},
            "residual_sum_after_timed_passes": resid,
            # optional dr_solver_skip_zero_blocks: all-zero 32 x 256 blocks of F are not read (bit-identical results);
            # not part of "value" -- the headline streams the whole dense matrix
            "zero_block_skipping": skip_report,
        }
        if world == 1 and not args.no_cpu_baseline:
            sweep_cpu, ff_cpu, allcores = cpu_baselines(ctx, sc, uv, 65536, 2)
            out["cpu_baseline"] = sweep_cpu
            out["cpu_baseline_formfactors"] = ff_cpu
            out["cpu_baseline_all_cores"] = allcores
            out["formfactors"]["vs_cpu_port"] = out["formfactors"]["value"] / ff_cpu["value"]
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ctx.close()
    if use_comm:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
