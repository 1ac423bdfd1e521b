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

N > 1 without a torchrun environment: this process starts the N ranks itself (children, one per
GPU, before anything here touches a GPU) and relays rank 0's line; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.
`--group`: one process drives all N GPUs through the library's dr_group (RCCL inside the library,
no torch.distributed); `--devices 0,0` rehearses that on one GPU.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PROFILE_ROUNDS = ("r03", "r02", "r01")


def sweep_bytes(nrows, N, S, n_mat):
    """ALGORITHMIC bytes one sweep launch moves on one rank:
    F shard read once + gathered residual read + residual out + B read/write + M."""
    return 4 * nrows * N + 4 * N * S + 4 * nrows * S * 3 + 4 * n_mat * S * S


def kernel_source_sha():
    """identifies the pass kernel a committed PMC file was taken on: sweep_kernels.hip and the SweepParams struct"""
    import re
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "daisyriot_amd", "csrc")
    h.update(open(os.path.join(csrc, "sweep_kernels.hip"), "rb").read())
    m = re.search(r"struct SweepParams \{.*?\n\};", open(os.path.join(csrc, "dr_internal.h")).read(), flags=re.S)
    h.update((m.group(0) if m else "").encode())
    return h.hexdigest()[:16]


def pmc_traffic(N, S, world):
    """HBM bytes per sweep launch from committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
    separate runs, gfx950 corrections applied: tools/pmc_sweep.py) -- only when they were taken on this exact
    workload AND on the pass kernel as it is now (source hash); otherwise (None, why).  PMC counters cannot be
    read from inside this process."""
    sha = kernel_source_sha()
    why = "no committed PMC passes for this workload"
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", rnd, "pmc_sweep_64k.json")
        try:
            d = json.load(open(path))
            w = d["workload"]
            if (w["patches"], w["bins"], w["world"]) != (N, S, world):
                continue
            if d.get("kernel_source_sha") != sha:
                why = "%s was taken on another version of the pass kernel" % os.path.relpath(path, ROOT)
                continue
            return d["hbm_bytes_per_launch"], {"file": os.path.relpath(path, ROOT), "commit": d.get("commit"),
                                               "kernel_source_sha": sha}
        except (OSError, KeyError, ValueError):
            continue
    return None, why


def code_sha(path):
    """hash of a source file's CODE: // comments, trailing blanks and empty lines do not count (tools/pmc_asm.py has the twin)"""
    import re
    lines = [re.sub(r"//.*$", "", l).rstrip() for l in open(path, encoding="utf-8", errors="replace").read().split("\n")]
    return hashlib.sha256("\n".join(l for l in lines if l).encode()).hexdigest()[:16]


def assembly_issue_profile(N, world):
    """what bounds the assembly kernel (it is issue-bound, not HBM-bound: SURVEY.md 8d): instruction counters of k_ff_tiles from
    committed rocprofv3 PMC passes (tools/pmc_asm3.sh) -- only when taken on this workload and on the kernel source as it is
    now; otherwise a one-line reason."""
    sha = code_sha(os.path.join(ROOT, "daisyriot_amd", "csrc", "geom_kernels.hip"))
    why = "no committed counter passes for this workload"
    if world != 1:
        return why
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", rnd, "pmc_asm_64k.json")
        try:
            d = json.load(open(path))
            if d["workload"]["patches"] != N:
                continue
            if d.get("kernel_source_sha") != sha:
                why = "%s was taken on another version of geom_kernels.hip" % os.path.relpath(path, ROOT)
                continue
            return {"bound": "vector instruction issue", "source": os.path.relpath(path, ROOT), "kernel_source_sha": sha,
                    "valu_pipe_busy": d["valu_pipe_busy"], "clock_GHz": d["clock_GHz"],
                    "valu_instructions_per_traced_pair": d["per_traced_pair"]["valu_instructions"],
                    "scalar_instructions_per_traced_pair": d["per_traced_pair"]["salu_and_branch_instructions"],
                    "formula": d["valu_pipe_busy_formula"]}
        except (OSError, KeyError, ValueError):
            continue
    return why


def bare_read_gbs():
    """the best a kernel that only reads 17.18 GB reaches on this part (tools/hbm_probe.hip), from profiles/"""
    for rnd in PROFILE_ROUNDS:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", rnd, "hbm_probe.json")))
            return float(d["best_GBs"]), "profiles/%s/hbm_probe.json" % rnd
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def device_report(torch, index):
    """name, size and clocks of the GPU the numbers were taken on.  Clocks: the driver's sysfs tables of the card at the device's
    PCI address (current level starred) -- read as files: no rocm-smi child process (a process that has initialised the GPU must
    not exec, and under rocprofv3 every child has)."""
    p = torch.cuda.get_device_properties(index)
    d = {"name": p.name, "compute_units": p.multi_processor_count, "memory_GB": round(p.total_memory / 1e9, 1),
         "arch": getattr(p, "gcnArchName", None)}
    try:
        addr = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        base = os.path.join("/sys/bus/pci/devices", addr)
        clocks = {}
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
            levels = [l.strip() for l in open(os.path.join(base, name)).read().strip().split("\n")]
            cur = [l for l in levels if l.endswith("*")]
            clocks[name[7:]] = {"now": cur[0].split(":")[1].strip(" *") if cur else None, "max": levels[-1].split(":")[1].strip(" *")}
        d["pci"] = addr
        d["clocks"] = clocks
    except Exception as e:                                      # no such files in this container: the name is what matters
        d["clocks"] = "unavailable (%s)" % type(e).__name__
    return d


def strided_rows(N, n):
    """n rows spread over the whole matrix (every wall and both inner boxes of the Cornell scene), not the first few"""
    step = N / float(n)
    return sorted(set(int((k + 0.37) * step) % N for k in range(n)))


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
    # assembly: rows spread over the whole matrix (floor, ceiling, walls, both inner boxes), one thread
    m = ob.Mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
    rows = strided_rows(N, n_rows_ff)
    t = time.perf_counter()
    ob.assemble_row_list(m, uv, rows, threads=1)
    dt2 = time.perf_counter() - t
    ff = {"value": len(rows) * (N - 1) / dt2, "unit": "pairs/s", "cores": 1, "kind": "port",
          "sample": "%d rows spread over the matrix (%s) x %d columns, integrand + %d rays per facing pair through the "
                    "oracle's BVH (built once for the sample: included), %.1f s" % (len(rows), ",".join(map(str, rows)), N, uv.shape[0], dt2)}
    # the same two samples with OpenMP over rows on every host core (the reference itself has no threading)
    # (a one-GPU box grants about 16 host CPUs however many the machine has: more threads only thrash)
    cores = max(1, min(ob.num_threads(), len(os.sched_getaffinity(0)), 16))
    B = np.zeros((n1, S), np.float32)
    t = time.perf_counter()
    ob.sweep_rows(F, sc.M, sc.mat_of_patch, R, B, row0=r0, threads=cores)
    dt3 = time.perf_counter() - t
    blocks = strided_rows(N, 4)
    per = max(1, cores // 4)
    t = time.perf_counter()
    for r in blocks:
        ob.assemble_rows(m, uv, row0=min(r, N - per), nrows=per, want_vis=False, threads=cores, bvh=True)
    dt4 = time.perf_counter() - t
    allcores = {"cores": cores, "kind": "port", "cpu": cpu_model, "iters_per_s": 1.0 / (dt3 * (N / n1)),
                "pairs_per_s": len(blocks) * per * (N - 1) / dt4,
                "sample": "same sweep rows (%d); assembly: 4 blocks of %d rows spread over the matrix, OpenMP over rows, "
                          "%.1f + %.1f s" % (n1, per, dt3, dt4)}
    return sweep, ff, allcores


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """--gpus N without a torchrun environment: start the N ranks as children of this process -- which has not touched
    a GPU (counting devices does not) -- and relay rank 0's JSON line.  Never re-executes a process that used the GPU."""
    import torch

    have = torch.cuda.device_count()
    fake = None
    if args.shared_gpu_rehearsal:
        fake_dir = os.path.join(ROOT, "tests", "fake_rccl")
        subprocess.check_call(["make", "-C", fake_dir], stdout=subprocess.DEVNULL)
        fake = os.path.join(fake_dir, "libfake_rccl.so")
    elif have < args.gpus:
        sys.exit("bench.py --gpus %d: this box has %d GPU(s); the multi-process path needs one GPU per rank "
                 "(rehearse the single-process path on one GPU with: --group --gpus %d --devices %s)"
                 % (args.gpus, have, args.gpus, ",".join(["0"] * args.gpus)))
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if fake else r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if fake:
            env["DR_RCCL_LIB"] = fake
            env.setdefault("FAKE_RCCL_SLOT_MB", "64")          # mailbox size of the stand-in: >= one all-to-all block
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0:
                rc = rc or code
                for q in alive:          # a rank died: its peers would wait in a collective for ever
                    q.terminate()
    sys.exit(rc)


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
    ap.add_argument("--no-converge-report", action="store_true",
                    help="skip the extra passes through dr_solver_converge (keeps a kernel profile of this run to plain passes)")
    ap.add_argument("--rehearse-comm", action="store_true",
                    help="single rank only: still create the torch process group and the library's RCCL communicator "
                         "(all-gather of one chunk per pass), to rehearse the multi-GPU code path on a one-GPU box")
    ap.add_argument("--shared-gpu-rehearsal", action="store_true",
                    help="NOT a measurement: run the --gpus N ranks as N processes on GPU 0, torch.distributed over gloo and the "
                         "library's collectives over tests/fake_rccl (real RCCL refuses several ranks on one device) -- a rehearsal "
                         "of the multi-process code path on a one-GPU box")
    ap.add_argument("--group", action="store_true",
                    help="one process drives all --gpus devices through the library's dr_group (no torch.distributed)")
    ap.add_argument("--group-exchange", choices=["p2p", "rccl", "inpass"], default=None,
                    help="--group: how the residual travels after a pass (dr_options::group_exchange; default: RCCL between distinct "
                         "devices, peer copies between ranks that share one)")
    ap.add_argument("--devices", type=str, default=None,
                    help="with --group: the HIP ordinals, e.g. 0,1,2,3 (the same ordinal several times rehearses on one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and not args.group and "RANK" not in os.environ:
        launch_ranks(args, sys.argv[1:])

    # stdout carries exactly one JSON line: libraries that print there on their own (RCCL's version banner at
    # communicator creation, for one) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    from daisyriot_amd import api, scenes

    if args.group:
        return run_group(args, json_fd)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback for the hot path)")
    torch.cuda.set_device(local_rank)
    use_comm = world > 1 or args.rehearse_comm
    rehearsal = args.shared_gpu_rehearsal and world > 1
    tdev = "cpu" if rehearsal else "cuda"          # where torch.distributed's own tensors live (gloo in the rehearsal)
    if use_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def barrier():
        if use_comm:
            dist.barrier()

    def gather_f64(vals):
        """every rank's values -> [world][len(vals)] on every rank"""
        t = torch.tensor(vals, dtype=torch.float64, device=tdev)
        if not use_comm:
            return [list(map(float, t))]
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [list(map(float, o)) for o in out]

    sc = scenes.cornell_box(args.patches, S=args.bins)
    uv = scenes.visibility_samples(args.rays)
    E = sc.emission(7.0)
    N, S = sc.N, sc.S

    ctx = api.Context(local_rank)
    ctx.set_shard(rank, world)
    if use_comm:
        idt = torch.zeros(128, dtype=torch.uint8, device=tdev)
        if rank == 0:
            idt.copy_(torch.from_numpy(api.comm_unique_id()))
        dist.broadcast(idt, 0)
        ctx.comm_init(idt.cpu().numpy(), rank, world)
    rccl_rank, rccl_world = ctx.comm_info()
    ctx.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)

    # ---- form-factor assembly (once; produces the F the passes stream) --------------------
    # the F shard is allocated before the clock starts: a first hipMalloc of 17 GB is 0.5 - 1 s of driver time on this pool
    # (tools/malloc_time.py), reported beside the assembly as F_alloc_seconds -- a property of the host, not of the path
    t0 = time.perf_counter()
    ctx.reserve()
    t_alloc = time.perf_counter() - t0
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.assemble(uv, rule=api.RULE_INTEGRAND)
    torch.cuda.synchronize()
    barrier()
    t_asm = time.perf_counter() - t0
    info = ctx.info()
    per = gather_f64([t_asm, info.last_assemble_ms / 1e3, float(info.pairs_traced), info.last_bvh_ms / 1e3])
    t_asm = max(p[0] for p in per)
    t_asm_kernel = max(p[1] for p in per)
    pairs_traced = sum(p[2] for p in per)
    t_bvh = max(p[3] for p in per)

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
    per = gather_f64([dt, info.sweep_ms_total / max(1, info.sweep_launches), float(info.nrows), float(rccl_world), float(rccl_rank)])
    dt = max(p[0] for p in per)
    resid = ctx.step(0, want_sum=True)

    # ---- the same passes through dr_solver_converge: the convergence test fused into the pass (sums in the residual's
    # tails, decision on the device, the host looks once per 8 queued passes) -- reported beside the headline
    converge_report = None
    if not args.no_converge_report:
        ctx.reset()
        ctx.synchronize()
        barrier()
        t0 = time.perf_counter()
        it = ctx.converge(-1.0, per_bin=False, max_iters=args.steps)
        ctx.synchronize()
        barrier()
        dt_conv = max(p[0] for p in gather_f64([time.perf_counter() - t0]))
        converge_report = {"iters_per_s": it / dt_conv, "ms_per_pass": dt_conv / max(it, 1) * 1e3, "passes": it,
                           "note": "dr_solver_converge with a threshold that is never met: %d passes queued in batches of 8, "
                                   "convergence sums formed inside the pass and gathered with the residual" % it}

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
        ps = gather_f64([dt_skip, info_skip.sweep_ms_total / max(1, info_skip.sweep_launches)])
        ctx.skip_zero_blocks(False)
        skip_report = {"iters_per_s": args.steps / max(p[0] for p in ps), "kernel_ms_avg": max(p[1] for p in ps),
                       "blocks_nonzero": int(info_skip.blocks_nonzero), "blocks_total": int(info_skip.blocks_total),
                       "F_bytes_read_per_launch": int(info_skip.blocks_nonzero) * 32 * 256 * 4,
                       "note": "rank 0's shard; same F, same passes, results bit-identical to the dense pass"}

    if rank == 0:
        n_mat = sc.M.shape[0]
        # the roofline figure is per rank: that rank's shard bytes over that rank's kernel time; the line carries the
        # slowest rank (they all run the same kernel on equal shards; the last shard may be shorter)
        per_rank = [{"rank": r, "rows": int(p[2]), "kernel_ms_avg": p[1],
                     "GBs": sweep_bytes(int(p[2]), N, S, n_mat) / (p[1] * 1e-3) / 1e9 if p[1] > 0 and p[2] > 0 else None}
                    for r, p in enumerate(per)]
        slow = min((q for q in per_rank if q["GBs"]), key=lambda q: q["GBs"])
        alg = sweep_bytes(slow["rows"], N, S, n_mat)
        achieved, kern_ms = slow["GBs"], slow["kernel_ms_avg"]
        traffic, traffic_source = pmc_traffic(N, S, world)
        bare, bare_src = bare_read_gbs()
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
                       "F_bytes_per_gpu": int(info.bytes_F), "rows_per_gpu": [q["rows"] for q in per_rank],
                       "launch": "one process per GPU (torch.distributed / RCCL)" if not rehearsal else
                                 "REHEARSAL, not a measurement: %d processes share GPU 0, gloo + tests/fake_rccl" % world},
            "rehearsal_on_shared_gpu": bool(rehearsal),
            "device": device_report(torch, local_rank),
            # the options this run ran with (dr_options: defaults + DR_* environment) and what the library made of them
            "options": ctx.options().as_dict(),
            "used": {"tree": {1: "lbvh", 2: "sah"}.get(info.tree_used), "tree_built_on": "host" if info.tree_on_host else "device",
                     "tree_depth": int(info.tree_depth), "walk": {1: "threaded", 2: "pairs", 3: "paths"}.get(info.walk_used),
                     "sweep_ksplit": int(info.sweep_ksplit)},
            # what every rank's RCCL communicator itself reports (ncclCommCount / ncclCommUserRank): 0 = no communicator
            "rccl_world": int(per[0][3]), "rccl_ranks": [int(p[4]) for p in per],
            "exchange_us_per_pass": (dt / args.steps * 1e3 - max(p[1] for p in per)) * 1e3,
            "roofline": {"bound": "hbm", "kernel": "k_sweep", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": alg, "kernel_ms_avg": kern_ms,
                         "bare_read_GBs": bare, "bare_read_source": bare_src,
                         "frac_of_bare_read": achieved / bare if bare else None,
                         "launches_timed": int(info.sweep_launches), "per_rank": per_rank,
                         "note": "per rank: that rank's algorithmic shard bytes / its hipEvent-timed kernel; slowest rank shown"},
            # seconds = the BVH build (inside dr_scene_set_mesh; all of it on the device) + dr_formfactors_assemble, as
            # SURVEY.md 8(d) defines t_assemble
            "formfactors": {"value": N * (N - 1) / (t_asm + t_bvh), "unit": "pairs/s", "seconds": t_asm + t_bvh,
                            "kernel_seconds": t_asm_kernel, "bvh_build_seconds": t_bvh, "F_alloc_seconds": t_alloc,
                            "pairs_traced": pairs_traced, "rays_per_s": pairs_traced * args.rays / (t_asm + t_bvh),
                            "issue_profile": assembly_issue_profile(N, world)},
            "residual_sum_after_timed_passes": resid,
            "converge_mode": converge_report,
            # optional dr_solver_skip_zero_blocks: all-zero 32 x 256 blocks of F are not read (bit-identical results);
            # not part of "value" -- the headline streams the whole dense matrix
            "zero_block_skipping": skip_report,
        }
        if world == 1 and not args.no_cpu_baseline:
            sweep_cpu, ff_cpu, allcores = cpu_baselines(ctx, sc, uv, 65536, 8)
            out["cpu_baseline"] = sweep_cpu
            out["cpu_baseline_formfactors"] = ff_cpu
            out["cpu_baseline_all_cores"] = allcores
            out["formfactors"]["vs_cpu_port"] = out["formfactors"]["value"] / ff_cpu["value"]
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ctx.close()
    if use_comm:
        dist.destroy_process_group()


def run_group(args, json_fd):
    """one process, all GPUs: the same workload through dr_group (the C-ABI's single-process multi-GPU path)"""
    import numpy as np
    import torch

    from daisyriot_amd import api, scenes

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback for the hot path)")
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devices) != args.gpus:
        sys.exit("--devices lists %d devices, --gpus says %d" % (len(devices), args.gpus))
    if max(devices) >= torch.cuda.device_count():
        sys.exit("--devices %s: this box has %d GPU(s)" % (args.devices, torch.cuda.device_count()))
    sc = scenes.cornell_box(args.patches, S=args.bins)
    uv = scenes.visibility_samples(args.rays)
    E = sc.emission(7.0)
    N, S = sc.N, sc.S
    g = api.Group(devices)
    if args.group_exchange:
        g.set_options(group_exchange={"p2p": api.GROUP_EXCHANGE_P2P, "rccl": api.GROUP_EXCHANGE_RCCL, "inpass": api.GROUP_EXCHANGE_INPASS}[args.group_exchange])
    g.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
    t0 = time.perf_counter()
    g.assemble(uv, rule=api.RULE_INTEGRAND)
    t_asm = time.perf_counter() - t0
    infos = [c.info() for c in g.ranks]
    pairs_traced = sum(int(i.pairs_traced) for i in infos)
    g.solver_init(E, sc.M, sc.mat_of_patch)
    g.step(args.warmup)
    for c in g.ranks:
        c.profile(True)
        c.profile_reset()
    t0 = time.perf_counter()
    g.step(args.steps)                       # queues args.steps passes on every device, one wait at the end
    dt = time.perf_counter() - t0
    infos = [c.info() for c in g.ranks]
    for c in g.ranks:
        c.profile(False)
    resid = g.step(0, want_sum=True)
    n_mat = sc.M.shape[0]
    per_rank = []
    for r, i in enumerate(infos):
        ms = i.sweep_ms_total / max(1, i.sweep_launches)
        per_rank.append({"rank": r, "device": devices[r], "rows": int(i.nrows), "kernel_ms_avg": ms,
                         "GBs": sweep_bytes(int(i.nrows), N, S, n_mat) / (ms * 1e-3) / 1e9 if ms > 0 and i.nrows > 0 else None})
    slow = min((q for q in per_rank if q["GBs"]), key=lambda q: q["GBs"])
    shared = len(set(devices)) < len(devices)
    out = {
        "metric": "radiosity iters/s at N=%dk patches (patch-pair form-factors/s in \"formfactors\")" % (N // 1024),
        "value": args.steps / dt, "unit": "iters/s", "n_gpus": len(set(devices)), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "subdivided Cornell box, %d patches, %d spectral bins, K=%d rays/pair, dense fp32 F row-sharded "
                               "over %d rank(s) of ONE process (dr_group)" % (N, S, args.rays, len(devices)),
                   "patches": N, "bins": S, "rays_per_pair": args.rays, "rows_per_gpu": [q["rows"] for q in per_rank],
                   "devices": devices, "launch": "one process, dr_group; exchange: %s" % ("RCCL (ncclCommInitAll)" if g.uses_rccl() else
                                                                                         ("in the pass (peer stores + gate kernel)" if args.group_exchange == "inpass" else "peer copies")),
                   "rehearsal_on_shared_gpu": shared},
        "rccl_world": len(devices) if g.uses_rccl() else 0,
        "device": device_report(torch, devices[0]),
        "options": g.ranks[0].options().as_dict(),
        # what a pass costs beyond its kernel: wall time per step minus the ranks' kernel time (ranks that share a GPU run one
        # after the other: the sum; ranks on their own GPUs side by side: the slowest)
        "exchange_us_per_pass": (dt / args.steps * 1e3 - (sum(q["kernel_ms_avg"] for q in per_rank) if shared else max(q["kernel_ms_avg"] for q in per_rank))) * 1e3,
        "roofline": {"bound": "hbm", "kernel": "k_sweep", "achieved": slow["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": slow["GBs"] / HBM_PEAK_GBS, "traffic": None,
                     "traffic_source": "not collected for the group path",
                     "algorithmic_bytes_per_launch": sweep_bytes(slow["rows"], N, S, n_mat), "kernel_ms_avg": slow["kernel_ms_avg"],
                     "per_rank": per_rank,
                     "note": "ranks that share one GPU (rehearsal) also share its HBM: per-rank figures are not chip figures" if shared else "per rank"},
        "formfactors": {"value": N * (N - 1) / t_asm, "unit": "pairs/s", "seconds": t_asm, "pairs_traced": pairs_traced,
                        "rays_per_s": pairs_traced * args.rays / t_asm},
        "residual_sum_after_timed_passes": resid,
    }
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    g.close()


if __name__ == "__main__":
    main()
