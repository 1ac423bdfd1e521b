"""Turns the raw output of tools/pmc_sweep.sh into the files bench.py and the judge read:
pmc_sweep_64k.json (HBM bytes per k_sweep launch, with the hash of the kernel source it was taken on),
pmc_summary.md, bench64k_kernel_stats.csv, hbm_probe.json."""
import csv, glob, hashlib, json, os, re, subprocess, sys
from collections import defaultdict

out = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha():
    # the same function as bench.py's kernel_source_sha
    h = hashlib.sha256()
    csrc = os.path.join(root, "daisyriot_amd", "csrc")
    h.update(open(os.path.join(csrc, "sweep_kernels.hip"), "rb").read())
    m = re.search(r"struct SweepParams \{.*?\n\};", open(os.path.join(csrc, "dr_internal.h")).read(), flags=re.S)
    h.update((m.group(0) if m else "").encode())
    return h.hexdigest()[:16]


vals = defaultdict(lambda: defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                vals[r["Kernel_Name"]][ctr].append(float(r["Counter_Value"]))
sweep = [k for k in vals if "k_sweep" in k and "mask" not in k]
res = {"workload": {"patches": 65536, "bins": 8, "world": 1}, "kernel_source_sha": sha(),
       "correction": "gfx950: FETCH_SIZE reports half the bytes of a wide coalesced stream (x2); WRITE_SIZE exact; units KiB "
                     "(MI355X_MICROARCH.md, HBM)",
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace --output-format csv -- python3 bench.py "
                  "--no-cpu-baseline --no-zero-block-report --no-converge-report --steps 6 --warmup 2"}
res["commit"] = os.environ.get("DR_HEAD")
if sweep:
    k = sweep[0]
    f = sum(vals[k]["FETCH_SIZE"]) / max(1, len(vals[k]["FETCH_SIZE"]))
    w = sum(vals[k]["WRITE_SIZE"]) / max(1, len(vals[k]["WRITE_SIZE"]))
    res.update({"kernel": k, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "dispatches": len(vals[k]["FETCH_SIZE"]),
                "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024,
                "algorithmic_bytes_per_launch": 4 * 65536 * 65536 + 4 * 65536 * 8 + 12 * 65536 * 8 + 4 * 4 * 64})
res["all_kernels"] = {k: {c + "_KB_mean": sum(v) / len(v) for c, v in d.items()} | {c + "_n": len(v) for c, v in d.items()}
                      for k, d in vals.items()}
json.dump(res, open(os.path.join(out, "pmc_sweep_64k.json"), "w"), indent=1)
with open(os.path.join(out, "pmc_summary.md"), "w") as fh:
    fh.write("# PMC passes, bench.py at N=65536 S=8 on 1 MI355X (round 3, kernel source %s)\n\n" % res["kernel_source_sha"])
    fh.write("Separate `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs. Per-dispatch means in KiB as rocprofv3 reports them.\n\n")
    fh.write("| kernel | counter | dispatches | mean (KiB) |\n|---|---|---|---|\n")
    for k, d in vals.items():
        for c, v in d.items():
            fh.write("| `%s` | %s | %d | %g |\n" % (k[:70], c, len(v), sum(v) / len(v)))
    if sweep:
        fh.write("\n`k_sweep`: 2 x FETCH_SIZE + WRITE_SIZE = %.4f GB per launch against %.4f GB algorithmic (ratio %.4f).\n"
                 % (res["hbm_bytes_per_launch"] / 1e9, res["algorithmic_bytes_per_launch"] / 1e9,
                    res["hbm_bytes_per_launch"] / res["algorithmic_bytes_per_launch"]))
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    open(os.path.join(out, "bench64k_kernel_stats.csv"), "w").write(open(f).read())
best = 0.0
try:
    for line in open(os.path.join(out, "hbm_probe.log")):
        m = re.search(r"([0-9.]+) GB/s", line)
        if m:
            best = max(best, float(m.group(1)))
    json.dump({"best_GBs": best, "bytes": 17.18e9, "tool": "tools/hbm_probe.hip 16 (bare streaming read, best of its grid shapes)"},
              open(os.path.join(out, "hbm_probe.json"), "w"))
except OSError:
    pass
print("sweep kernel:", sweep[:1], "bytes/launch", res.get("hbm_bytes_per_launch"), "bare read", best)
