#!/bin/bash
# instruction-cache and issue-stall counters of the tile kernel (NPATCH patches, default 16384)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_icache
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES" \
           "SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_BRANCH" \
           "SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/run$i -o c -- python3 tools/asm_one.py > $OUT/run$i.log 2>&1 || { echo "group $i failed"; tail -3 $OUT/run$i.log; }
done
python3 - <<'PY'
import csv, glob, collections, re
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_icache/run*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ff_tiles" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = max(tot[r["Counter_Name"]], float(r["Counter_Value"]))
pairs = None
for f in glob.glob("gpurun_out/pmc_icache/run*.log"):
    m = re.search(r"traced (\d+)", open(f).read())
    if m: pairs = float(m.group(1))
for k in sorted(tot): print("%-24s %16.0f  per pair %10.1f" % (k, tot[k], tot[k] / pairs))
PY
