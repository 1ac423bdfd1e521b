#!/bin/bash
# per-kernel time of the BVH build (SAH on the device) at N patches: rocprofv3 kernel trace of tools/bvh_build_time.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-65536}
OUT=gpurun_out/bvh_prof
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python3 tools/bvh_build_time.py $N > $OUT/run.log 2>&1
tail -4 $OUT/run.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/b_kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print("%-70s calls %6s  total us %10.1f  avg us %8.2f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
