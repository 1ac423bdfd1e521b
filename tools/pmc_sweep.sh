#!/bin/bash
# HBM traffic of the pass kernel at HEAD: separate rocprofv3 --pmc passes (no tracing domains beside --kernel-trace),
# gfx950 corrections applied by tools/pmc_sweep.py; plus the kernel-trace stats of the same command and the bare-read probe.
# Writes gpurun_out/r03/* -- copy what should be judged into profiles/r03/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03
rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --no-cpu-baseline --no-zero-block-report --no-converge-report --steps 6 --warmup 2"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$ctr -o c -- $CMD > $OUT/pmc_$ctr.json 2> $OUT/pmc_$ctr.err || { echo "$ctr pass failed"; tail -5 $OUT/pmc_$ctr.err; }
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --no-cpu-baseline --no-zero-block-report --no-converge-report > $OUT/bench64k_under_rocprof.json 2> $OUT/stats.err || { echo "stats pass failed"; tail -5 $OUT/stats.err; }
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/hbm_probe tools/hbm_probe.hip 2>/dev/null
timeout -k 10 300 ./tools/hbm_probe 16 > $OUT/hbm_probe.log 2>&1
python3 tools/pmc_sweep.py $OUT
