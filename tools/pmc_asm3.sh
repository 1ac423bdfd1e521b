#!/bin/bash
# instruction, scalar-cache and wait counters of the assembly kernel (NPATCH patches, default 16384); one counter group per run.
# Writes $OUT/summary.txt and $OUT/pmc_asm.json (copy into profiles/rNN/: bench.py reports it when the kernel source matches).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_asm3
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQC_TC_REQ SQC_TC_DATA_READ_REQ SQC_DCACHE_REQ_READ_8 SQC_DCACHE_REQ_READ_16" \
           "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/run$i -o c -- python3 tools/asm_one.py > $OUT/run$i.log 2>&1 || { echo "group $i failed"; tail -5 $OUT/run$i.log; }
done
python3 tools/pmc_asm.py $OUT
