#!/bin/bash
# assembly kernel time, default build; DR_SHAFT=1 for the shaft-list variant
cd $GRAFT_REPO_ROOT
for n in 16384 65536; do
  for sh in ${SHAFTS:-0 1}; do
    NPATCH=$n DR_SHAFT=$sh timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep RES | sed "s/^/shaft=$sh /"
  done
done
