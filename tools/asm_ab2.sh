#!/bin/bash
# kernel ms of the default tile kernel at 16k and 64k patches (two runs each) + the counted build's visits at 16k
cd $GRAFT_REPO_ROOT
for n in 16384 65536; do
  for rep in 1 2; do NPATCH=$n timeout -k 10 120 python3 tools/asm_one.py 2>&1; done
done
DR_TILE_STATS=1 NPATCH=16384 timeout -k 10 200 python3 tools/asm_one.py 2>&1
