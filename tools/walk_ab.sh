#!/bin/bash
# the three walks of k_ff_tiles: from the root (shipped), tile-pair shaft lists (DR_SHAFT=1), per-patch path records (DR_PATHS=1)
cd $GRAFT_REPO_ROOT
for n in ${SIZES:-16384 65536}; do
  for mode in "DR_SHAFT=0 DR_PATHS=0" "DR_SHAFT=1" "DR_PATHS=1"; do
    [ $n -le 16384 ] && env $mode NPATCH=$n DR_TILE_STATS=1 timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep -E "daisyriot" | sed "s/^/$mode /"
    env $mode NPATCH=$n timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep RES | sed "s/^/$mode /"
  done
done
