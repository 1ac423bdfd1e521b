#!/bin/bash
cd $GRAFT_REPO_ROOT
for n in ${SIZES:-16384 65536}; do
  for mode in "DR_WALK2=0" "DR_WALK2=1"; do
    [ $n -le 16384 ] && env $mode NPATCH=$n DR_TILE_STATS=1 timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep -E "daisyriot" | cut -c1-140 | sed "s/^/$mode /"
    env $mode NPATCH=$n timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep RES | sed "s/^/$mode /"
  done
done
