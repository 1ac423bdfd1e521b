#!/bin/bash
# per-patch path records (DR_PATHS=1, default) against the walk from the root (DR_PATHS=0): visits per pair and kernel ms
cd $GRAFT_REPO_ROOT
for n in ${SIZES:-16384 65536}; do
  for pt in 0 1; do
    [ $n -le 16384 ] && NPATCH=$n DR_PATHS=$pt DR_TILE_STATS=1 timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep -E "daisyriot" | sed "s/^/paths=$pt /"
    NPATCH=$n DR_PATHS=$pt timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep RES | sed "s/^/paths=$pt /"
  done
done
