#!/bin/bash
# BVH sort key: plain Morton (0), orientation class + Morton (1), orientation class + Morton with the dominant axis leading (2)
cd $GRAFT_REPO_ROOT
for n in ${SIZES:-16384 65536}; do
  for k in ${KEYS:-0 1 2}; do
    [ $n -le 16384 ] && DR_BVH_KEY=$k NPATCH=$n DR_TILE_STATS=1 timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep -E "daisyriot" | cut -c1-120 | sed "s/^/key $k /"
    DR_BVH_KEY=$k NPATCH=$n timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep RES | sed "s/^/key $k /"
  done
done
