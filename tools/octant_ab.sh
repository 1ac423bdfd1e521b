#!/bin/bash
# sign-specialised node test (default) against the general one (DR_OCTANT=0): visits per pair (16k) and kernel ms
cd $GRAFT_REPO_ROOT
for n in ${SIZES:-16384 65536}; do
  for mode in "DR_OCTANT=0" "DR_OCTANT=1"; do
    [ $n -le 16384 ] && env $mode NPATCH=$n DR_TILE_STATS=1 timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep -E "daisyriot" | cut -c1-120 | sed "s/^/$mode /"
    env $mode NPATCH=$n timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep RES | sed "s/^/$mode /"
  done
done
