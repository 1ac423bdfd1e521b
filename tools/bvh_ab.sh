#!/bin/bash
# the tree: Morton / Karras on the device (default) against the host's binned-SAH topology (DR_BVH=sah)
cd $GRAFT_REPO_ROOT
for n in ${SIZES:-16384 65536}; do
  for k in lbvh sah; do
    [ $n -le 16384 ] && DR_BVH=$k NPATCH=$n DR_TILE_STATS=1 timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep -E "daisyriot" | cut -c1-120 | sed "s/^/$k /"
    DR_BVH=$k NPATCH=$n timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep RES | sed "s/^/$k /"
  done
done
for k in lbvh sah; do echo $k; DR_BVH=$k timeout -k 10 200 python3 tools/ref_scene_time.py 2>&1 | grep assemble | cut -c1-100; done
