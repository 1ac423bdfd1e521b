#!/bin/bash
# A/B of the tile kernel's walks: threaded tree (round 2), sibling-pair records from the root, path records + pair walk (default);
# kernel ms at 16k and 64k patches, node visits per pair from the counted (DR_TILE_STATS) build at 16k
cd $GRAFT_REPO_ROOT
for w in ${WALKS:-threaded pairs paths}; do
  for n in 16384 65536; do
    for rep in 1 2; do DR_WALK=$w NPATCH=$n timeout -k 10 120 python3 tools/asm_one.py 2>&1 | grep RES | sed "s/^/$w /"; done
  done
  DR_WALK=$w DR_TILE_STATS=1 NPATCH=16384 timeout -k 10 200 python3 tools/asm_one.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$w stats /"
done
