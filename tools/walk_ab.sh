#!/bin/bash
# A/B of the two tile-kernel walks: threaded tree (round 2) against sibling-pair records; kernel ms at 16k and 64k patches,
# node visits per pair from the counted (DR_TILE_STATS) build at 16k
cd $GRAFT_REPO_ROOT
for w in threaded pairs; do
  for n in 16384 65536; do
    for rep in 1 2; do DR_WALK=$w NPATCH=$n timeout -k 10 120 python3 tools/asm_one.py 2>&1 | sed "s/^/$w /"; done
  done
  DR_WALK=$w DR_TILE_STATS=1 NPATCH=16384 timeout -k 10 200 python3 tools/asm_one.py 2>&1 | sed "s/^/$w stats /"
done
