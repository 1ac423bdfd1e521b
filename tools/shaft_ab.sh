#!/bin/bash
# A/B of the tile-pair shaft culling in k_ff_tiles: visit counts (DR_TILE_STATS) and kernel time, shaft on/off
cd $GRAFT_REPO_ROOT
N=${1:-16384}
for sh in 0 1; do
  echo "== DR_SHAFT=$sh stats"; NPATCH=$N DR_SHAFT=$sh DR_TILE_STATS=1 timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep -E "daisyriot|RES"
  echo "== DR_SHAFT=$sh time";  NPATCH=$N DR_SHAFT=$sh timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep -E "RES"
done
