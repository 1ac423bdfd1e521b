#!/bin/bash
# the SAH builder's two parameters against the assembly time (one box): dr_options::sah_dilate (box growth in the cost, in mean
# patch diagonals) and sah_bins; kernel ms at NPATCH (default 65536), and node / leaf visits per pair from the counted build at 16k
cd $GRAFT_REPO_ROOT
N=${NPATCH:-65536}
for d in ${DILATES:-0 0.25 0.5 1 2 4}; do
  echo -n "dilate $d bins 32: "; DR_SAH_DILATE=$d NPATCH=$N timeout -k 10 120 python3 tools/asm_one.py 2>&1 | grep RES
  DR_SAH_DILATE=$d DR_TILE_STATS=1 NPATCH=16384 timeout -k 10 200 python3 tools/asm_one.py 2>&1 | grep daisyriot | cut -c1-130
done
for b in ${BINS:-8 16 64 128}; do
  echo -n "dilate 0.5 bins $b: "; DR_SAH_BINS=$b NPATCH=$N timeout -k 10 120 python3 tools/asm_one.py 2>&1 | grep RES
  DR_SAH_BINS=$b DR_TILE_STATS=1 NPATCH=16384 timeout -k 10 200 python3 tools/asm_one.py 2>&1 | grep daisyriot | cut -c1-130
done
echo -n "again dilate 0.5 bins 32: "; NPATCH=$N timeout -k 10 120 python3 tools/asm_one.py 2>&1 | grep RES
