#!/bin/bash
# compile the geometry kernels and report, for the default tile kernel: SGPR/VGPR/spill counts and the v_readlane count around the hand-written walk (must be 0: spill reloads there cost 9 %)
mkdir -p /tmp/isa; cd $(dirname $(readlink -f $0))/../daisyriot_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -save-temps=obj -c geom_kernels.hip -o /tmp/isa/geom.o 2> /tmp/isa/res.txt
grep -A12 "k_ff_tilesILi256ELb0ELi0EEE" /tmp/isa/res.txt | grep -i "TotalSGPRs\|  VGPRs:\|occupancy\|SGPRs Spill"
cd /tmp/isa
L=$(grep -n "^_ZN2dr10k_ff_tilesILi256ELb0ELi0EEEvNS_10TileParamsE:" geom_kernels-hip-amdgcn-amd-amdhsa-gfx950.s | cut -d: -f1)
awk -v L=$L 'NR>=L{print} NR>L && /s_endpgm/{exit}' geom_kernels-hip-amdgcn-amd-amdhsa-gfx950.s > tile.s
S=$(grep -n "ds_read_u16" tile.s | head -1 | cut -d: -f1); E=$(grep -n "ds_write_b8.*offset:10752" tile.s | head -1 | cut -d: -f1)
sed -n "${S},${E}p" tile.s > loop.s
A=$(grep -n "ASMSTART" loop.s | head -1 | cut -d: -f1); Z=$(grep -n "ASMEND" loop.s | head -1 | cut -d: -f1)
echo "pair loop: VALU $(grep -c '^\s*v_' loop.s) readlane/writelane $(grep -c 'v_readlane\|v_writelane' loop.s); within 12 lines before asm: $(sed -n "$((A-12)),${A}p" loop.s | grep -c v_readlane); within 20 after: $(sed -n "${Z},$((Z+20))p" loop.s | grep -c v_readlane)"
