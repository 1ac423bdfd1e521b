#!/bin/bash
# soak of the visibility definition over every tree and walk: tools/fuzz_visibility.py with the options set through the environment
# (DR_BVH, DR_WALK, DR_OCTANT are the defaults of the contexts it creates); 0 mismatching ray counts expected in every line
cd $GRAFT_REPO_ROOT
n=${1:-400}
seed=${SEED:-20000}
for cfg in "DR_BVH=sah" "DR_BVH=sah DR_WALK=paths" "DR_BVH=sah DR_WALK=threaded" "DR_BVH=lbvh DR_WALK=pairs" "DR_BVH=sah DR_SAH_HOST=1" "DR_BVH=sah DR_OCTANT=0" "DR_BVH=sah FUZZ_SCALE_EXP=-8,8"; do
  env $cfg timeout -k 10 600 python tools/fuzz_visibility.py $n $seed > gpurun_out/fuzz_soak_$seed.log 2>&1
  echo "$cfg: $(tail -1 gpurun_out/fuzz_soak_$seed.log)"
  seed=$((seed + n))
done
