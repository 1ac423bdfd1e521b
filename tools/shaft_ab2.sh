#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { # env...
  env "$@" DR_TILE_STATS=1 timeout -k 10 200 python3 tools/shaft_morton.py 2>&1 | grep -E "daisyriot" | sed 's/\[daisyriot\] //'
  env "$@" timeout -k 10 200 python3 tools/shaft_morton.py 2>&1 | grep RES
}
run MORTON=1 DR_SHAFT=0
for mn in 8 32 128; do for shr in 0 0.25; do run MORTON=1 DR_SHAFT=1 DR_SHAFT_MIN=$mn DR_SHAFT_SHRINK=$shr; done; done
run MORTON=0 DR_SHAFT=1 DR_SHAFT_MIN=32 DR_SHAFT_SHRINK=0.25
