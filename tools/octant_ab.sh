#!/bin/bash
# sign-specialised node test (default) against the general one (DR_OCTANT=0)
cd $GRAFT_REPO_ROOT
for n in ${SIZES:-16384 65536}; do
  for mode in "DR_OCTANT=0" "DR_OCTANT=1"; do
    env $mode NPATCH=$n timeout -k 10 300 python3 tools/asm_one.py 2>&1 | grep RES | sed "s/^/$mode /"
  done
done
