#!/bin/bash
# Same-box A/B of the assembly kernel between builds of the library (boxes differ by +-1 % among themselves: only numbers from one
# run compare).  Two steps:
#   here (no GPU):  tools/asm_ab_libs.sh build NAME [GIT_REV]   -> ab_libs/libNAME.so from geom_kernels.hip at GIT_REV (default: the working tree)
#   on the box:     gpurun -- 'bash tools/asm_ab_libs.sh run A B ...'   -> kernel ms at 16k and 64k patches, three interleaved rounds
# ab_libs/ is git-ignored; remove it afterwards (it travels with every gpurun call).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  name=$2; rev=$3
  mkdir -p $ROOT/ab_libs
  cd $ROOT/daisyriot_amd/csrc
  if [ -n "$rev" ]; then cp geom_kernels.hip /tmp/_ab_keep.hip; git show $rev:daisyriot_amd/csrc/geom_kernels.hip > geom_kernels.hip; fi
  make > /dev/null 2>&1 || { echo "build failed"; [ -n "$rev" ] && cp /tmp/_ab_keep.hip geom_kernels.hip; exit 1; }
  cp ../lib/libdaisyriot_hip.so $ROOT/ab_libs/lib$name.so
  if [ -n "$rev" ]; then cp /tmp/_ab_keep.hip geom_kernels.hip; make > /dev/null 2>&1; fi
  echo "ab_libs/lib$name.so"
elif [ "$1" = run ]; then
  shift
  cd ${GRAFT_REPO_ROOT:-$ROOT}
  for rep in 1 2 3; do for v in "$@"; do for n in ${SIZES:-16384 65536}; do
    echo -n "$v "; DR_LIB=$PWD/ab_libs/lib$v.so NPATCH=$n timeout -k 10 120 python3 tools/asm_one.py 2>&1 | grep RES
  done; done; done
else
  sed -n 2,8p "$0"
fi
