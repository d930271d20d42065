#!/bin/bash
# usage: scripts/ab_libs_flags.sh <outdir> "<wl> <flags>" -- lib1 lib2 ...
OUT=$1; CFG=$2; shift 3
mkdir -p $OUT
for LIB in "$@"; do
  TAG=$(basename "$LIB" .so)
  if [ "$LIB" = default ]; then unset RTPT_LIB_PATH; else export RTPT_LIB_PATH="$PWD/$LIB"; fi
  scripts/ab_flags.sh $OUT/$TAG "$CFG" | sed "s/^/$TAG /" | cut -c1-110
done
