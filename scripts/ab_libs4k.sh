#!/bin/bash
OUT=$1; shift
mkdir -p $OUT
for LIB in "$@"; do
  TAG=$(basename "$LIB" .so)
  if [ "$LIB" = default ]; then unset RTPT_LIB_PATH; else export RTPT_LIB_PATH="$PWD/$LIB"; fi
  scripts/ab_flags.sh $OUT/$TAG "4k 0" | sed "s/^/$TAG /" | cut -c1-120
done
