#!/bin/bash
# build an A/B variant of the library next to the shipped one: scripts/build_variant.sh <name> <extra hipcc flags...>
#   -> real_time_path_tracing_with_spatiotemporal_filtering_amd/variants/librtpt_<name>.so  (load it with RTPT_LIB_PATH)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/real_time_path_tracing_with_spatiotemporal_filtering_amd/csrc
OUT=$ROOT/real_time_path_tracing_with_spatiotemporal_filtering_amd/variants
mkdir -p "$OUT" "$CS/build_$NAME"
make -s -C "$CS" OBJDIR="build_$NAME" OUT="$OUT/librtpt_$NAME.so" EXTRA="$*"
echo "$OUT/librtpt_$NAME.so"
