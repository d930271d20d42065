#!/bin/bash
# one emulated strip (bench.py --emulate-strip R/N) per differently-built library: frame time and kernel table
# usage: scripts/ab_strip_libs.sh <outdir> <R/N> lib1.so ... ("default" = the shipped one)
OUT=$1; STRIP=$2; shift 2
mkdir -p "$OUT"
for LIB in "$@"; do
  TAG=$(basename "$LIB" .so)
  if [ "$LIB" = default ]; then unset RTPT_LIB_PATH; else export RTPT_LIB_PATH="$PWD/$LIB"; fi
  timeout -k 10 300 python bench.py --emulate-strip "$STRIP" --steps 400 --warmup 40 --no-cpu-baseline $AB_ARGS > "$OUT/strip-$TAG.json" 2> "$OUT/strip-$TAG.err" || { echo "$TAG failed"; tail -3 "$OUT/strip-$TAG.err"; }
  python - "$OUT/strip-$TAG.json" "$TAG" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "strip", d["emulated_strip"], "ms/frame", d["ms_per_step"], {n: v["avg_us"] for n, v in d["kernels"].items()})
PY
done
