#!/bin/bash
# A/B of differently-built libraries on the GPU box: bench.py per library (RTPT_LIB_PATH), kernel table only.
# usage: scripts/ab_libs.sh <outdir> <workload> lib1.so lib2.so ...   (paths relative to the repo root; "default" = the shipped one)
OUT=$1; WL=$2; shift 2
mkdir -p "$OUT"
for LIB in "$@"; do
  TAG=$(basename "$LIB" .so)
  if [ "$LIB" = default ]; then unset RTPT_LIB_PATH; else export RTPT_LIB_PATH="$PWD/$LIB"; fi
  timeout -k 10 300 python bench.py --workload "$WL" --steps 100 --warmup 10 --no-cpu-baseline --no-secondary $AB_ARGS > "$OUT/$WL-$TAG.json" 2> "$OUT/$WL-$TAG.err" || { echo "$TAG failed"; tail -3 "$OUT/$WL-$TAG.err"; }
  python - "$OUT/$WL-$TAG.json" "$TAG" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["kernels"]
print(sys.argv[2], d["config"]["workload"], "ms/frame", d["ms_per_step"], {n: v["avg_us"] for n, v in k.items()})
PY
done
