#!/bin/bash
# bench.py per (workload, flags) pair on the GPU box, K2 kernel table only.  usage: scripts/ab_flags.sh <outdir> "4k 0x400" "1080p 0" ...
OUT=$1; shift
mkdir -p "$OUT"
for CFG in "$@"; do
  set -- $CFG
  timeout -k 10 300 python bench.py --workload "$1" --flags "$2" --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > "$OUT/$1-$2.json" 2> "$OUT/$1-$2.err" || { echo "$CFG failed"; tail -3 "$OUT/$1-$2.err"; continue; }
  python - "$OUT/$1-$2.json" "$1 $2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["kernels"]
print(sys.argv[2], "ms/frame", d["ms_per_step"], {n: v["avg_us"] for n, v in k.items()})
PY
done
