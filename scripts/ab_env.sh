#!/bin/bash
# generic in-frame A/B: scripts/ab_env.sh <tag> <workload> "ENV=a ENV2=b" "ENV=c" ...   (one bench run per env set)
TAG=$1; WL=$2; shift 2
out=gpurun_out/$TAG; mkdir -p $out
i=0
for e in "$@"; do
  i=$((i+1))
  env $e python bench.py --workload $WL --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $out/run$i.json 2> $out/run$i.err || { tail -3 $out/run$i.err; exit 1; }
  python - "$e" $out/run$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1] or "(default)", "| ms/frame", d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
done
