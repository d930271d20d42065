#!/bin/bash
# A/B of the chained a-trous kernels inside the 4K / 1080p frame: RTPT_CHAIN_SW=0 (round-2 kernel) vs 1 (sliding window)
out=${1:-gpurun_out/ab_chain}
mkdir -p $out
for wl in 4k 1080p; do
  for sw in 0 1; do
    RTPT_CHAIN_SW=$sw python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $out/${wl}_sw$sw.json 2> $out/${wl}_sw$sw.err || exit 1
    python - <<PY
import json
d=json.load(open("$out/${wl}_sw$sw.json"))
print("$wl sw=$sw ms/frame", d["ms_per_step"], {k:v["avg_us"] for k,v in d["kernels"].items()})
PY
  done
done
