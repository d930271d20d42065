#!/bin/bash
# quick PMC passes over scratch/quickbench.py (4K only): usage scripts/pmc_quick.sh <tag>
TAG=${1:-pmcq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
i=0
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" ; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/pass$i" -- python3 scratch/quickbench.py 3840x2160 > "$OUT/pass$i.out" 2> "$OUT/pass$i.err" || echo "pass $i failed"
  echo "pass $i done: $G"
done
python3 scripts/summarize_pmc.py "$OUT" > "$OUT/summary.json"
python3 - "$OUT/summary.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in ("k_atrous","k_atrous_final","k_gradient"):
    v=d.get(k,{})
    print(k, {n: float('%.4g'%v[n]) for n in sorted(v) if n in ("fetch_bytes_corrected_x2","write_bytes","l2_hit_rate","valu_insts_per_wave","SQ_WAIT_INST_ANY","SQ_WAIT_ANY","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_ACTIVE_INST_VALU","TCC_EA0_RDREQ_DRAM_sum","TCC_EA0_RDREQ_sum","SQ_INSTS_VMEM_RD")})
PY
