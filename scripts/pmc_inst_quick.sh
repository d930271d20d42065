#!/bin/bash
TAG=${1:-pmc_instq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/pass$i" -- python3 bench.py --workload instanced --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || echo "pass $i failed"
done
python3 scripts/summarize_pmc.py "$OUT" > "$OUT/summary.json"
python3 - "$OUT/summary.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in ("k_pathtrace","k_gbuffer"):
    v=d.get(k,{}); print(k,{n: float('%.4g'%x) for n,x in sorted(v.items()) if isinstance(x,(int,float))})
PY
