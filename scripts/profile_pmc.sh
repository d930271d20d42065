#!/bin/bash
# PMC passes for the kernels of one bench workload (run on the GPU box through gpurun).
# (a TA_*/GRBM_* group hung rocprofv3 on this pool and is not collected.)
# One counter group per rocprofv3 invocation (FETCH_SIZE / WRITE_SIZE cannot share a pass,
# MI355X_MICROARCH.md "rocprofv3 PMC slots"); kernel-trace only, no other tracing domains.
# usage: scripts/profile_pmc.sh <tag> [bench args...]
set -e
TAG=${1:-pmc}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary $*"
i=0
for G in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/pass$i" -- python3 bench.py $ARGS > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || { echo "pass $i failed"; tail -5 "$OUT/pass$i.err"; }
  echo "pass $i done: $G"
done
python3 scripts/summarize_pmc.py "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json" | head -c 4000
