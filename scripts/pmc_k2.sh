#!/bin/bash
# SQ counters of the K2 kernels only (one rocprofv3 --pmc pass per counter group, kernel-trace only)
OUT=${1:-gpurun_out/pmc_k2}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
for G in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_INST_CYCLES_SMEM" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $G --kernel-include-regex "pathtrace" --output-format csv -d "$OUT/pass$i" -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary $* > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || { echo "pass $i failed"; tail -3 "$OUT/pass$i.err"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].split("::")[-1]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    n = len(next(iter(cs.values())))
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["_dispatches"] = n
print(json.dumps(out, indent=1, sort_keys=True))
PY
