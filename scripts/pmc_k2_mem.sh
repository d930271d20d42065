#!/bin/bash
# memory-side counters of the K2 kernels (one rocprofv3 --pmc pass per group, kernel-trace only), per launch position
OUT=${1:-gpurun_out/pmc_k2_mem}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
for G in "FETCH_SIZE" "WRITE_SIZE" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITEBACK_sum" \
  "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $G --kernel-include-regex "pathtrace" --output-format csv -d "$OUT/pass$i" -- python3 bench.py --steps 6 --warmup 2 --prewarm-seconds 0 --no-cpu-baseline --no-secondary $* > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || { echo "pass $i failed"; tail -3 "$OUT/pass$i.err"; }
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, re, sys
out = {}
for f in sorted(glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    names = {int(r["Dispatch_Id"]): re.search(r"(k_\w+)", r["Kernel_Name"]).group(1) for r in rows}
    # position inside the frame: count launches since the last tile-kernel launch
    pos, p = {}, 0
    for d in ids:
        p = 0 if ("binned" not in names[d] and "queue" not in names[d]) else p + 1
        pos[d] = p
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        d = int(r["Dispatch_Id"])
        acc["launch%d:%s" % (pos[d], names[d])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        out.setdefault(k, {}).update({c: round(sum(v) / len(v), 1) for c, v in cs.items()})
print(json.dumps(out, indent=1, sort_keys=True))
PY
