#!/bin/bash
# SQ counters of the chained a-trous launches, per workgroup shape (pair (1,2) / (3,4)).  usage: scripts/pmc_chain.sh <tag> [ENV=..]
TAG=${1:-pmc_chain}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in "$@"; do export "$v"; done
i=0
for G in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/p$i -- python3 bench.py --workload ${WL:-4k} --steps 6 --warmup 2 --prewarm-seconds 0 --no-cpu-baseline --no-secondary > $OUT/p$i.json 2> $OUT/p$i.err || { echo pass $i failed; tail -3 $OUT/p$i.err; }
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "atrous_chain" not in r["Kernel_Name"]:
            continue
        k = ("sw" if "chain_sw" in r["Kernel_Name"] else "v1") + " wg" + r["Workgroup_Size"] + " grid" + r["Grid_Size"]
        c = acc[k][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"]); c[1] += 1
res = {k: {n: v[0] / v[1] for n, v in cs.items()} for k, cs in acc.items()}
for k, d in res.items():
    w = d.get("SQ_WAVES", 1)
    d["valu_per_wave"] = d.get("SQ_INSTS_VALU", 0) / w
    d["lds_per_wave"] = d.get("SQ_INSTS_LDS", 0) / w
    d["salu_per_wave"] = d.get("SQ_INSTS_SALU", 0) / w
print(json.dumps(res, indent=1, sort_keys=True))
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
PY
rm -rf $OUT/p1 $OUT/p2
