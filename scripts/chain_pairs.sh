#!/bin/bash
# per-pair durations of the chained a-trous launches (pair (1,2) and pair (3,4) alternate inside a frame), from a rocprofv3
# kernel trace of a short bench run.  usage: scripts/chain_pairs.sh <tag> [env assignments...]
TAG=${1:-pairs}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in "$@"; do export "$v"; done
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --workload ${WL:-4k} --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "atrous_chain" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
for i, r in enumerate(rows):
    d[(i % 2, r["Kernel_Name"].split("(")[0][-40:], r.get("Workgroup_Size_Y") or r.get("Workgroup_Size"), r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v = v[len(v)//3:]
    print(k, "n", len(v), "avg_us %.2f" % (sum(v) / len(v)), "min %.2f" % min(v))
PY
rm -rf $OUT/trace
