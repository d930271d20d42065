#!/bin/bash
# per-launch durations of the K2 kernels of the last traced frame (rocprofv3 --kernel-trace), one library per argument
# usage: [K2_ARGS="--flags 0x200"] scripts/k2_launches.sh <outdir> default|lib.so ...
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for LIB in "$@"; do
  TAG=$(basename "$LIB" .so)
  if [ "$LIB" = default ]; then unset RTPT_LIB_PATH; else export RTPT_LIB_PATH="$PWD/$LIB"; fi
  rm -rf "$OUT/$TAG"; mkdir -p "$OUT/$TAG"
  timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$TAG" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $K2_ARGS > "$OUT/$TAG.json" 2> "$OUT/$TAG.err" || echo "$TAG failed"
  python3 - "$OUT/$TAG" "$TAG" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pathtrace" in r["Kernel_Name"]]
per_frame = 1
for i in range(len(rows) - 1, 0, -1):   # launches per frame = distance between the last two tile-kernel launches
    if "binned" not in rows[i]["Kernel_Name"] and "queue" not in rows[i]["Kernel_Name"]:
        for j in range(i - 1, -1, -1):
            if "binned" not in rows[j]["Kernel_Name"] and "queue" not in rows[j]["Kernel_Name"]:
                per_frame = i - j
                break
        break
last = rows[-per_frame:]
print(sys.argv[2], [(re.search(r"(k_\w+)", r["Kernel_Name"]).group(1), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1)) for r in last])
PY
done
