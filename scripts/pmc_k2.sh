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
import collections, csv, glob, json, re, sys
out = {}
for f in sorted(glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    names = {int(r["Dispatch_Id"]): re.search(r"(k_\w+)", r["Kernel_Name"]).group(1) for r in rows}
    tiles = [d for d in ids if "binned" not in names[d] and "queue" not in names[d]]
    per_frame = (tiles[1] - tiles[0]) if len(tiles) > 1 else len(ids)   # K2 launches per frame (only K2 kernels are in the file)
    pos = {d: (i % per_frame) for i, d in enumerate(ids)}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        d = int(r["Dispatch_Id"])
        acc["launch%d:%s" % (pos[d], names[d])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
for k, d in sorted(out.items()):
    w = d.get("SQ_WAVES", 0)
    if w:
        d["valu_per_wave"] = d.get("SQ_INSTS_VALU", 0) / w
        d["salu_per_wave"] = d.get("SQ_INSTS_SALU", 0) / w
print(json.dumps(out, indent=1, sort_keys=True))
PY
