#!/bin/bash
# Everything profiles/ holds for one state of the code, in one GPU call:
#   kernel stats (rocprofv3 --kernel-trace --stats) + the bench line of the same run, per workload
#   PMC passes (one counter group per rocprofv3 run, kernel-trace only) folded per kernel, per workload
#   the per-pass counter_collection.csv files themselves (so traffic.json can be re-derived from profiles/)
# usage: COMMIT=<hash> scripts/profile_all.sh <tag>          -> gpurun_out/<tag>/
set -u
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for WL in 4k 1080p reference instanced; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$WL" -- python3 bench.py --workload $WL --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > "$OUT/${WL}_bench_under_rocprof.json" 2> "$OUT/stats_$WL.err" || echo "stats $WL failed"
  find "$OUT/stats_$WL" -name "*kernel_stats.csv" -exec cp {} "$OUT/${WL}_kernel_stats.csv" \;
  rm -rf "$OUT/stats_$WL"
  echo "kernel stats $WL done"
done
for WL in 4k 1080p; do
  mkdir -p "$OUT/pmc_$WL"
  i=0
  for G in \
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS" \
    "FETCH_SIZE" \
    "WRITE_SIZE" \
    "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" ; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/pmc_$WL/pass$i" -- python3 bench.py --workload $WL --steps 10 --warmup 3 --prewarm-seconds 0 --no-cpu-baseline --no-secondary > "$OUT/pmc_$WL/pass$i.json" 2> "$OUT/pmc_$WL/pass$i.err" || { echo "pmc $WL pass $i failed"; tail -3 "$OUT/pmc_$WL/pass$i.err"; }
    find "$OUT/pmc_$WL/pass$i" -name "*counter_collection.csv" -exec cp {} "$OUT/pmc_${WL}_pass${i}_counter_collection.csv" \;
    echo "pmc $WL pass $i done: $G"
  done
  python3 scripts/summarize_pmc.py "$OUT/pmc_$WL" > "$OUT/pmc_${WL}_summary.json"
  rm -rf "$OUT/pmc_$WL"
  python3 scripts/fold_pmc.py "$OUT"/pmc_${WL}_pass*_counter_collection.csv
done
python3 - "$OUT" "${COMMIT:-unknown}" <<'PY'
import json, sys, datetime
out, commit = sys.argv[1], sys.argv[2]
t = {"_note": "HBM-side bytes per launch from rocprofv3 --pmc (separate passes): FETCH_SIZE*1024*2 (gfx950 tallies 128-B read "
              "requests at 64 B, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE*1024, averaged over the launches of each kernel; "
              "inputs: the pmc_<workload>_pass*_counter_collection.csv files next to this one, folded by scripts/summarize_pmc.py; "
              "valu: SQ_INSTS_VALU and SQ_INSTS_VALU_TRANS_F32 per launch (wave64 instructions) from the same passes",
     "_source": {"commit": commit, "captured": datetime.date.today().isoformat(), "tool": "scripts/profile_all.sh"}}
for wl in ("4k", "1080p"):
    try:
        s = json.load(open(f"{out}/pmc_{wl}_summary.json"))
    except Exception:
        continue
    t[wl] = {k: int(v["hbm_traffic_bytes"]) for k, v in s.items() if "hbm_traffic_bytes" in v}
    t.setdefault("valu", {})[wl] = {k: {"insts": int(v["SQ_INSTS_VALU"]), "trans": int(v.get("SQ_INSTS_VALU_TRANS_F32", 0))}
                                    for k, v in s.items() if "SQ_INSTS_VALU" in v}
json.dump(t, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(t))
PY
python3 bench.py --steps 200 --warmup 20 > "$OUT/4k_bench.json" 2> "$OUT/4k_bench.err" && echo "bench 4k done"
