#!/bin/bash
# PMC of the BVH traversal kernel (k_pathtrace<BVH> + its queue windows) and the BVH G-buffer on BASELINE configs[4]
# (1,152,000 triangles, 3840x2160, 8 segments): L2 hit rate, resident waves per SIMD, VALU busy, lane utilisation —
# the numbers north_star asks for on the traversal kernel.  usage: COMMIT=<hash> scripts/pmc_traversal.sh <tag>
TAG=${1:-pmc_trav}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for G in \
  "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" ; do
  i=$((i+1))
  # RTPT_NO_TRACE_FUSION=1: K0 + K1 and K2 as launches of their own, so that the counters are the traversal's alone
  RTPT_NO_TRACE_FUSION=1 timeout -k 10 280 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/p$i -- python3 bench.py --workload instanced --steps 6 --warmup 2 --prewarm-seconds 0 --no-cpu-baseline --no-secondary > $OUT/p$i.json 2> $OUT/p$i.err || { echo pass $i failed; tail -3 $OUT/p$i.err; }
  echo "pass $i done"
done
python3 - $OUT "${COMMIT:-unknown}" <<'PY'
import csv, glob, sys, collections, json
out, commit = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
def short(n):
    if "k_gbuffer_pathtrace" in n:
        return "k_gbuffer_pathtrace"
    for k in ("k_pathtrace_queue", "k_pathtrace", "k_gbuffer", "k_atrous"):
        if k in n:
            return k
    return None
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        c = acc[k][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"]); c[1] += 1
for f in glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; dur[k][1] += 1
res = {"_source": commit, "_note": "rocprofv3 --pmc, separate passes, bench.py --workload instanced (1,152,000 triangles, 3840x2160, 8 segments); "
       "per launch averages.  waves_per_simd = SQ_WAVE_CYCLES (quad-cycles) * 4 / (GRBM_GUI_ACTIVE / 8 XCDs) / 1024 SIMDs; "
       "lane_utilisation = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64); valu_busy = SQ_ACTIVE_INST_VALU * 4 / (GRBM_GUI_ACTIVE / 8 * 1024)"}
for k, cs in acc.items():
    d = {n: v[0] / v[1] for n, v in cs.items()}
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    if "GRBM_GUI_ACTIVE" in d and "SQ_WAVE_CYCLES" in d:
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0
        d["waves_per_simd"] = d["SQ_WAVE_CYCLES"] * 4.0 / cyc / 1024.0
        d["valu_busy"] = d.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / cyc / 1024.0
    if "SQ_THREAD_CYCLES_VALU" in d and d.get("SQ_ACTIVE_INST_VALU"):
        d["lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / (d["SQ_ACTIVE_INST_VALU"] * 64.0)
    if dur[k][1]:
        d["avg_us_under_pmc"] = dur[k][0] / dur[k][1]
    res[k] = d
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
for k in res:
    if k[0] != "_":
        print(k, {n: round(v, 3) for n, v in res[k].items() if n in ("l2_hit_rate", "waves_per_simd", "valu_busy", "lane_utilisation", "avg_us_under_pmc")})
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
