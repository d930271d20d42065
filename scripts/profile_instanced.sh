cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_b2; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_instanced" -- python3 bench.py --workload instanced --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > "$OUT/instanced_bench_under_rocprof.json" 2> "$OUT/stats_instanced.err"
find "$OUT/stats_instanced" -name "*kernel_stats.csv" -exec cp {} "$OUT/instanced_kernel_stats.csv" \;
rm -rf "$OUT/stats_instanced"
python - <<'PY'
import json
b=json.load(open("gpurun_out/r03_b2/instanced_bench_under_rocprof.json"))
print(b['ms_per_step'], b['value'], {k:v['avg_us'] for k,v in b['kernels'].items()})
PY
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "million or blit or present or instanced" 2>&1 | tail -2
