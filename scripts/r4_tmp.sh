V=$PWD/real_time_path_tracing_with_spatiotemporal_filtering_amd/variants
for lib in "" lr1 lr3 lr4 ""; do
  L=""; [ -n "$lib" ] && L=$V/librtpt_$lib.so
  RTPT_LIB_PATH=$L python bench.py --workload instanced --steps 60 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$lib]', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done
