for e in "A=1" "RTPT_CHAIN_WG_PER_CU=1" "RTPT_CHAIN_MIN_PIXELS=100000000"; do
  env $e python bench.py --workload 1080p --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$e |', d['ms_per_step'], {k:(v['avg_us'], v['launches_per_frame']) for k,v in d.get('kernels',{}).items()})"
done
