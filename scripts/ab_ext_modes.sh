
for f in 0x1F0 0x60 0x100 0x160; do
  python bench.py --workload 4k --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --flags $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('flags $f |', d['ms_per_step'], {k:(v['avg_us'], v['launches_per_frame']) for k,v in d.get('kernels',{}).items()})"
done
