# in-frame A/B on one rank's strip of an 8-rank job and on the full frame: scripts/strip_ab.sh "ENV=.." "ENV=.."
for e in "$@"; do
  for s in ${STRIP_AB_RUNS:-"--emulate-strip 3/8" ""}; do
    env $e python bench.py --workload 4k --steps 200 --warmup 20 --no-cpu-baseline --no-secondary $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$e | $s |', d['ms_per_step'], {k:(v['avg_us'], v['launches_per_frame']) for k,v in d.get('kernels',{}).items()})"
  done
done
