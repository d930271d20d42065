#!/bin/bash
# frame time of emulated strips (one process = one rank's strip, no communication) and whole frames, per library
# usage: scripts/ab_strips.sh default|lib.so ...
for LIB in "$@"; do
  TAG=$(basename "$LIB" .so)
  if [ "$LIB" = default ]; then unset RTPT_LIB_PATH; else export RTPT_LIB_PATH="$PWD/$LIB"; fi
  for ES in 3/8 1/4 0/2; do
    timeout -k 10 200 python bench.py --emulate-strip $ES --steps 400 --warmup 40 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$TAG', d['emulated_strip'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
  done
  for WL in 1080p 4k; do
    timeout -k 10 200 python bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$TAG', '$WL', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
  done
done
