#!/bin/bash
# on-box A/B of compile-time macros on the reference's default config (1000x800, 32 segments, N = 9)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PKG=real_time_path_tracing_with_spatiotemporal_filtering_amd
mkdir -p gpurun_out/ab
i=0
for V in "$@"; do
  i=$((i+1))
  touch $PKG/csrc/*.hip
  make -s -C $PKG/csrc "EXTRA=$V" > gpurun_out/ab/build_$i.log 2>&1 || { echo "build '$V' failed"; tail -5 gpurun_out/ab/build_$i.log; exit 1; }
  echo "== $V"
  for W in reference; do
  timeout -k 10 200 python3 bench.py --workload $W --no-cpu-baseline --no-secondary | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})" || exit 1
  done
done
