#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PKG=real_time_path_tracing_with_spatiotemporal_filtering_amd
mkdir -p gpurun_out/ab
i=0
for V in "$@"; do
  i=$((i+1))
  touch $PKG/csrc/*.hip
  make -s -C $PKG/csrc "EXTRA=$V" > gpurun_out/ab/build_$i.log 2>&1 || { echo "build '$V' failed"; tail -5 gpurun_out/ab/build_$i.log; exit 1; }
  echo "== $V"
  for S in 3/8 1/4 0/2; do
  timeout -k 10 200 python3 bench.py --emulate-strip $S --steps 400 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('strip $S', d['ms_per_step'], d['kernels']['k_pathtrace']['avg_us'])" || exit 1
  done
done
