#!/bin/bash
# on-box A/B of compile-time macros on the 1.15M-triangle workload: usage scripts/ab_inst.sh "<EXTRA 0>" "<EXTRA 1>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PKG=real_time_path_tracing_with_spatiotemporal_filtering_amd
mkdir -p gpurun_out/ab
i=0
for V in "$@"; do
  i=$((i+1))
  touch $PKG/csrc/*.hip $PKG/csrc/*.cpp
  make -s -C $PKG/csrc "EXTRA=$V" > gpurun_out/ab/build_$i.log 2>&1 || { echo "build '$V' failed"; tail -5 gpurun_out/ab/build_$i.log; exit 1; }
  echo "== $V"
  timeout -k 10 200 python3 bench.py --workload instanced --steps 30 --warmup 3 --no-cpu-baseline --no-secondary | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})" || exit 1
done
