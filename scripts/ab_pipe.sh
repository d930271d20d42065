#!/bin/bash
# on-box A/B of compile-time macros, serial vs two frames in flight at 4K: usage scripts/ab_pipe.sh "<EXTRA 0>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PKG=real_time_path_tracing_with_spatiotemporal_filtering_amd
mkdir -p gpurun_out/ab
i=0
for V in "$@"; do
  i=$((i+1))
  touch $PKG/csrc/*.hip
  make -s -C $PKG/csrc "EXTRA=$V" > gpurun_out/ab/build_$i.log 2>&1 || { echo "build '$V' failed"; tail -5 gpurun_out/ab/build_$i.log; exit 1; }
  echo "== $V"
  timeout -k 10 200 python3 scratch/overlap3.py || exit 1
done
