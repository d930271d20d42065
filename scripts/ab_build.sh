#!/bin/bash
# on-box A/B of compile-time macros (whole library): usage scripts/ab_build.sh "<EXTRA flags 0>" "<EXTRA flags 1>" ...
# each variant is rebuilt on the box and timed with scratch/quickbench.py ($QB_ARGS, default 4K)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PKG=real_time_path_tracing_with_spatiotemporal_filtering_amd
mkdir -p gpurun_out/ab
i=0
for V in "$@"; do
  i=$((i+1))
  touch $PKG/csrc/*.hip
  make -s -C $PKG/csrc "EXTRA=$V" > gpurun_out/ab/build_$i.log 2>&1 || { echo "build '$V' failed"; tail -5 gpurun_out/ab/build_$i.log; exit 1; }
  echo "== $V"
  timeout -k 10 120 python3 scratch/quickbench.py ${QB_ARGS:-3840x2160} || exit 1
  timeout -k 10 120 python3 scratch/quickbench.py ${QB_ARGS:-3840x2160} || exit 1
done
