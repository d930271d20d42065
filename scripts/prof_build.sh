#!/bin/bash
# rebuild with EXTRA flags on the box and print rocprofv3 kernel stats of scratch/quickbench.py (4K)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PKG=real_time_path_tracing_with_spatiotemporal_filtering_amd
touch $PKG/csrc/*.hip
make -s -C $PKG/csrc "EXTRA=$1" > gpurun_out/build_prof.log 2>&1 || { tail -5 gpurun_out/build_prof.log; exit 1; }
rm -rf gpurun_out/prof_build; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_build -- python3 scratch/quickbench.py 3840x2160 > gpurun_out/prof_build.out 2>&1 || tail -5 gpurun_out/prof_build.out
cut -c1-150 $GRAFT_REPO_ROOT/gpurun_out/prof_build/*/*kernel_stats.csv | head -9
