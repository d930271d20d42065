#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', d['ms_per_step'], d.get('rays_per_frame'), {k:(v['avg_us'], v['launches_per_frame']) for k,v in d.get('kernels',{}).items()})" "$1"; }
export RTPT_TRACE_POOL=1
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py tests/test_cpp_host.py tests/test_scene_ext.py tests/test_parity_gpu.py -x -q -m gpu -k "million or lattice or refit or bvh or materials" > $O/pytest_pool.txt 2>&1; echo "pytest rc $?" >> $O/pytest_pool.txt; tail -5 $O/pytest_pool.txt
for e in "RTPT_TRACE_POOL=1" "RTPT_TRACE_POOL=0" "RTPT_TRACE_POOL=1 RTPT_NO_TRACE_FUSION=1" "RTPT_TRACE_POOL=0 RTPT_NO_TRACE_FUSION=1"; do
  env $e timeout -k 10 300 python bench.py --workload instanced --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line "$e" >> $O/pool_ab.txt
done
env RTPT_TRACE_POOL=1 timeout -k 10 300 python bench.py --workload instanced --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --emulate-strip 3/8 2>/dev/null | line "pool strip 3/8" >> $O/pool_ab.txt
cat $O/pool_ab.txt
