#!/bin/bash
V=real_time_path_tracing_with_spatiotemporal_filtering_amd/variants
O=gpurun_out/r4j; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', d['ms_per_step'], {k:(v['avg_us'], v['launches_per_frame']) for k,v in d.get('kernels',{}).items()})" "$1"; }
RTPT_LIB_PATH=$PWD/$V/librtpt_pool_r8f48.so RTPT_TRACE_POOL=1 RTPT_NO_TRACE_FUSION=1 timeout -k 10 300 python bench.py --workload instanced --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line "pool_r8f48" >> $O/ab.txt
RTPT_LIB_PATH=$PWD/$V/librtpt_ptrows8.so RTPT_NO_TRACE_FUSION=1 timeout -k 10 300 python bench.py --workload instanced --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line "ptrows8 instanced" >> $O/ab.txt
RTPT_LIB_PATH=$PWD/$V/librtpt_ptrows8.so RTPT_NO_TRACE_FUSION=1 timeout -k 10 300 python bench.py --workload 4k --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line "ptrows8 4k" >> $O/ab.txt
RTPT_NO_TRACE_FUSION=1 timeout -k 10 300 python bench.py --workload instanced --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line "default unfused instanced" >> $O/ab.txt
RTPT_LIB_PATH=$PWD/$V/librtpt_ab.so RTPT_TRACE_POOL=1 timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "million_triangle_4k_frame or bvh" > $O/pytest_pool_ab.txt 2>&1; tail -2 $O/pytest_pool_ab.txt
cat $O/ab.txt
