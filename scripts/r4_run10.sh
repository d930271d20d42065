#!/bin/bash
V=real_time_path_tracing_with_spatiotemporal_filtering_amd/variants
O=gpurun_out/r4i; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', d['ms_per_step'], {k:(v['avg_us'], v['launches_per_frame']) for k,v in d.get('kernels',{}).items()})" "$1"; }
for l in pool_r8 pool_r32 pool_f8 pool_f32 pool_f48; do
  RTPT_LIB_PATH=$PWD/$V/librtpt_$l.so RTPT_TRACE_POOL=1 RTPT_NO_TRACE_FUSION=1 timeout -k 10 300 python bench.py --workload instanced --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line "$l" >> $O/pool_ab.txt
done
for w in "--workload 4k --emulate-strip 3/8" "--workload 4k --emulate-strip 0/8" "--workload 4k" "--workload 1080p" "--workload instanced --steps 60" "--workload instanced --steps 100 --emulate-strip 3/8"; do
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary $w 2>/dev/null | line "balanced comb $w" >> $O/pool_ab.txt
done
cat $O/pool_ab.txt
