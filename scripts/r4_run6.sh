#!/bin/bash
V=real_time_path_tracing_with_spatiotemporal_filtering_amd/variants
O=gpurun_out/r4e; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', d['ms_per_step'], {k:(v['avg_us'], v['launches_per_frame']) for k,v in d.get('kernels',{}).items()})" "$1"; }
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_scene_ext.py tests/test_fuzz_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "bvh or BVH or million or closest or scene or refit or fuzz or sequence" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
for l in default prerot lr1 lr2 lr3; do
  if [ $l = default ]; then unset RTPT_LIB_PATH; else export RTPT_LIB_PATH=$PWD/$V/librtpt_$l.so; fi
  RTPT_NO_TRACE_FUSION=1 timeout -k 10 300 python bench.py --workload instanced --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line "$l unfused" >> $O/bvh_ab.txt
  timeout -k 10 300 python bench.py --workload instanced --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line "$l" >> $O/bvh_ab.txt
done
for l in count lr2c; do
  RTPT_LIB_PATH=$PWD/$V/librtpt_$l.so timeout -k 10 300 python scripts/bvh_count.py --frames 2 --out $O/bvh_count_$l.json > $O/bvh_count_$l.txt 2>&1
done
cat $O/bvh_ab.txt; grep "K2 total" $O/bvh_count_*.txt
