#!/bin/bash
V=real_time_path_tracing_with_spatiotemporal_filtering_amd/variants
O=gpurun_out/r4f; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', d['ms_per_step'], {k:(v['avg_us'], v['launches_per_frame']) for k,v in d.get('kernels',{}).items()})" "$1"; }
timeout -k 10 1700 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -5 $O/pytest.txt
RTPT_LIB_PATH=$PWD/$V/librtpt_ab.so timeout -k 10 600 python -m pytest tests/test_chain_gpu.py -x -q -m gpu -k sliding > $O/pytest_ab.txt 2>&1; tail -2 $O/pytest_ab.txt
for w in "--workload 4k --emulate-strip 3/8" "--workload 4k --emulate-strip 0/8" "--workload 4k" "--workload 1080p" "--workload instanced --steps 60" "--workload instanced --steps 100 --emulate-strip 3/8" "--workload reference"; do
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary $w 2>/dev/null | line "$w" >> $O/bench.txt
done
for w in "--workload 4k --emulate-strip 3/8" "--workload 1080p" "--workload 4k"; do
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --frames-in-flight 2 $w 2>/dev/null | line "2 in flight $w" >> $O/bench.txt
done
cat $O/bench.txt
