#!/bin/bash
V=real_time_path_tracing_with_spatiotemporal_filtering_amd/variants
O=gpurun_out/r4g; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -5 $O/pytest.txt
RTPT_LIB_PATH=$PWD/$V/librtpt_count.so timeout -k 10 300 python scripts/bvh_count.py --frames 2 --out $O/bvh_count.json > $O/bvh_count.txt 2>&1
cat $O/bvh_count.txt
