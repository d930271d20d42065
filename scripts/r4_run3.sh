#!/bin/bash
# round-4 measurement batch (run on the GPU box): timelines, chain policy on a strip, leaf-ratio variants of the traversal
V=real_time_path_tracing_with_spatiotemporal_filtering_amd/variants
O=gpurun_out/r4b; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', d['ms_per_step'], {k:(v['avg_us'], v['launches_per_frame']) for k,v in d.get('kernels',{}).items()})" "$1"; }
RTPT_LIB_PATH=$PWD/$V/librtpt_timeline.so timeout -k 10 200 python scripts/tile_timeline.py --strip 3/8 --out $O/tl_strip.json > $O/tl_strip.txt 2>&1
RTPT_LIB_PATH=$PWD/$V/librtpt_timeline.so timeout -k 10 200 python scripts/tile_timeline.py --out $O/tl_full.json > $O/tl_full.txt 2>&1
for e in "X=0" "RTPT_CHAIN_WG_PER_CU=1" "RTPT_CHAIN_MAX=3" "RTPT_CHAIN_MAX=3 RTPT_CHAIN_FINAL=1" "RTPT_CHAIN_FINAL=1" "RTPT_CHAIN_G1=2" "RTPT_CHAIN_MIN_PIXELS=100000000"; do
  env $e timeout -k 10 200 python bench.py --workload 4k --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --emulate-strip 3/8 2>/dev/null | line "$e" >> $O/chain_strip_ab.txt
done
for l in lr2 lr4; do
  RTPT_LIB_PATH=$PWD/$V/librtpt_$l.so timeout -k 10 300 python bench.py --workload instanced --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line $l >> $O/lr.txt
  RTPT_LIB_PATH=$PWD/$V/librtpt_${l}c.so timeout -k 10 300 python scripts/bvh_count.py --frames 2 --out $O/bvh_count_$l.json > $O/bvh_count_$l.txt 2>&1
done
timeout -k 10 300 python bench.py --workload instanced --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line default >> $O/lr.txt
cat $O/*.txt
