#!/bin/bash
V=real_time_path_tracing_with_spatiotemporal_filtering_amd/variants
O=gpurun_out/r4c; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', d['ms_per_step'], {k:(v['avg_us'], v['launches_per_frame']) for k,v in d.get('kernels',{}).items()})" "$1"; }
timeout -k 10 200 scripts/micro/node_step > $O/node_step.txt 2>&1
RTPT_LIB_PATH=$PWD/$V/librtpt_timeline.so timeout -k 10 300 python scripts/tile_timeline.py --strip 3/8 --lpt > $O/tl_strip_lpt.txt 2>&1
RTPT_LIB_PATH=$PWD/$V/librtpt_timeline.so timeout -k 10 300 python scripts/tile_timeline.py --workload 1080p --lpt > $O/tl_1080p_lpt.txt 2>&1
RTPT_LIB_PATH=$PWD/$V/librtpt_timeline.so timeout -k 10 300 python scripts/tile_timeline.py --lpt > $O/tl_4k_lpt.txt 2>&1
for l in chg4 chg2; do for e in "X=0" "RTPT_CHAIN_WG_PER_CU=1"; do
  RTPT_LIB_PATH=$PWD/$V/librtpt_$l.so env $e timeout -k 10 200 python bench.py --workload 4k --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --emulate-strip 3/8 2>/dev/null | line "$l $e" >> $O/chain_g_strip.txt
done; done
for l in chg4; do for e in "X=0"; do
  RTPT_LIB_PATH=$PWD/$V/librtpt_$l.so env $e timeout -k 10 200 python bench.py --workload 1080p --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | line "1080p $l $e" >> $O/chain_g_strip.txt
done; done
cat $O/*.txt
