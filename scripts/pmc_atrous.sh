#!/bin/bash
TAG=${1:-pmc_at}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
i=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES" ; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/pass$i" -- python3 scratch/atrous_k.py 3840x2160 > "$OUT/pass$i.out" 2> "$OUT/pass$i.err" || echo "pass $i failed"
  echo "pass $i done: $G"
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,os
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(lambda:[0.0,0]))
for path in glob.glob(os.path.join(sys.argv[1],"pass*","**","*counter_collection.csv"),recursive=True):
    for row in csv.DictReader(open(path)):
        if 'atrous' not in row['Kernel_Name']: continue
        key=(row['Kernel_Name'].split('(')[0][-40:], row.get('LDS_Block_Size',''))
        c=acc[key][row['Counter_Name']]; c[0]+=float(row['Counter_Value']); c[1]+=1
for key in sorted(acc):
    print(key, {n: float('%.4g'%(v[0]/v[1])) for n,v in sorted(acc[key].items())})
PY
