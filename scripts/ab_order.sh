#!/bin/bash
# on-box A/B of compile-time macros of atrous.hip: usage [PMC=1] scripts/ab_order.sh "<EXTRA flags 0>" "<EXTRA flags 1>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PKG=real_time_path_tracing_with_spatiotemporal_filtering_amd
mkdir -p gpurun_out/ab
i=0
for V in "$@"; do
  i=$((i+1))
  touch $PKG/csrc/atrous.hip
  make -s -C $PKG/csrc "EXTRA=$V" > gpurun_out/ab/build_$i.log 2>&1 || { echo "build '$V' failed"; tail -5 gpurun_out/ab/build_$i.log; exit 1; }
  echo "== $V"
  timeout -k 10 120 python3 scratch/atrous_k.py 3840x2160 1920x1080 || exit 1
  if [ -n "$PMC" ]; then
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ab/pmc_$i -- python3 scratch/atrous_k.py 3840x2160 > gpurun_out/ab/pmc_$i.out 2> gpurun_out/ab/pmc_$i.err || { echo "pmc $V failed"; exit 1; }
  python3 - gpurun_out/ab/pmc_$i <<'PY'
import csv,glob,sys,os
from collections import defaultdict
acc=defaultdict(lambda:[0.0,0])
for path in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for row in csv.DictReader(open(path)):
        if 'atrous' not in row['Kernel_Name']: continue
        c=acc[(row['Kernel_Name'].split('(')[0][-34:],row['Counter_Name'])]; c[0]+=float(row['Counter_Value']); c[1]+=1
for k,v in sorted(acc.items()): print(k, 'fetch (KB x2 corrected) %.1f MB'%(v[0]/v[1]*2048/1e6), 'n',v[1])
PY
  fi
done
