#!/bin/bash
# kernel-trace timing of the five body convs under environment variants given as arguments "NAME=VAL,NAME=VAL ..."
mkdir -p gpurun_out/ab; export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  ( IFS=','; for kv in $v; do export "$kv"; done
    timeout -s KILL 90 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab/v$i -- python3 tools/prof_conv.py 5 > gpurun_out/ab/v$i.log 2>&1 ) || { echo "variant $v failed"; tail -3 gpurun_out/ab/v$i.log; exit 1; }
  python3 - "$v" gpurun_out/ab/v$i <<'PY'
import csv,glob,sys,collections
rows=list(csv.DictReader(open(glob.glob(sys.argv[2]+'/vm/*kernel_trace.csv')[0])))
k=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows if 'conv3x3' in r['Kernel_Name']]
k=k[5:]  # drop first rep
per=[sorted(k[i::5])[len(k[i::5])//2]/1e3 for i in range(5)]
print(sys.argv[1], ' '.join(f'{p:.1f}' for p in per), 'sum %.1f us'%sum(per))
PY
done
