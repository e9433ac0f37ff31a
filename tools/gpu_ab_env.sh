#!/bin/bash
# Same-box A/B of an environment switch of the library: tools/gpu_ab_env.sh TAG VAR A B [pmc]
# (chain launch time alternating A / B three times; with `pmc` also the fabric traffic of each setting)
TAG=$1; VAR=$2; A=$3; B=$4; export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
for i in 1 2 3; do
  for v in $A $B; do
    echo -n "$VAR=$v " >> $O/time.log
    env $VAR=$v timeout -k 10 120 python3 tools/time_chain.py 16 128 20 >> $O/time.log 2>&1 || { echo "timing failed"; tail -3 $O/time.log; exit 1; }
  done
done
cat $O/time.log | grep -v amdgpu.ids
if [ "$5" = "pmc" ]; then
  for v in $A $B; do
    export $VAR=$v
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$v/$c -- python3 tools/time_chain.py 16 128 2 > $O/pmc_${v}_$c.log 2>&1 || { echo "pass $c failed"; tail -3 $O/pmc_${v}_$c.log; exit 1; }
    done
    python3 tools/pmc_summary.py $O/pmc_$v > $O/traffic_$v.json && python3 -c "
import json,sys; d=json.load(open('$O/traffic_$v.json'))['kernels']
for k,v in d.items():
    if 'chain' in k: print('$VAR=$v chain traffic GB: read %.2f write %.2f total %.2f' % (v['read_bytes_per_launch']/1e9, v['write_bytes_per_launch']/1e9, v['traffic_bytes_per_launch']/1e9))"
  done
fi
