#!/bin/bash
# Same-box A/B of an environment switch on the training iterations: tools/gpu_ab_env_train.sh TAG VAR A B [C ...]
TAG=$1; VAR=$2; shift 2; O=gpurun_out/$TAG; mkdir -p $O
for i in 1 2 3; do
  for v in "$@"; do
    echo -n "$VAR=$v " >> $O/time.log
    env $VAR=$v timeout -k 10 200 python3 tools/time_pixel.py 2>&1 | grep -v amdgpu | tr -s " " >> $O/time.log || { echo "timing failed"; exit 1; }
  done
done
cat $O/time.log
