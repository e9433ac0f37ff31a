#!/bin/bash
# Same-box A/B of two builds of the library: tools/gpu_ab_lib.sh TAG OTHER.so
# (pixel and adversarial training iterations, alternating the in-tree library and OTHER three times)
TAG=$1; OTHER=$2; O=gpurun_out/$TAG; mkdir -p $O
for i in 1 2 3; do
  for lib in "" "$OTHER"; do
    echo "lib=${lib:-in-tree}" >> $O/time.log
    SR355_LIB=$lib timeout -k 10 200 python3 tools/time_pixel.py >> $O/time.log 2>&1 || { echo "timing failed"; tail -3 $O/time.log; exit 1; }
    SR355_LIB=$lib timeout -k 10 200 python3 tools/time_adv.py 20 pair >> $O/time.log 2>&1 || { echo "timing failed"; tail -3 $O/time.log; exit 1; }
  done
done
grep -v amdgpu.ids $O/time.log
