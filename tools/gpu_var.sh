#!/bin/bash
# Parity tests with the ring path forced on under one kernel variant, timelines, then kernel-trace timing of variants.
# Usage: tools/gpu_var.sh TESTVARIANT "ENV=..,ENV=.." ...
V=$1; shift
mkdir -p gpurun_out
SR355_RING=1 SR355_VARIANT=$V timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/tests_var$V.log 2>&1
rc=$?; echo "tests_var${V}_exit=$rc"; tail -n 4 gpurun_out/tests_var$V.log
if [ $rc -ne 0 ]; then exit $rc; fi
for c in "8 32 0" "20 32 0" "24 64 1" "24 64 2"; do SR355_VARIANT=$V timeout -k 5 60 super_resolution_amd/csrc/tools/trace_conv $c || exit 1; done
./tools/gpu_ab2.sh "$@"
