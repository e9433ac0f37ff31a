#!/bin/bash
# Parity tests then the bench line (no CPU baseline), under the environment given as "NAME=VAL ..." arguments.
mkdir -p gpurun_out
for kv in "$@"; do export "$kv"; done
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/tests_quick.log 2>&1
rc=$?; echo "tests_exit=$rc"; tail -n 3 gpurun_out/tests_quick.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-train > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err || { tail -5 gpurun_out/bench_quick.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_quick.json'))
print('value',d['value'],'ms',d['ms_per_step'],'roof',d['roofline']['frac'],[(k['shape'],k['us']) for k in d['kernel_shapes']])
PY
