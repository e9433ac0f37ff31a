#!/bin/bash
# A/B on the GPU box: parity tests with the ring kernel forced on, then bench with ring off / on.
mkdir -p gpurun_out
SR355_RING=1 timeout -k 10 600 python -u -m pytest tests -m gpu -q -s --timeout 300 --timeout-method thread > gpurun_out/tests_ring.log 2>&1
rc=$?; echo "tests_ring_exit=$rc"; tail -n 12 gpurun_out/tests_ring.log
if [ $rc -gt 1 ]; then exit $rc; fi
SR355_RING=0 timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/bench_ring0.json 2> gpurun_out/bench_ring0.err || exit 1
python -c "import json;d=json.load(open('gpurun_out/bench_ring0.json'));print('ring0',d['value'],d['ms_per_step'],[(k['shape'],k['us']) for k in d['kernel_shapes']])"
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/bench_ring1.json 2> gpurun_out/bench_ring1.err || exit 1
python -c "import json;d=json.load(open('gpurun_out/bench_ring1.json'));print('ring1',d['value'],d['ms_per_step'],[(k['shape'],k['us']) for k in d['kernel_shapes']])"
