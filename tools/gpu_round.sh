#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel summary.  Usage: tools/gpu_round.sh TAG
TAG=${1:-r}
mkdir -p gpurun_out
timeout -k 10 700 python -u -m pytest tests -m gpu -q -s --timeout 300 --timeout-method thread > gpurun_out/tests_$TAG.log 2>&1
rc=$?; echo "tests_exit=$rc"; tail -n 15 gpurun_out/tests_$TAG.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 400 python -u bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rc=$?; echo "bench_exit=$rc"; cat gpurun_out/bench_$TAG.json; tail -n 5 gpurun_out/bench_$TAG.err
if [ $rc -ne 0 ]; then exit $rc; fi
export TMPDIR=/tmp
# per-kernel summary: kernels serialised (one stream), so a kernel's average duration is its own
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --streams 1 --no-cpu-baseline --no-train > gpurun_out/prof_$TAG.log 2>&1
rc=$?; echo "prof_exit=$rc"; tail -n 2 gpurun_out/prof_$TAG.log | cut -c1-1500
find gpurun_out/prof_$TAG -name "*stats*" | head
