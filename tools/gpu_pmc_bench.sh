#!/bin/bash
# HBM-side traffic of the bench's kernels: one rocprofv3 PMC pass per counter over a short bench run
# (kernels serialised), then per-kernel averages.  Usage: tools/gpu_pmc_bench.sh TAG
TAG=${1:-pmcb}; export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 170 rocprofv3 --pmc $c --output-format csv -d gpurun_out/$TAG/$c -- python3 bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-train > gpurun_out/$TAG/$c.log 2>&1 || { echo "pass $c failed"; tail -3 gpurun_out/$TAG/$c.log; exit 1; }
done
python3 tools/pmc_summary.py gpurun_out/$TAG > gpurun_out/$TAG/traffic.json && cat gpurun_out/$TAG/traffic.json
