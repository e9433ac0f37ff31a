#!/bin/bash
# Matrix-pipe utilisation and wave states of the bench's kernels: one rocprofv3 PMC pass (8 SQ + 1 GRBM counter)
# over a short bench run with the kernels serialised.  Usage: tools/gpu_pmc_sq.sh TAG
TAG=${1:-pmcsq}; export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
timeout -s KILL 170 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$TAG/sq -- python3 bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-extras --no-roofline > gpurun_out/$TAG/sq.log 2>&1 || { echo "pass failed"; tail -5 gpurun_out/$TAG/sq.log; exit 1; }
python3 tools/pmc_sq_summary.py gpurun_out/$TAG > gpurun_out/$TAG/sq.json && head -c 3000 gpurun_out/$TAG/sq.json
