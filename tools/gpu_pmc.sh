#!/bin/bash
# PMC passes over the body convs (one rocprofv3 run per counter set).  Usage: tools/gpu_pmc.sh TAG RING
TAG=${1:-pmc}; export SR355_RING=${2:-1}; export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
rocprofv3 -L > gpurun_out/$TAG/counters.txt 2>&1
run() { name=$1; shift; timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$TAG/$name -- python3 tools/prof_conv.py 2 > gpurun_out/$TAG/$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/$TAG/$name.log; exit 1; }; }
timeout -s KILL 90 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/trace -- python3 tools/prof_conv.py 2 > gpurun_out/$TAG/trace.log 2>&1 || exit 1
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
echo passes done
