#!/bin/bash
# One GPU-box session for the committed profile summaries of a tree.  Usage: tools/gpu_profiles_r03.sh TAG
# (the programs stand directly behind `--`: the profiler's preloaded library initialises the GPU before they start;
# counters are collected in runs of their own, one pass each, without any trace option)
TAG=${1:-r03}; export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
step() { echo "== $1" >> $O/session.log; }
# per-kernel summary of the headline bench: kernels of one stream
step bench; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --no-cpu-baseline --no-train --no-extras --steps 20 --warmup 5 > $O/bench.log 2>&1 || { echo "bench profile failed"; tail -3 $O/bench.log; exit 1; }
# the same with one launch per conv (what the chain kernel replaces), kernels serialised
step nochain; SR355_CHAIN=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_nochain -- python3 bench.py --streams 1 --no-cpu-baseline --no-train --no-extras --steps 20 --warmup 5 > $O/bench_nochain.log 2>&1 || { echo "no-chain profile failed"; exit 1; }
# pixel and adversarial training iterations (weight gradients on the main stream: a kernel's duration is its own)
step pixel; SR355_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pixel -- python3 tools/prof_train.py 10 > $O/pixel.log 2>&1 || { echo "pixel profile failed"; exit 1; }
step pixel64; SR355_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pixel64 -- python3 tools/prof_train.py 6 64 > $O/pixel64.log 2>&1 || { echo "pixel batch-64 profile failed"; exit 1; }
step adv; SR355_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/adv -- python3 tools/prof_adv.py 5 > $O/adv.log 2>&1 || { echo "adversarial profile failed"; exit 1; }
# fabric traffic: one PMC pass per counter, over the headline forward and over the pixel training iteration
for c in FETCH_SIZE WRITE_SIZE; do
  step "pmc $c bench"; timeout -s KILL 170 rocprofv3 --pmc $c --output-format csv -d $O/pmc/$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-extras --no-roofline > $O/pmc_$c.log 2>&1 || { echo "pass $c failed"; tail -3 $O/pmc_$c.log; exit 1; }
  step "pmc $c train"; SR355_WGRAD_STREAM=0 timeout -s KILL 170 rocprofv3 --pmc $c --output-format csv -d $O/pmc_train/$c -- python3 tools/prof_train.py 2 > $O/pmc_train_$c.log 2>&1 || { echo "train pass $c failed"; tail -3 $O/pmc_train_$c.log; exit 1; }
done
python3 tools/pmc_summary.py $O/pmc > $O/pmc_traffic.json || exit 1
python3 tools/pmc_summary.py $O/pmc_train > $O/pmc_traffic_train.json || exit 1
# matrix pipe and wave states: one pass of 8 SQ counters + GRBM_GUI_ACTIVE each
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
step "sq bench"; timeout -s KILL 170 rocprofv3 --pmc $SQ --output-format csv -d $O/sq/sq -- python3 bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-extras --no-roofline > $O/sq.log 2>&1 || { echo "sq pass failed"; tail -5 $O/sq.log; exit 1; }
step "sq train"; SR355_WGRAD_STREAM=0 timeout -s KILL 170 rocprofv3 --pmc $SQ --output-format csv -d $O/sq_train/sq -- python3 tools/prof_train.py 2 > $O/sq_train.log 2>&1 || { echo "sq train pass failed"; tail -5 $O/sq_train.log; exit 1; }
python3 tools/pmc_sq_summary.py $O/sq > $O/pmc_sq.json || exit 1
python3 tools/pmc_sq_summary.py $O/sq_train > $O/pmc_sq_train.json || exit 1
find $O -name "*kernel_stats.csv"
python3 - <<PY
import json
for f in ("pmc_traffic", "pmc_traffic_train", "pmc_sq", "pmc_sq_train"):
    d = json.load(open("$O/%s.json" % f))["kernels"]
    for k, v in d.items():
        print(f, k[:70], {a: b for a, b in v.items() if a != "per_launch"})
PY
