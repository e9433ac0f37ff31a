#!/bin/bash
# One GPU-box session for the committed profile summaries.  Usage: tools/gpu_profiles.sh TAG
# (the programs stand directly behind `--`: the profiler's preloaded library initialises the GPU before they start)
TAG=${1:-r02}; export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
# per-kernel summary of the headline bench: kernels of one stream
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --no-cpu-baseline --no-train --no-extras --steps 20 --warmup 5 > $O/bench.log 2>&1 || { echo "bench profile failed"; tail -3 $O/bench.log; exit 1; }
# the same with one launch per conv (what the chain kernel replaces), kernels serialised
SR355_CHAIN=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_nochain -- python3 bench.py --streams 1 --no-cpu-baseline --no-train --no-extras --steps 20 --warmup 5 > $O/bench_nochain.log 2>&1 || { echo "no-chain profile failed"; exit 1; }
# pixel and adversarial training iterations (weight gradients on the main stream: a kernel's duration is its own)
SR355_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pixel -- python3 tools/prof_train.py 10 > $O/pixel.log 2>&1 || { echo "pixel profile failed"; exit 1; }
SR355_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/adv -- python3 tools/prof_adv.py 5 > $O/adv.log 2>&1 || { echo "adversarial profile failed"; exit 1; }
# HBM-side traffic: one PMC pass per counter
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 170 rocprofv3 --pmc $c --output-format csv -d $O/pmc/$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-extras --no-roofline > $O/pmc_$c.log 2>&1 || { echo "pass $c failed"; tail -3 $O/pmc_$c.log; exit 1; }
done
python3 tools/pmc_summary.py $O/pmc > $O/pmc_traffic.json && head -c 1500 $O/pmc_traffic.json
find $O -name "*kernel_stats.csv"
