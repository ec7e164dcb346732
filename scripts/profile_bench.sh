#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel-trace summary of the same command.
# Outputs under gpurun_out/; copy the ones to keep into profiles/.
set -u
mkdir -p gpurun_out
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -1 gpurun_out/bench_full.json | cut -c1-400
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rule-n > $REPO/gpurun_out/bench_prof.json 2> $REPO/gpurun_out/bench_prof.err
cd $REPO
db=$(find /tmp/prof_bench -name "*.db" | head -1)
python scripts/kstats.py "$db" 25 > gpurun_out/rocprof_kernel_stats.txt 2>&1
head -30 gpurun_out/rocprof_kernel_stats.txt
