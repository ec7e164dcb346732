#!/bin/bash
# GPU box: rocprofv3 kernel-trace summary of the C2 step (bench.py without the rule_n / C5 / CPU legs) -> gpurun_out/
set -u
mkdir -p gpurun_out
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c2
rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o b -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-rule-n --no-c5 --no-e2e > $REPO/gpurun_out/prof_r04_c2.out 2> $REPO/gpurun_out/prof_r04_c2.err
cd $REPO
db=$(find /tmp/prof_c2 -name "*.db" | head -1)
python scripts/kstats.py "$db" 40 > gpurun_out/kstats_r04_c2.txt 2>&1
cat gpurun_out/kstats_r04_c2.txt
