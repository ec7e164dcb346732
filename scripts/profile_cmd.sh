#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace summary of an arbitrary command.
#   scripts/profile_cmd.sh <tag> <command...>   ->  gpurun_out/kstats_<tag>.txt
set -u
tag=$1; shift
REPO=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
( cd $REPO && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- "$@" > $REPO/gpurun_out/prof_$tag.out 2> $REPO/gpurun_out/prof_$tag.err )
cd $REPO
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python scripts/kstats.py "$db" 30 > gpurun_out/kstats_$tag.txt 2>&1
