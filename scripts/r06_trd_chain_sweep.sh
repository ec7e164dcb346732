#!/bin/bash
# Break-point sweep of the chained persistent reduction (round 6): scripts/probes/trd_probe n cplx reps, checksum(d, e) must not
# depend on where the chain is cut.  Output: gpurun_out/r06_trd_chain_sweep.txt
out=gpurun_out/r06_trd_chain_sweep.txt
mkdir -p gpurun_out
: > $out
P=scripts/probes/trd_probe
run() { echo "== $*" >> $out; env "$@" 2>&1 | grep -v "^$" >> $out; }
for spec in "2920 0" "2501 1" "2048 1" "2048 0" "1500 0" "1300 1"; do
  set -- $spec
  run XMCA_TRD_CHAIN=0 $P $1 $2 4
  run XMCA_TRD_CHAIN=1 $P $1 $2 4
done
for b in 256 512 768 1024 1280 1536 "1024,2048" "768,2048" "1024,1792" "1024,2304" "512,1024,2048" "1024,1536,2048" "1024,2048,2560"; do
  run XMCA_TRD_BREAKS=$b $P 2920 0 4
done
for b in 256 512 768 1024 "512,1536" "512,1280" "512,1792" "256,512,1536" "512,1024,1536" "512,1536,2048"; do
  run XMCA_TRD_BREAKS=$b $P 2501 1 4
done
run XMCA_TRD_CHAIN=1 $P 2920 0 3 1
run XMCA_TRD_CHAIN=0 $P 2920 0 3 1
cat $out
