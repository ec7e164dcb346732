#!/bin/bash
# Round 6: timing of the chained reduction + determinism soak of rule_n inside surrogate lanes (bit-equality of repeated calls; the
# reduction's input/output checksums are traced with XMCA_TRACE=trdsum, which also makes a hazard MORE likely to show: it adds host
# syncs between the lanes' kernels).  Output: gpurun_out/r06_trd_determinism.txt
out=gpurun_out/r06_trd_determinism.txt; : > $out
run() { echo "== $*" >> $out; env "$@" 2>&1 | grep "reduction\|checksum\|error" | sed 's/trace.*lam max.*checksum/checksum/' >> $out; }
for spec in "2920 0" "2501 1" "2048 1" "2048 0" "1500 0" "1300 1" "1000 0" "1000 1" "451 1" "500 0"; do
  set -- $spec
  run XMCA_TRD_CHAIN=0 scripts/probes/trd_probe $1 $2 4
  run XMCA_TRD_CHAIN=1 scripts/probes/trd_probe $1 $2 4
done
soak() { echo "== soak: $*" >> $out; env "$@" 2>/dev/null | tail -3 >> $out; rm -f gpurun_out/de_*.bin; }
soak XMCA_RULE_N_LANES=4 XMCA_TRACE=trdsum python scripts/det_probe.py 2000 5000 4000 1 400
soak XMCA_RULE_N_LANES=4 python scripts/det_probe.py 2000 5000 4000 1 300
soak XMCA_RULE_N_LANES=2 XMCA_TRACE=trdsum python scripts/det_probe.py 2600 6000 5000 1 100
soak XMCA_RULE_N_LANES=4 XMCA_TRACE=trdsum python scripts/det_probe.py 1500 4000 3000 0 150
soak XMCA_RULE_N_LANES=3 XMCA_TRACE=trdsum python scripts/det_probe.py 900 2200 1700 1 300
cat $out
