#!/bin/bash
# After scripts/r05_profiles.sh on the GPU box: copy what is to be judged from gpurun_out/ (scratch, merged back by gpurun)
# into profiles/ (tracked) - INCLUDING the PMC files, whose csrc stamp bench.py checks.
set -e
for t in c2 gram_c2 gram_c5; do cp gpurun_out/r05_pmc_$t.json profiles/r05_pmc_$t.json; done
cp gpurun_out/r05_bench_c2.json profiles/r05_bench_c2.json
cp gpurun_out/kstats_r05_c2.txt profiles/r05_rocprof_kernel_stats_c2.txt
for t in c3 c5; do [ -f gpurun_out/kstats_r05_$t.txt ] && cp gpurun_out/kstats_r05_$t.txt profiles/r05_rocprof_kernel_stats_$t.txt; done
[ -f gpurun_out/kstats_r05_c4.txt ] && cp gpurun_out/kstats_r05_c4.txt profiles/r05_rocprof_kernel_stats_c4_rule_n.txt
for t in c3_through_class c5_through_class rule_n_single_gpu; do [ -f gpurun_out/r05_$t.json ] && cp gpurun_out/r05_$t.json profiles/r05_$t.json; done
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
for f in ("profiles/r05_pmc_c2.json", "profiles/r05_pmc_gram_c2.json", "profiles/r05_pmc_gram_c5.json"):
    have = json.load(open(f))["csrc_sha16"]
    print(f, "csrc", bench.csrc_hash(), "pmc", have, "OK" if have == bench.csrc_hash() else "STALE: rerun scripts/r05_profiles.sh")
PY
