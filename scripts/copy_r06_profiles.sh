#!/bin/bash
# After scripts/r06_profiles.sh on the GPU box: copy what is to be judged from gpurun_out/ (scratch, merged back by gpurun)
# into profiles/ (tracked) - INCLUDING the PMC files, whose csrc stamp bench.py checks.
set -e
for t in c2 gram_c2 gram_c5; do cp gpurun_out/r06_pmc_$t.json profiles/r06_pmc_$t.json; done
cp gpurun_out/r06_bench_c2.json profiles/r06_bench_c2.json
cp gpurun_out/kstats_r06_c2.txt profiles/r06_rocprof_kernel_stats_c2.txt
for t in c3 c5; do [ -f gpurun_out/kstats_r06_$t.txt ] && cp gpurun_out/kstats_r06_$t.txt profiles/r06_rocprof_kernel_stats_$t.txt; done
[ -f gpurun_out/kstats_r06_c4.txt ] && cp gpurun_out/kstats_r06_c4.txt profiles/r06_rocprof_kernel_stats_c4_rule_n.txt
for t in c3_through_class c5_through_class rule_n_single_gpu host_budget; do [ -f gpurun_out/r06_$t.json ] && cp gpurun_out/r06_$t.json profiles/r06_$t.json; done
for t in lanes_sweep trd_determinism; do [ -f gpurun_out/r06_$t.txt ] && cp gpurun_out/r06_$t.txt profiles/r06_$t.txt; done
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
for f in ("profiles/r06_pmc_c2.json", "profiles/r06_pmc_gram_c2.json", "profiles/r06_pmc_gram_c5.json"):
    have = json.load(open(f))["csrc_sha16"]
    print(f, "csrc", bench.csrc_hash(), "pmc", have, "OK" if have == bench.csrc_hash() else "STALE: rerun scripts/r06_profiles.sh")
PY
