#!/bin/bash
# After scripts/r03_profiles.sh (+ the final bench / phase-stamp commands) on the GPU box: copy what is to be judged from
# gpurun_out/ (scratch, merged back by gpurun) into profiles/ (tracked) - INCLUDING the PMC file, whose csrc stamp bench.py checks.
set -e
cp gpurun_out/r03_pmc_c2.json profiles/r03_pmc_c2.json
[ -f gpurun_out/r03_bench_c2_final.json ] && cp gpurun_out/r03_bench_c2_final.json profiles/r03_bench_c2.json || cp gpurun_out/r03_bench_c2.json profiles/r03_bench_c2.json
cp gpurun_out/kstats_r03_c2.txt profiles/r03_rocprof_kernel_stats_c2.txt
cp gpurun_out/kstats_r03_c3.txt profiles/r03_rocprof_kernel_stats_c3.txt
cp gpurun_out/kstats_r03_c4.txt profiles/r03_rocprof_kernel_stats_c4_rule_n.txt
cp gpurun_out/kstats_r03_c5.txt profiles/r03_rocprof_kernel_stats_c5.txt
cp gpurun_out/r03_c3.json profiles/r03_c3_through_class.json
cp gpurun_out/r03_c5.json profiles/r03_c5_through_class.json
cp gpurun_out/r03_rule_n.json profiles/r03_rule_n_single_gpu.json
[ -f gpurun_out/r03_trd_phase_summary.txt ] && cp gpurun_out/r03_trd_phase_summary.txt profiles/r03_trd_resident_phase_summary.txt
[ -f gpurun_out/r03_ph_real2920.txt ] && cp gpurun_out/r03_ph_real2920.txt profiles/r03_trd_resident_phase_stamps_n2920.txt
[ -f gpurun_out/r03_ph_cplx2501.txt ] && cp gpurun_out/r03_ph_cplx2501.txt profiles/r03_trd_resident_phase_stamps_n2501c.txt
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
have = json.load(open("profiles/r03_pmc_c2.json"))["csrc_sha16"]
print("csrc", bench.csrc_hash(), "pmc", have, "OK" if have == bench.csrc_hash() else "STALE: rerun scripts/r03_profiles.sh")
PY
