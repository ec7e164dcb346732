#!/bin/bash
# Run on the GPU box (gpurun): the artefacts of round 6 under gpurun_out/r06_*; copy what is to be judged into profiles/
# (scripts/copy_r06_profiles.sh).   scripts/r06_profiles.sh [quick]   quick = bench + C2 kernel stats + the PMC passes only
set -u
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
# 1. the bench line (N = 1)
python bench.py --steps 10 --warmup 3 > gpurun_out/r06_bench_c2.json 2> gpurun_out/r06_bench_c2.err
# 2. rocprofv3 kernel stats of the bench command (C2 step only)
scripts/profile_cmd.sh r06_c2 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-rule-n --no-e2e --no-c5
# 3. PMC passes (separate runs, kernel-trace only): the C2 step, and the Gram product alone at C2 and C5
pmc() {   # pmc <tag> <command...>
  tag=$1; shift
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    p=$(echo $pass | cut -d' ' -f1)
    rm -rf /tmp/pmc_${tag}_$p
    ( cd /tmp && rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_${tag}_$p -o g --output-format csv -- "$@" > /dev/null 2>&1 )
  done
  python - "$tag" "$*" <<'PY' > gpurun_out/r06_pmc_$tag.json
import csv, collections, glob, json, sys
sys.path.insert(0, ".")
import bench
tag, cmd = sys.argv[1], sys.argv[2]
out = {}
def key(name):
    for k in ("trd_resident_kernel", "trd_step_kernel", "trd_bisect_kernel", "trd_twisted_kernel", "trd_wy_tinv_kernel", "gemm_kernel",
              "varimax_persistent", "jacobi_fused_round"):
        if k in name: return k
    return None
for p in ["SQ_WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE"]:
    for f in glob.glob("/tmp/pmc_%s_%s/**/*counter_collection.csv" % (tag, p), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = key(r["Kernel_Name"])
            if k is None: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
        for k in agg:
            d = out.setdefault(k, {"dispatches": len(cnt[k])})
            for c, v in agg[k].items(): d[c + "_per_launch"] = v / len(cnt[k])
for k, d in out.items():
    if "FETCH_SIZE_per_launch" in d and "WRITE_SIZE_per_launch" in d:
        d["hbm_side_bytes_per_launch_gfx950_corrected"] = (2 * d["FETCH_SIZE_per_launch"] + d["WRITE_SIZE_per_launch"]) * 1024
        # (round 6: a reduction is a CHAIN of trd_resident_kernel launches - bench.py multiplies the per-launch mean by the number
        #  of links, xmca_get_reduction_info, to get the bytes of one reduction)
        d["hbm_side_bytes_all_dispatches"] = d["hbm_side_bytes_per_launch_gfx950_corrected"] * d["dispatches"]
    if "SQ_VALU_MFMA_BUSY_CYCLES_per_launch" in d and "SQ_BUSY_CYCLES_per_launch" in d:
        d["mfma_busy_fraction"] = d["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] / (1024 * d["SQ_BUSY_CYCLES_per_launch"] / 32)
out["csrc_sha16"] = bench.csrc_hash()
out["_command"] = "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- %s (three separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md)" % cmd
print(json.dumps(out, indent=1))
PY
}
pmc c2 python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rule-n --no-e2e --no-c5
pmc gram_c2 python $REPO/scripts/gram_only.py c2 3
pmc gram_c5 python $REPO/scripts/gram_only.py c5 2
if [ "${1:-}" != "quick" ]; then
  # 4. the other configurations through the class + their kernel stats
  scripts/profile_cmd.sh r06_c3 python scripts/run_config.py C3
  cp gpurun_out/prof_r06_c3.out gpurun_out/r06_c3_through_class.json
  scripts/profile_cmd.sh r06_c4 python scripts/rule_n_bench.py
  cp gpurun_out/prof_r06_c4.out gpurun_out/r06_rule_n_single_gpu.json
  scripts/profile_cmd.sh r06_c5 python scripts/c5_device_ctor.py
  cp gpurun_out/prof_r06_c5.out gpurun_out/r06_c5_through_class.json
fi
if [ "${1:-}" != "quick" ]; then
  # 5. lanes sweep, host budget, determinism soak + chain timings
  python scripts/lanes_sweep.py > gpurun_out/r06_lanes_sweep.txt 2>&1
  python scripts/host_budget.py > gpurun_out/r06_host_budget.json 2> gpurun_out/r06_host_budget.err
  bash scripts/r06_trd_determinism.sh > /dev/null 2>&1
fi
ls -la gpurun_out/r06_* gpurun_out/kstats_r06_*
