#!/bin/bash
# Run on the GPU box (gpurun): every artefact of the round under gpurun_out/r02_*; copy what is to be judged into profiles/.
set -u
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
# 1. the bench line (N = 1) and the multi-rank flow on the one GPU of the box
python bench.py --steps 5 --warmup 2 --rule-n-rotated > gpurun_out/r02_bench_c2.json 2> gpurun_out/r02_bench_c2.err
XMCA_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r02_bench_shared_gpu_2.json 2> gpurun_out/r02_bench_shared_gpu_2.err
XMCA_BENCH_SHARE_GPU=1 python bench.py --gpus 4 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r02_bench_shared_gpu_4.json 2> gpurun_out/r02_bench_shared_gpu_4.err
# 2. rocprofv3 kernel stats of the bench command
scripts/profile_cmd.sh r02_c2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rule-n --no-e2e
# 3. the other configurations through the class + their kernel stats
scripts/profile_cmd.sh r02_c3 python scripts/run_config.py C3
cp gpurun_out/prof_r02_c3.out gpurun_out/r02_c3.json
scripts/profile_cmd.sh r02_c4 python scripts/rule_n_bench.py
cp gpurun_out/prof_r02_c4.out gpurun_out/r02_rule_n.json
scripts/profile_cmd.sh r02_c5 python scripts/c5_device_ctor.py
cp gpurun_out/prof_r02_c5.out gpurun_out/r02_c5.json
# 4. PMC passes (separate runs, kernel-trace only)
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  ( cd $REPO && rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_$tag -o g --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rule-n --no-e2e > /dev/null 2>&1 )
done
cd $REPO
python - <<'PY' > gpurun_out/r02_pmc_c2.json
import csv, collections, glob, json
out = {}
for tag in ["SQ_WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE"]:
    for f in glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            k = "jacobi_fused_round_kernel<64,false>" if ("jacobi_fused_round" in name and "64" in name) else ("gemm_kernel" if "gemm_kernel" in name else ("gemm_nt_kernel" if "gemm_nt" in name else ("splitk_reduce" if "splitk" in name else ("varimax_persistent" if "varimax_persistent" in name else None))))
            if k is None: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
        for k in agg:
            d = out.setdefault(k, {"dispatches": len(cnt[k])})
            for c, v in agg[k].items(): d[c + "_per_launch"] = v / len(cnt[k])
for k, d in out.items():
    if "FETCH_SIZE_per_launch" in d and "WRITE_SIZE_per_launch" in d:
        d["hbm_side_bytes_per_launch_gfx950_corrected"] = (2 * d["FETCH_SIZE_per_launch"] + d["WRITE_SIZE_per_launch"]) * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES_per_launch" in d and "SQ_BUSY_CYCLES_per_launch" in d:
        d["mfma_busy_fraction"] = d["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] / (1024 * d["SQ_BUSY_CYCLES_per_launch"] / 32)
out["_command"] = "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rule-n --no-e2e (three separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md)"
print(json.dumps(out, indent=1))
PY
ls -la gpurun_out/r02_* gpurun_out/kstats_r02_*
