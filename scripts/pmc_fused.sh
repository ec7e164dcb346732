export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc_f1 -o g --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rule-n > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f2 -o g --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rule-n > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc_f3 -o g --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rule-n > /dev/null 2>&1
python - <<'PY'
import csv, collections
for d in ["gpurun_out/pmc_f1","gpurun_out/pmc_f2","gpurun_out/pmc_f3"]:
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set); dur=collections.defaultdict(float)
    for r in csv.DictReader(open(d+"/g_counter_collection.csv")):
        k=r["Kernel_Name"][:40]
        if "jacobi_fused" not in k and "jacobi_update" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k in agg:
        n=len(cnt[k]); print(d.split('/')[-1], k, "dispatches", n, {c: round(v/n,1) for c,v in agg[k].items()})
PY
