#!/bin/bash
# HBM-side traffic of the C2 Gram GEMM (run on the GPU box): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes.
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_g1 gpurun_out/pmc_g2
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_g1 -o g --output-format csv -- python scripts/gemm_one.py 2920 2920 10000 f64 1 0 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_g2 -o g --output-format csv -- python scripts/gemm_one.py 2920 2920 10000 f64 1 0 1 > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob, json
out = {}
for d, c in (("gpurun_out/pmc_g1", "FETCH_SIZE"), ("gpurun_out/pmc_g2", "WRITE_SIZE")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = "gemm_kernel" if "gemm_kernel" in r["Kernel_Name"] else ("splitk_reduce" if "splitk" in r["Kernel_Name"] else None)
            if k and r["Counter_Name"] == c:
                agg[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k in agg:
            out["%s_KB_per_launch_%s" % (c, k)] = agg[k] / len(n[k])
out["hbm_bytes_per_launch_gemm_kernel_corrected"] = (2 * out.get("FETCH_SIZE_KB_per_launch_gemm_kernel", 0) + out.get("WRITE_SIZE_KB_per_launch_gemm_kernel", 0)) * 1024
print(json.dumps(out, indent=1))
PY
