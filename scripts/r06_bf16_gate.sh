#!/bin/bash
# VERDICT r05 #4 gate, answered with a kernel: the C5 Gram product (T = 1200 x N = 1 036 800 float32) by the shipped bf16x3 kernel, and by
# the SAME kernel with the three-way VALU split taken out of the k-loop (gpurun_tmp_libs/lib_nosplit.so, built with
# -DXMCA_X3_NOSPLIT_EXPERIMENT: wrong numbers, same six MFMAs per product, same LDS traffic of 4 bytes per element - pre-split planes
# would move 6).  The second number bounds from BELOW what pre-split bfloat16 planes can take with the 128 x 128 tile pipeline.
out=gpurun_out/r06_bf16_presplit_gate.txt; : > $out
cp xmca_amd/libxmca_hip.so /tmp/m.so
for v in main nosplit; do
  if [ $v = main ]; then cp /tmp/m.so xmca_amd/libxmca_hip.so; else cp gpurun_tmp_libs/lib_$v.so xmca_amd/libxmca_hip.so; fi
  echo "== $v" >> $out
  python scripts/c5_gram_bench.py 2 >> $out 2>&1
  python scripts/c5_device_ctor.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({'solve_s': d['solve_s'], 'stages_ms': d['stages_ms']})" >> $out
done
cp /tmp/m.so xmca_amd/libxmca_hip.so

# rocprofv3 kernel stats of the two builds (the Gram launch)
for v in main nosplit; do
  if [ $v = main ]; then cp /tmp/m.so xmca_amd/libxmca_hip.so; else cp gpurun_tmp_libs/lib_$v.so xmca_amd/libxmca_hip.so; fi
  scripts/profile_cmd.sh r06_bf16_$v python scripts/c5_gram_bench.py 1
  echo "== rocprofv3 --kernel-trace --stats, $v" >> $out
  grep "gemm_kernel\|total kernel" gpurun_out/kstats_r06_bf16_$v.txt >> $out
done
cp /tmp/m.so xmca_amd/libxmca_hip.so

cat $out
