#!/bin/bash
# builds scripts/probes/gemm_probe[_suffix] and prints registers / scratch per kernel:  build_gemm_probe.sh [suffix] [extra hipcc flags]
cd "$(dirname "$0")/../.."
suf=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include "$@" scripts/probes/gemm_probe.cpp -o scripts/probes/gemm_probe$suf -save-temps=obj \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|ScratchSize|VGPRs Spill" | grep -v "fill_kernel" | \
  sed -e 's/remark: [^ ]* *//' -e 's/\[-Rpass.*//' | paste - - - - | grep gemm_kernel | sed -e 's/Function Name: _ZN4xmca11gemm_kernelI//' -e 's/EEEvNS_10GemmParamsIT_T0_EE//'
