#!/bin/bash
# scripts/build_variant.sh <name> [extra hipcc flags...]  ->  scripts/variants/libxmca_<name>.so  (for scripts/try_variants.sh)
set -e
name=$1; shift
mkdir -p scripts/variants
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-unused-variable \
  -Wno-unused-but-set-variable -I include "$@" xmca_amd/csrc/xmca_hip.cpp -o scripts/variants/libxmca_$name.so
echo scripts/variants/libxmca_$name.so
