#!/bin/bash
# Run on the GPU box: scripts/try_variants.sh "<variant names>" <command...>  - runs the command once per library
# variant in scripts/variants/ (libxmca_<name>.so copied over the in-tree library, restored afterwards).
names=$1; shift
cp xmca_amd/libxmca_hip.so /tmp/xmca_keep.so
for v in $names; do
  echo "== variant $v"
  cp scripts/variants/libxmca_$v.so xmca_amd/libxmca_hip.so; touch xmca_amd/libxmca_hip.so
  "$@"
done
cp /tmp/xmca_keep.so xmca_amd/libxmca_hip.so
