#!/bin/bash
# attention microbenchmark on two builds of the library (tools/probes/bin/lib_prev.so vs in-tree), same box
set -e
cd "$(dirname "$0")/.."
cp instantir_amd/libinstantir_hip.so /tmp/lib_new.so
for which in prev new prev new; do
  if [ $which = prev ]; then cp tools/probes/bin/lib_prev.so instantir_amd/libinstantir_hip.so; else cp /tmp/lib_new.so instantir_amd/libinstantir_hip.so; fi
  echo "== $which"; python tools/attnbench.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_new.so instantir_amd/libinstantir_hip.so
