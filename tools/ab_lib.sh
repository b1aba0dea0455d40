#!/bin/bash
# A/B two builds of the library on ONE box (box-to-box spread is ~1.5 ms per step): tools/probes/bin/lib_prev.so vs the in-tree one.
# usage (on the GPU box): bash tools/ab_lib.sh [bench args]
set -e
cd "$(dirname "$0")/.."
cp instantir_amd/libinstantir_hip.so /tmp/lib_new.so
for round in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/probes/bin/lib_prev.so instantir_amd/libinstantir_hip.so; else cp /tmp/lib_new.so instantir_amd/libinstantir_hip.so; fi
    python bench.py --no-cpu-baseline --no-vae --steps 20 "$@" 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('$which', d['ms_per_step'], d['value'], "")"
  done
done
cp /tmp/lib_new.so instantir_amd/libinstantir_hip.so
