#!/bin/bash
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
bash tools/final_profiles.sh
python bench.py --config 4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final/bench_config4_fp8.json
python bench.py --config 4 --fp16 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final/bench_config4_fp16.json
python -c "
import json
for n in ('fp8','fp16'):
    d=json.load(open('gpurun_out/final/bench_config4_%s.json'%n)); print('config4', n, d['ms_per_step'], d['value'], d['unit'])"
