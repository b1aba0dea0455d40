#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py tests/test_engine_gpu.py tests/test_kernels_gpu.py -x -q -k "fp8 or gemm8" 2>&1 | tail -5
(GEGLU=1 F8=1 TILES=0,91 SHAPES=2048x10240x1280,4096x10240x1280,16384x5120x640 timeout -k 10 300 python tools/kbench.py gemm) 2>&1 | grep "^gemm"
for c in 0 1; do
IIR_G8=$c timeout -k 10 300 python bench.py --config 4 --no-cpu-baseline 2>/tmp/err.log | tail -1 > gpurun_out/r3/config4_g8$c.json
python -c "import json; d=json.load(open('gpurun_out/r3/config4_g8$c.json')); print('config4 fp8, 8-wave kernel $c', d['ms_per_step'], d.get('config',{}).get('finite'))" || tail -5 /tmp/err.log
done
