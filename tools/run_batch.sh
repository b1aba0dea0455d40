#!/bin/bash
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "cross_attention_epilogue" 2>&1 | tail -2
timeout -k 10 300 python tools/xattn_fused_bench.py 2>&1 | grep "^R=" | tee gpurun_out/r3/xattn_fused_bench.log
timeout -k 10 900 python tools/racecheck_concurrent.py > gpurun_out/r3/racecheck.log 2>&1 || true
tail -3 gpurun_out/r3/racecheck.log; grep -c "nondeterministic_runs=0/" gpurun_out/r3/racecheck.log; grep "cross-attention epilogue" gpurun_out/r3/racecheck.log
