#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "fp8" 2>&1 | tail -8
(F8=1 W8=1 TILES=0 SHAPES=2048x1280x1280,2048x1280x5120,2048x3840x1280,8192x640x640,8192x1920x640,8192x640x2560,4096x1280x1280,4096x1280x5120,16384x640x640,16384x640x2560 timeout -k 10 300 python tools/kbench.py gemm; GEGLU=1 F8=1 W8=1 TILES=0 SHAPES=2048x10240x1280,4096x10240x1280,16384x5120x640 timeout -k 10 300 python tools/kbench.py gemm) 2>&1 | grep "^gemm" | tee gpurun_out/r3/kbench_f8.log
