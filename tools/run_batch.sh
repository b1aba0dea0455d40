#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
for xm in 0 1 2 4 8 0; do
  echo "== IIR_XM=$xm"
  IIR_XM=$xm TILES=0 SHAPES=2048x1280x1280,2048x1280x5120,8192x640x640,8192x640x2560,2048x3840x1280 timeout -k 10 200 python tools/kbench.py gemm 2>&1 | grep "^gemm"
done > gpurun_out/r3/xm.log 2>&1
cat gpurun_out/r3/xm.log
