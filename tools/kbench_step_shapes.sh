#!/bin/bash
# GEMM microbenchmark on the step's own shapes: auto tile, the explicit tiles, hipBLASLt (torch.matmul) beside them.
TILES=0,24,25,35,27 SHAPES=2048x3840x1280,2048x1280x1280,2048x10240x1280,2048x1280x5120,8192x1920x640,8192x640x640,8192x5120x640,8192x640x2560,4096x10240x1280,16384x5120x640,8192x8192x8192 python tools/kbench.py gemm
