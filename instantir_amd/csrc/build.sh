#!/bin/bash
# Build libinstantir_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libinstantir_hip.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -w"
mkdir -p build
pids=()
for f in gemm_conv gemm8 norm; do
  hipcc $FLAGS -c $f.hip -o build/$f.o &
  pids+=($!)
done
# attention: keep MFMA accumulators in VGPRs (softmax reads/rescales them every tile; AGPR form costs
# a v_accvgpr_read/write pair per touched element).  -fno-honor-nans: the row maxima are fmaxf over MFMA outputs; with NaNs
# honoured hipcc canonicalises every operand first (one extra v_max per score); -inf (masking) stays legal.
hipcc $FLAGS -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans -c attention.hip -o build/attention.o &
pids+=($!)
hipcc $FLAGS -ffp-contract=off -c pointwise.hip -o build/pointwise.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT build/gemm_conv.o build/gemm8.o build/attention.o build/norm.o build/pointwise.o
echo "built $(realpath $OUT)"
