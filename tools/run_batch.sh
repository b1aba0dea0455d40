mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm_plain or conv3x3 or layernorm_fold or transposed" > $O/pytest_g2.log 2>&1
tail -4 $O/pytest_g2.log
TILES=0,55,65 SHAPES=2048x1280x1280,2048x1280x5120,2048x1280x2560,4096x1280x1280 timeout -k 10 200 python tools/kbench.py gemm 2>&1 | grep -v amdgpu > $O/kb_g2.log
TILES=0,55,65 timeout -k 10 200 python tools/kbench.py conv 2>&1 | grep -v amdgpu >> $O/kb_g2.log
TILES=0,55,65 timeout -k 10 200 python tools/tilebench.py 2>&1 | grep -v amdgpu >> $O/kb_g2.log
cat $O/kb_g2.log
