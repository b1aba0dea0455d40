mkdir -p gpurun_out/r3
O=gpurun_out/r3
rm -f $O/ab5.log
run() { env $1 python bench.py --no-cpu-baseline --no-vae --no-roofline --steps 20 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['ms_per_step'])" >> $O/ab5.log; }
run "IIR_ST_WT=0" ""; run "IIR_ST_WT=1" ""; run "IIR_ST_WT=0" ""; run "IIR_ST_WT=1" ""
run "IIR_ST_WT=0" "--no-overlap"; run "IIR_ST_WT=1" "--no-overlap"
cat $O/ab5.log
IIR_ST_WT=1 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -2
IIR_ST_WT=1 TILES=0 SHAPES=2048x1280x1280,8192x640x640,2048x10240x1280 timeout -k 10 100 python tools/kbench.py gemm 2>&1 | grep -v amdgpu
TILES=0 SHAPES=2048x1280x1280,8192x640x640,2048x10240x1280 timeout -k 10 100 python tools/kbench.py gemm 2>&1 | grep -v amdgpu
