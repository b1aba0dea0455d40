mkdir -p gpurun_out/r3
timeout -k 10 200 python tools/g8_bench.py > gpurun_out/r3/g8_2.log 2>&1
IIR_ATTN_PRE=0 timeout -k 10 100 python tools/xattn_bench.py > gpurun_out/r3/xattn_ring.log 2>&1
timeout -k 10 100 python tools/xattn_bench.py > gpurun_out/r3/xattn_pre.log 2>&1
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/r3/pytest_kernels.log 2>&1
for v in "IIR_G8=0 IIR_ATTN_PRE=0" "IIR_G8=1 IIR_ATTN_PRE=0" "IIR_G8=0 IIR_ATTN_PRE=1" "IIR_G8=1 IIR_ATTN_PRE=1"; do
  env $v python bench.py --no-cpu-baseline --no-vae --no-roofline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])" >> gpurun_out/r3/ab2.log
done
cat gpurun_out/r3/ab2.log
