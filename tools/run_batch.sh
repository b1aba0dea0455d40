mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "groupnorm or gemm_plain or conv3x3" > $O/pytest_gn.log 2>&1
tail -15 $O/pytest_gn.log
timeout -k 10 900 python -m pytest tests/test_pipeline_gpu.py tests/test_goldens_gpu.py -x -q -m gpu > $O/pytest_gn2.log 2>&1
tail -5 $O/pytest_gn2.log
rm -f $O/ab6.log
run() { env $1 python bench.py --no-cpu-baseline --no-vae --no-roofline --steps 20 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['ms_per_step'])" >> $O/ab6.log; }
run "IIR_GN_FUSE=0" ""; run "IIR_GN_FUSE=1" ""; run "IIR_GN_FUSE=0" ""; run "IIR_GN_FUSE=1" ""
cat $O/ab6.log
