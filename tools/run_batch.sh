mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_2.log 2>&1
tail -6 $O/pytest_gpu_2.log
python bench.py --no-cpu-baseline > $O/bench_e2e.json 2> $O/bench_e2e.err
tail -4 $O/bench_e2e.err
python -c "import json; d=json.load(open('$O/bench_e2e.json')); print(d['ms_per_step'], d['config']['images_per_s_30step'], d['config']['end_to_end'], d['config']['ms_per_step_by_rank'])"
REP=12 timeout -k 10 600 python tools/racecheck_concurrent.py > $O/racecheck.log 2>&1
tail -14 $O/racecheck.log
