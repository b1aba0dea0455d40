#!/bin/bash
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_pipeline_gpu.py tests/test_goldens_gpu.py -x -q > gpurun_out/r3/pipe_tests.log 2>&1 || true
tail -4 gpurun_out/r3/pipe_tests.log
for c in 1 0; do
IIR_LOOP_CACHE=$c python bench.py --no-cpu-baseline --no-roofline --steps 10 2>/dev/null | tail -1 > /tmp/b.json
python -c "import json; d=json.load(open('/tmp/b.json')); print('cache $c', d['ms_per_step'], d['config']['end_to_end']['seconds_per_image'], d['config']['images_per_s_measured'])"
done
