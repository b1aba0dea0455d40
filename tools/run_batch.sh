#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
for v in "0 --no-graph" "1 --no-graph" "0 " "0 --no-graph" "1 --no-graph"; do
set -- $v
IIR_CU_SPLIT=$1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --no-vae --no-e2e --steps 20 $2 2>/tmp/err.log | tail -1 > /tmp/b.json
python -c "import json; d=json.load(open('/tmp/b.json')); print('cu_split $1 $2', d['ms_per_step'], d['config']['finite'])" || tail -5 /tmp/err.log
done
