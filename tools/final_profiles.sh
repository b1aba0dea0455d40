#!/bin/bash
# Round-end measurement artifacts, one call on the GPU box; everything lands in gpurun_out/final/ (copy what is judged into profiles/).
#   bash tools/final_profiles.sh
# rocprofv3 gets the program itself after `--` (python3 bench.py ...), counters in their own passes (no trace domains with --pmc).
set -e
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
echo "[1/5] bench"; python bench.py 2> $O/bench.stderr | tail -1 > $O/bench.json
echo "[2/5] kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep -a '^{"metric' $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json || true
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python tools/trace_by_shape.py $O/trace 0.25 > $O/kernel_time_by_shape.txt
rm -rf $O/trace
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-vae"
echo "[3/5] pmc FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 bench.py $ARGS > $O/pmc_f.log 2>&1
echo "[4/5] pmc WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 bench.py $ARGS > $O/pmc_w.log 2>&1
python tools/pmc_traffic.py $(find $O/pmc_f -name '*counter_collection.csv' | head -1) $(find $O/pmc_w -name '*counter_collection.csv' | head -1) > $O/pmc_traffic.json
rm -rf $O/pmc_f $O/pmc_w
echo "[5/5] pmc MFMA busy"; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -- python3 bench.py $ARGS > $O/pmc_m.log 2>&1
python tools/pmc_mfma_busy.py $(find $O/pmc_m -name '*counter_collection.csv' | head -1) > $O/pmc_mfma_busy.json
rm -rf $O/pmc_m
python -c "import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"
head -12 $O/kernel_time_by_shape.txt
