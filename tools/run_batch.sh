#!/bin/bash
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r3/full_suite.log 2>&1 || true
tail -5 gpurun_out/r3/full_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3/smoke.log 2>&1 && tail -1 gpurun_out/r3/smoke.log
