#!/bin/bash
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r3/full_suite.log 2>&1 || true
tail -5 gpurun_out/r3/full_suite.log
bash tools/final_profiles.sh
