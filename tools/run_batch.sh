mkdir -p gpurun_out/r3
./tools/probes/bin/fill_probe co > gpurun_out/r3/fill_probe_co3.log 2>&1
cat gpurun_out/r3/fill_probe_co3.log
