#!/bin/bash
mkdir -p gpurun_out/r2l
timeout 200 python tools/cubic_group_probe.py 2>&1 | tee gpurun_out/r2l/cubic_group_probe.txt
exit 0
