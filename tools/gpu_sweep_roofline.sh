#!/bin/bash
# sweep an env knob and print the bench's roofline numbers: usage gpu_sweep_roofline.sh VAR v1 v2 ...
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --concurrent 0 > /tmp/b.json 2> /tmp/b.err
  echo "== $VAR=$v: $(python -c "import json;d=json.load(open('/tmp/b.json'));r=d['roofline'];b=d['roofline_bind_top'];print(round(d['ms_per_step'],2),'ms | cubic',r['achieved'],'GB/s avg_us',r['avg_launch_us'],'| bind',b['achieved'])")"
done
