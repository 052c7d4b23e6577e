#!/bin/bash
# sweep an env knob over the traced bench: usage gpu_sweep.sh VAR v1 v2 ...
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  LASSO_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > /tmp/b.json 2> /tmp/b.err
  echo "== $VAR=$v: $(python -c "import json;print(json.load(open('/tmp/b.json'))['ms_per_step'])") ms; $(grep 'DotProductProofLog' /tmp/b.err | tail -4 | awk '{print $3}' | tr '\n' ' ')"
done
