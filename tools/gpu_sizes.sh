#!/bin/bash
# headline metric at other lookup counts (AND, C=1): one line per size
for ls in "$@"; do
  timeout 400 python bench.py --log-s $ls --steps 3 --warmup 1 --no-cpu-baseline --no-prof --concurrent 0 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('log_s', $ls, round(d['ms_per_step'],2), 'ms', round(d['value']/1e6,1), 'M lookups/s')"
done
