#!/bin/bash
OUT=gpurun_out/r2h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-prof > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "n1 rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench_n1.json').read().strip().splitlines()[-1]); print('N=1 ms', round(d['ms_per_step'],2), 'slab', {k: d['slab_mode'].get(k) for k in ('n_gpus','ms_per_proof','rccl_ranks','error','proof_sha256')})"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --backend gloo --log-s 20 --steps 2 --warmup 1 --no-cpu-baseline --no-prof --concurrent 0 --slab-kind and --slab-c 2 --slab-log-s 20 > $OUT/bench_2ranks.json 2> $OUT/bench_2ranks.err; echo "2-rank rc=$?"; tail -2 $OUT/bench_2ranks.err; python -c "
import json; d=json.loads(open('$OUT/bench_2ranks.json').read().strip().splitlines()[-1]); print('2 ranks value', d['value'], 'slab', {k: d['slab_mode'].get(k) for k in ('n_gpus','ms_per_proof','rccl_ranks','error','proof_sha256','exchange')})"
timeout 100 python bench.py --log-s 20 --steps 2 --warmup 1 --no-cpu-baseline --no-prof --concurrent 0 --slab-kind and --slab-c 2 --slab-log-s 20 > $OUT/bench_1rank_ref.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_1rank_ref.json').read().strip().splitlines()[-1]); print('1 rank ref slab sha', d['slab_mode'].get('proof_sha256'), d['slab_mode'].get('ms_per_proof'))"
# a failing leg must not take the line: an impossible strategy makes the worker exit non-zero
timeout 100 python bench.py --log-s 18 --steps 1 --warmup 1 --no-cpu-baseline --no-prof --concurrent 0 --slab-kind and --slab-c 9 --slab-log-s 12 > $OUT/bench_bad_leg.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_bad_leg.json').read().strip().splitlines()[-1]); print('bad leg: value present', d['value'] > 0, 'slab', str(d['slab_mode'])[:160])"
exit 0
