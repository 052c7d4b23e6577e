#!/bin/bash
# closing visit: 4 ranks sharing the one GPU (gloo) through bench.py's N > 1 path incl. the slab leg as child processes, then the whole validation
OUT=gpurun_out/r2q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 4 --backend gloo --log-s 20 --steps 2 --warmup 1 --no-cpu-baseline --no-prof --concurrent 0 --slab-kind range --slab-c 2 --slab-log-s 20 > $OUT/bench_4ranks.json 2> $OUT/bench_4ranks.err; echo "4-rank rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench_4ranks.json').read().strip().splitlines()[-1]); print('4 ranks value', d['value'], 'distinct proofs', d['config']['distinct_proofs'], 'slab', {k: d['slab_mode'].get(k) for k in ('n_gpus','ms_per_proof','rccl_ranks','error','proof_sha256')})"
timeout 100 python bench.py --log-s 20 --steps 2 --warmup 1 --no-cpu-baseline --no-prof --concurrent 0 --slab-kind range --slab-c 2 --slab-log-s 20 > $OUT/bench_1rank_ref.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_1rank_ref.json').read().strip().splitlines()[-1]); print('1 rank ref slab sha', d['slab_mode'].get('proof_sha256'))"
bash tools/gpu_r2g.sh
exit 0
