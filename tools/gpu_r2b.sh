#!/bin/bash
# Round 2, second box visit: the whole -m gpu suite on HEAD (oracle threads fixed: expect minutes, not 19), the full default bench line (all-core CPU baseline,
# parity check, slab leg at N = 1), the N > 1 bench code path with two ranks sharing the one GPU over gloo (shared-memory exchange; RCCL must decline
# consistently: duplicate GPU), LT C=16 (configs[4]'s shape) and the proof artefact at the metric size.
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log; grep -E "^\[oracle|^\[verify" $OUT/pytest_gpu.log
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r2b/bench.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"]); print("cpu", d.get("cpu_baseline")); print("parity", d.get("parity_checked")); print("slab", d.get("slab_mode"))
    print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "achieved_traffic", "frac_traffic")}); print("msm", {k: (v or {}).get("frac") for k, v in d["roofline_msm"].items()})
except Exception as e:
    print("bench parse failed", e)
PY
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --log-s 20 --steps 2 --warmup 1 --no-cpu-baseline --no-prof --concurrent 0 --slab-kind and --slab-c 2 --slab-log-s 20 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err; echo "2-rank rc=$?"; tail -2 $OUT/bench_2ranks_gloo.err; python -c "
import json; d=json.loads(open('$OUT/bench_2ranks_gloo.json').read().strip().splitlines()[-1]); print('2 ranks: value', d['value'], 'distinct', d['config']['distinct_proofs'], 'slab', d.get('slab_mode'))"
timeout 120 python bench.py --slab-kind and --slab-c 2 --slab-log-s 20 --log-s 20 --steps 2 --no-cpu-baseline --no-prof --concurrent 0 > $OUT/bench_1rank_same_slab.json 2> /dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_1rank_same_slab.json').read().strip().splitlines()[-1]); print('1 rank slab ref', d.get('slab_mode'))"
for ls in 20 24; do timeout 300 python bench.py --kind lt --c 16 --log-s $ls --steps 2 --warmup 1 --no-cpu-baseline --concurrent 0 --no-slab-leg > $OUT/bench_lt_c16_2p$ls.json 2> $OUT/bench_lt_c16_2p$ls.err; echo "lt c16 2^$ls rc=$? $(python -c "import json;d=json.loads(open('$OUT/bench_lt_c16_2p$ls.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])")"; done
timeout 200 python tools/dump_proof.py $OUT/artefact_and_c1_2p24 --log-s 24 > $OUT/dump_proof.log 2>&1; tail -c 600 $OUT/dump_proof.log; rm -f $OUT/artefact_and_c1_2p24/*.bin.tmp; ls -la $OUT/artefact_and_c1_2p24
ls $OUT
exit 0
