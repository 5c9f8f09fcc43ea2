#!/bin/bash
# GPU box: the c4 form of bench.py with two ranks on one GPU over gloo (BWT_BENCH_REHEARSAL=1),
# on a smaller genome so that two replicas fit beside each other; then the c2 form the same way.
set -u
OUT=gpurun_out/$1; mkdir -p $OUT
export BWT_BENCH_REHEARSAL=1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --config c4 --genome-bp 250000000 --reads 20000000 --steps 3 --warmup 1 > $OUT/c4_two_ranks.json 2> $OUT/c4_two_ranks.err
echo "c4 rc=$?"; tail -c 600 $OUT/c4_two_ranks.err
python - <<PY
import json
for line in open("$OUT/c4_two_ranks.json"):
    if line.startswith("{"):
        d = json.loads(line); print({k: d[k] for k in ("value", "n_gpus", "scaling", "ms_per_step", "checks")})
PY
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 \
  bench.py --gpus 2 --steps 5 --warmup 1 > $OUT/c2_two_ranks.json 2> $OUT/c2_two_ranks.err
echo "c2 rc=$?"; tail -c 600 $OUT/c2_two_ranks.err
python - <<PY
import json
for line in open("$OUT/c2_two_ranks.json"):
    if line.startswith("{"):
        d = json.loads(line); print({k: d[k] for k in ("value", "n_gpus", "scaling", "ms_per_step")}); print(json.dumps(d["sa_build_sharded"])[:4000])
PY
