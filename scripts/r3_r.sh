#!/bin/bash
# round 3: shared top on the four-wave form — shard tests, then the 2-rank rehearsal on the one GPU (gloo), before/after is in the record
O=gpurun_out/r3r; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "shard or windows or dist or cfg5" > $O/pytest_shard.log 2>&1; rc=$?; echo "pytest exit=$rc" | tee -a $O/pytest_shard.log
tail -4 $O/pytest_shard.log
[ $rc -eq 0 ] || exit $rc
GS_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 5 > $O/bench2.json 2> $O/bench2.err; echo "bench2(gloo rehearsal) exit=$?"
tail -c 1500 $O/bench2.json
