#!/bin/bash
# round 4, batch Z: rank-local ingestion (gs_dist_set_landmark_windows) — the sharded GPU tests, the footprint of a rank of 8 x cfg4 both ways, the 2-rank gloo rehearsal of bench.py
O=gpurun_out/r4z; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "sharded or rank_of_eight or cpp_consumer" > $O/suite.txt 2>&1; rc=$?; tail -3 $O/suite.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python scripts/shard_footprint.py 8 cfg4 | tee $O/shard_footprint.txt
GS_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 5 > $O/bench_2rank_gloo.json 2> $O/bench2.err; echo "gloo2 exit=$?"
python -c "
import json; d=json.load(open('$O/bench_2rank_gloo.json')); print(d['value'], d.get('structure_ms_slowest_rank'), d.get('ingestion'))"
GS_BENCH_FULL_INGEST=1 GS_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 50 --warmup 5 > $O/bench_2rank_gloo_full.json 2> $O/bench2f.err; echo "gloo2 full exit=$?"
python -c "
import json; d=json.load(open('$O/bench_2rank_gloo_full.json')); print(d['value'], d.get('structure_ms_slowest_rank'), d.get('ingestion'))"
