#!/bin/bash
# bench.py with several rank PROCESSES on the one GPU of a gpurun box (gloo carries the exchange; at most 6 processes on the card: the largest rehearsal has 5 ranks, and the runs wait for each other's processes to be gone): what a multi-GPU run
# does on the host side — rank-local ingestion, shard plans, the failure report across ranks — without the hardware.  Flags DO arrive late here (the
# processes are time-sliced): the runs exercise the timeout report / fallback / re-measure path as a matter of course.
O=gpurun_out/rehearse; mkdir -p $O
run() { name=$1; shift; n=$1; shift; GS_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus $n "$@" > $O/$name.json 2> $O/$name.err; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$O/$name.json")); print("$name: exit $rc  %.0f it/s  n_gpus %d  %s  structure %.1f ms  %s  retries %d" % (d["value"], d["n_gpus"], d["scaling"], d.get("structure_ms_slowest_rank", 0), d.get("ingestion", "")[:40], open("$O/$name.err").read().count("measuring again") // max(d["n_gpus"], 1)))
except Exception as e:
    print("$name: exit $rc  NO LINE:", e)
PY
  sleep 3
}
run weak4_cfg3 4 --workload cfg3 --steps 30 --warmup 5
run weak5_cfg3 5 --workload cfg3 --steps 30 --warmup 5      # (at most 6 processes may have the card open on a gpurun box: 5 ranks leave room for one that is still exiting)
run strong4_cfg4 4 --workload cfg4 --shard --steps 30 --warmup 5
run weak3_cfg4 3 --workload cfg4 --steps 20 --warmup 5
GS_BENCH_FULL_INGEST=1 run weak4_cfg3_full_ingest 4 --workload cfg3 --steps 30 --warmup 5
