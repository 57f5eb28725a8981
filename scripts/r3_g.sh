#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "shard or cfg5" > $O/pytest_shard.log 2>&1; echo "pytest shard exit=$?" | tee -a $O/pytest_shard.log
tail -5 $O/pytest_shard.log
timeout -k 10 600 python scripts/shard_footprint.py 8 cfg4 2>&1 | tee $O/shard_footprint.txt
