#!/bin/bash
# round 4, batch Y: the randomised stress scripts on the last code (the planner changed: one edge pass, parallel numbering, balanced adjacency, window-only odometry records)
O=gpurun_out/r4y; mkdir -p $O
timeout -k 10 300 python scripts/stress_gpu.py 600 60 2>&1 | tail -1 | tee $O/stress_gpu.txt
timeout -k 10 300 python scripts/stress_gpu.py 700 30 fat 2>&1 | tail -1 | tee -a $O/stress_gpu.txt
timeout -k 10 300 python scripts/stress_growth.py 100 30 2>&1 | tail -1 | tee $O/stress_growth.txt
timeout -k 10 300 python scripts/stress_failures.py 100 40 2>&1 | tail -1 | tee $O/stress_failures.txt
timeout -k 10 300 python scripts/stress_slam.py 100 30 2>&1 | tail -1 | tee $O/stress_slam.txt
timeout -k 10 300 python scripts/stress_threads.py 4 2>&1 | tail -1 | tee $O/stress_threads.txt
