#!/usr/bin/env python3
"""bench.py — GraphSLAM Gauss-Newton iterations/s on a synthetic cone track (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--workload cfg4]

A "step" is ONE full Gauss-Newton iteration (linearise A5-A7, factorise + solve A8, update A9) of the
hot path over the resident graph.  At N=1 the workload is the configuration the north_star target is
quoted on, 100k poses / 10k cones (BASELINE.json configs[3], "cfg4"); inputs are in HBM before the timed
region.  For N>1 (launched by torch.distributed.run, one rank per GPU over RCCL) every rank owns a pose
window of the same size: weak scaling, value = iterations/s of the whole job x windows.

One JSON line on stdout from rank 0, with `roofline` (edge-linearisation kernel, HIP events in this
process) and `cpu_baseline` (the CPU oracle + the reference's vendored Eigen solver on this host).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(pkg, g, iters):
    """The Eigen CPU path: oracle restatement of the g2o arithmetic + the reference's
    vendored Eigen 3.3.4 SimplicialLDLT/AMD (oracle/_ref) when present, else the oracle's own LDLT.
    Single thread, -O3 -DNDEBUG, no -march.  Bounded sample: `iters` iterations of the SAME graph."""
    from oracle import pyoracle as po
    og = po.OracleGraph()
    og.add_poses(g["pose_est"]); og.add_landmarks(g["lm_est"])
    og.add_odometry_edges(g["pp_i"], g["pp_j"], g["pp_z"], g["pp_info"])
    og.add_observation_edges(g["pl_p"], g["pl_l"], g["pl_z"], g["pl_info"])
    for i in g["fixed_poses"]:
        og.set_fixed_pose(int(i))
    for l in g["fixed_landmarks"]:
        og.set_fixed_landmark(int(l))
    solver, kind = None, "port"
    if po.ref_eigen() is not None:
        solver = po.EigenSolver(0)
    t0 = time.perf_counter()
    done, chi, tm = og.optimize(iters, ordering=1, solver=solver)
    wall = time.perf_counter() - t0
    analyze_ms = float(solver.timings()[0]) if solver is not None else 0.0
    per_iter_s = (wall - analyze_ms * 1e-3) / max(done, 1)      # symbolic analysis is iteration-0 work, like the GPU plan
    return og, dict(value=1.0 / per_iter_s, unit="GN iterations/s", cores=1, kind=kind,
                    sample="%d GN iterations of the same %d-pose / %d-cone graph, single thread; g2o arithmetic restated "
                           "in C (oracle/), linear solve = %s; symbolic analysis (%.0f ms) excluded like the GPU plan build"
                           % (done, len(g["pose_est"]), len(g["lm_est"]),
                              "reference's vendored Eigen 3.3.4 SimplicialLDLT+AMD (oracle/_ref)" if solver is not None
                              else "oracle's own up-looking LDLT (oracle/_ref absent)", analyze_ms),
                    ms_linearize=float(tm[0]) / max(done, 1), ms_solve=float(tm[2] - analyze_ms) / max(done, 1),
                    host_cpus=os.cpu_count())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg4")
    ap.add_argument("--cpu-iters", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch                                   # first: its HIP runtime is the one the process uses
    import numpy as np
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU: the GraphSLAM back-end has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")

    N, M = pkg.track.CONFIGS[args.workload]
    track = pkg.track.generate(N, M)
    fe = pkg.Graph(device=local)
    g = pkg.track.bench_graph(track, fe)           # A0 on the device
    fe.close()
    G = pkg.Graph(device=local)
    G.load_bench_graph(g)
    G.initialize_optimization()                    # structure phase (iteration-0 work): plan + upload to HBM
    plan = G.stats()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        G.iterate()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        G.iterate()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt * 1e3 / args.steps
    value = world * args.steps / dt                # every rank iterates its own window of N poses (weak scaling)

    # ---- roofline of the edge-linearisation kernel (HIP events on the library's stream, this process)
    phases = G.time_iterations(20)
    lin_ms = G.time_linearize(50)
    alg_bytes = G.linearize_bytes()                # E_pp*152 + E_pl*96 + N*120 + M*64  
    achieved = alg_bytes / (lin_ms * 1e-3) / 1e9
    roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                    traffic=None, kernel="edge linearisation + assembly (A5-A7)", ms_per_launch=lin_ms,
                    algorithmic_bytes=alg_bytes)

    # ---- parity of what was timed: the reference's optimize(10) from the initial estimates vs the oracle
    out = dict(metric="GraphSLAM Gauss-Newton iters/sec at N poses x M cones; pose RMSE vs ref",
               value=value, unit="GN iterations/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=ms_per_step, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64",
               data="synthetic",
               config=dict(workload="%s: %d poses / %d cones closed synthetic cone track, K=8 observations per pose, "
                                    "gauge = first 2 poses + first 2 cones" % (args.workload, N, M),
                           n_poses=N, n_cones=M, n_odometry_edges=G.n_pp, n_observation_edges=G.n_pl,
                           unknowns=3 * plan.n_free_poses + 2 * plan.n_free_landmarks,
                           parallelism="1 pose window per GPU" if world > 1 else "single GPU",
                           fronts=plan.n_fronts, levels=plan.n_levels, max_front=plan.max_front),
               roofline=roofline,
               phases_ms=dict(linearize=phases.ms_linearize, factor=phases.ms_factor, backsolve=phases.ms_backsolve,
                              update=phases.ms_update, structure_once=plan.ms_structure))
    if rank == 0 and not args.no_cpu:
        og, cb = cpu_baseline(pkg, g, args.cpu_iters)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_baseline"] = value / world / cb["value"]
        # same number of iterations from the same initial estimates on the GPU
        G2 = pkg.Graph(device=local); G2.load_bench_graph(g)
        done, st = G2.optimize(args.cpu_iters)
        P, Lm = G2.poses(), G2.landmarks()
        rms = float(np.sqrt((og.poses()[:, :2] ** 2).sum(1).mean()))
        out["pose_rmse_vs_oracle_rel"] = float(np.sqrt(((P[:, :2] - og.poses()[:, :2]) ** 2).sum(1).mean()) / rms)
        out["landmark_rmse_vs_oracle_rel"] = float(np.sqrt(((Lm - og.landmarks()) ** 2).sum(1).mean()) / rms)
        out["heading_max_abs_diff_vs_oracle"] = float(np.abs(P[:, 2] - og.poses()[:, 2]).max())
        out["parity_iterations"] = int(done)
        G2.close()
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    G.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
