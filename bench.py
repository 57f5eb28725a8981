#!/usr/bin/env python3
"""bench.py — GraphSLAM Gauss-Newton iterations/s on a synthetic cone track (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--workload cfg4]

A "step" is ONE full Gauss-Newton iteration (linearise A5-A7, factorise + solve A8, update A9) of the hot path
over the resident graph; inputs are in HBM before the timed region.

N = 1: the workload is the configuration the north_star target is quoted on, 100k poses / 10k cones
(BASELINE.json configs[3], "cfg4").
N > 1 (launched by torch.distributed.run, one rank per GPU): ONE graph of N x 100k poses / N x 10k cones is
sharded by contiguous pose window: every rank linearises and factorises its own window, the
contributions to the shared top of the assembly tree (window-boundary poses and the landmarks seen from two
windows — the shared rows of Omega / xi) are summed by one RCCL all-reduce (fp64) per iteration, and every rank
finishes the top redundantly.  Weak scaling: value = N x (iterations/s of the whole graph), i.e. 100k-pose-window
iterations per second, the same unit as the N = 1 line.
`--workload cfg5 --shard` (N > 1) instead shards THE NAMED WORKLOAD: `--gpus 8 --workload cfg5 --shard` is BASELINE config 5
itself, 1M poses / 50k cones by pose window across 8 GPUs (strong scaling: value = iterations/s of that one graph).

One JSON line on stdout from rank 0, with `roofline` (edge-linearisation kernel, HIP events in this process)
and, at N = 1, `cpu_baseline` (the CPU oracle + the reference's vendored Eigen solver on this host).
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


def cpu_baseline(pkg, g, iters, budget_s=22.0):
    """The Eigen CPU path: oracle restatement of the g2o arithmetic + the reference's
    vendored Eigen 3.3.4 SimplicialLDLT/AMD (oracle/_ref) when present, else the oracle's own LDLT.
    Single thread, -O3 -DNDEBUG, no -march.  Bounded sample: `iters` iterations of the SAME graph, one warm-up run and
    then the MEDIAN over up to 5 repeats (as many as fit ~budget_s of CPU work; every repeat starts from the same
    initial estimates on a fresh graph with a fresh solver, i.e. includes its own symbolic analysis, which is
    reported and excluded per repeat like the GPU plan build)."""
    from oracle import pyoracle as po
    import numpy as np

    def one_run():
        og = po.OracleGraph()
        og.add_poses(g["pose_est"]); og.add_landmarks(g["lm_est"])
        og.add_odometry_edges(g["pp_i"], g["pp_j"], g["pp_z"], g["pp_info"])
        og.add_observation_edges(g["pl_p"], g["pl_l"], g["pl_z"], g["pl_info"])
        for i in g["fixed_poses"]:
            og.set_fixed_pose(int(i))
        for l in g["fixed_landmarks"]:
            og.set_fixed_landmark(int(l))
        solver = po.EigenSolver(0) if po.ref_eigen() is not None else None
        t0 = time.perf_counter()
        done, chi, tm = og.optimize(iters, ordering=1, solver=solver)
        wall = time.perf_counter() - t0
        analyze_ms = float(solver.timings()[0]) if solver is not None else 0.0
        return og, dict(wall=wall, analyze_ms=analyze_ms, done=int(done), per_iter_s=(wall - analyze_ms * 1e-3) / max(done, 1),
                        ms_linearize=float(tm[0]) / max(done, 1), ms_solve=float(tm[2] - analyze_ms) / max(done, 1)), solver is not None

    og, warm, eig = one_run()                                    # warm-up (page faults, caches, clocks); also sizes the sample
    reps = int(max(1, min(5, budget_s // max(warm["wall"], 1e-3))))
    runs = []
    for _ in range(reps):
        og, r, eig = one_run(); runs.append(r)
    med = lambda k: float(np.median([r[k] for r in runs]))
    per_iter_s = med("per_iter_s")
    return og, dict(wall_ms_end_to_end=med("wall") * 1e3, ms_symbolic=med("analyze_ms"), value=1.0 / per_iter_s, unit="GN iterations/s", cores=1, kind="port",
                    repeats=reps, warmup_runs=1, value_min=1.0 / max(r["per_iter_s"] for r in runs), value_max=1.0 / min(r["per_iter_s"] for r in runs),
                    sample="median of %d runs (after 1 warm-up run) of %d GN iterations of the same %d-pose / %d-cone graph, single thread; g2o "
                           "arithmetic restated in C (oracle/), linear solve = %s; symbolic analysis (%.0f ms per run) excluded like the GPU plan build"
                           % (reps, runs[0]["done"], len(g["pose_est"]), len(g["lm_est"]),
                              "reference's vendored Eigen 3.3.4 SimplicialLDLT+AMD (oracle/_ref)" if eig
                              else "oracle's own up-looking LDLT (oracle/_ref absent)", med("analyze_ms")),
                    ms_linearize=med("ms_linearize"), ms_solve=med("ms_solve"), host_cpus=os.cpu_count())


def warm_clocks(G, seconds=0.4):
    """Untimed Gauss-Newton iterations for `seconds` before a per-kernel measurement: the clocks of a GPU that has just been idle (plan build, upload) are
    not the ones it runs at under sustained work — measured on MI355X: the linearisation kernel 26.8-27.1 us right after a 0.05 s timed loop, 26.0-26.3 after
    0.5 s of iterations (iterations/s unchanged).  Outside every timed region."""
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            G.iterate()
        G.synchronize()


def linearize_roofline_of(pkg, name, device, reps=10):
    """The roofline kernel (k_linearize_ell) at another configuration, inside full Gauss-Newton iterations (HIP events around
    the phase, gs_time_iterations) and back to back: cfg3 is launch-bound, cfg4 is the headline size, cfg5 (1.05 GB per pass)
    is where the kernel is bound by HBM bandwidth."""
    N, M = pkg.track.CONFIGS[name]
    t = pkg.track.generate(N, M)
    fe = pkg.Graph(device=device); g = pkg.track.bench_graph(t, fe); fe.close()
    G = pkg.Graph(device=device); G.load_bench_graph(g); G.initialize_optimization()
    warm_clocks(G)
    ph = G.time_iterations(reps); b2b = G.time_linearize(reps); B = G.linearize_bytes()
    G.close()
    lin = ph.ms_linearize_kernel if ph.ms_linearize_kernel > 0 else ph.ms_linearize      # the kernel's own begin -> end (events attached to its dispatch)
    inside = B / (lin * 1e-3) / 1e9
    return dict(algorithmic_bytes=B, ms_per_launch=lin, ms_event_to_event=ph.ms_linearize, achieved=inside, frac=inside / HBM_PEAK_GBS,
                frac_event_to_event=B / (ph.ms_linearize * 1e-3) / 1e9 / HBM_PEAK_GBS,
                ms_per_launch_back_to_back=b2b, frac_back_to_back=B / (b2b * 1e-3) / 1e9 / HBM_PEAK_GBS,
                iteration_ms=ph.ms_total, iterations_per_s=1e3 / ph.ms_total)


def association_roofline(pkg, np, track, g, device, reps=20):
    """Batched A1 with everything resident (gs_associate_resident, reference loop src/slam.cpp:570-607 over ALL keyframes at once): the
    map in HBM with its grid built on the device, poses / observations / result in device memory; the query kernel's own begin -> end
    from HIP events attached to its dispatch (gs_debug_time_associate_resident).  Algorithmic bytes: SURVEY 8(d), N K 36 + N 24 + M 20."""
    DA = pkg.binding.DeviceArray
    N, K = len(track["odom_poses"]), track["K"]
    obs = np.ascontiguousarray(track["obs"].reshape(-1, 4)); po_ = np.repeat(np.arange(N, dtype=np.int32), K)
    mxy, mty = np.ascontiguousarray(g["lm_est"]), g["lm_type"].astype(np.int32)
    G = pkg.Graph(device=device); G.map_append(mxy, mty)
    n = len(obs)
    d_p, d_po, d_ob, d_out = DA(track["odom_poses"]), DA(po_), DA(obs), DA(nbytes=4 * n)      # (the map was built from these poses: track.bench_graph)
    t0 = time.perf_counter(); G.associate_resident(d_p, N, d_po, d_ob, n, 1.2, d_out); G.synchronize(); first_ms = (time.perf_counter() - t0) * 1e3
    ms = G.time_associate_resident(d_p, N, d_po, d_ob, n, 1.2, d_out, reps)
    idx = d_out.to_host(np.int32, n)
    for a in (d_p, d_po, d_ob, d_out):
        a.free()
    G.close()
    B = n * 36 + N * 24 + len(mxy) * 20
    return dict(kernel="k_pose_trig + k_associate_grid_dev: cone -> global, then the first match in map order over the nine hashed grid cells around the query (A0 + A1, batched)", observations=int(n), map_cones=int(len(mxy)),
                algorithmic_bytes=int(B), ms_per_launch=ms, achieved=B / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, bound="hbm",
                matched_fraction=float((idx >= 0).mean()), first_call_ms_with_grid_build=first_ms,
                note="everything resident in HBM (map, poses, observations, result); hashed grid built on the device once per map change; HIP start event attached to the first dispatch (cos / sin per pose), stop event to the second (the queries), mean of %d calls" % reps)


def frame_latency(pkg, np):
    """Per-keyframe latency of the real-time path (reference loop: src/slam.cpp:570-607, budgets of 20 ms gathering / 500 ms
    keyframe period in usecase/docker-compose.yml:16): A0 + A1 of ONE frame of K = 16 cones against a resident map of 200
    and of 10k cones (gs_frame_frontend: one launch, one wait), next to the CPU oracle's A0 + insertion-order scan of the
    same frame, and whole gs_slam_perform frames (graph insertion included) on the 1k-pose / 200-cone track."""
    from oracle import pyoracle as po
    fe = po.OracleFrontend()                                    # CPU baseline of the SAME frame only (cpu_frame below); the maps come from the product
    out = {"cones_per_frame": 16}
    for name, key in (("cfg2", "map_200"), ("cfg4", "map_10k")):
        N, M = pkg.track.CONFIGS[name]
        t = pkg.track.generate(N, M)
        feg = pkg.Graph(); g = pkg.track.bench_graph(t, feg); feg.close()      # the HIP front end (A0 on the device) builds the map
        mxy, mty = g["lm_est"], g["lm_type"].astype(np.int32)
        F = pkg.Graph(); F.map_append(mxy, mty)
        frames = [(t["odom_poses"][k], np.vstack([t["obs"][k], t["obs"][k + 1]])) for k in range(10, N - 2, max(1, N // 200))][:200]
        zero = np.zeros(16, dtype=np.int32)
        def cpu_frame(pose, obs):
            fe.polar_to_xy(obs[:, 0], obs[:, 1], obs[:, 2]); fe.cone_to_global(pose[None], zero, obs); fe.associate(pose[None], zero, obs, mxy, mty, 1.2)
        for fn, name2 in ((lambda pose, obs: F.frame_frontend(pose, obs, 1.2), "_gpu_us"), (cpu_frame, "_cpu_oracle_us")):
            for pose, obs in frames[:20]:                       # warm: first launches, pinned staging, clocks
                fn(pose, obs)
            ts = []
            for pose, obs in frames:
                t0 = time.perf_counter(); fn(pose, obs); ts.append(time.perf_counter() - t0)
            out[key + name2] = float(np.median(ts)) * 1e6       # median per frame
        F.close()
    N, M = pkg.track.CONFIGS["cfg2"]
    t = pkg.track.generate(N, M)
    S = pkg.Slam(same_cone_threshold=1.2, cone_mapping_threshold=67.0)
    tm_map, tm_loc = [], []
    for k in list(range(N)) + list(range(60)):
        closed = S.loop_closed
        t0 = time.perf_counter(); S.perform_slam(t["odom_poses"][k], t["obs"][k]); dt = time.perf_counter() - t0
        if closed: tm_loc.append(dt)
        elif not S.loop_closed: tm_map.append(dt)
        else: out["slam_loop_closing_frame_ms"] = dt * 1e3           # optimizeGraph (structure + 10 iterations) + updateMap + localizer
    out["slam_perform_mapping_frame_us"] = float(np.median(tm_map)) * 1e6 if tm_map else None
    out["slam_perform_localizer_frame_us"] = float(np.median(tm_loc)) * 1e6 if tm_loc else None
    out["slam_track"] = "1000 poses / 200 cones, 8 cones per frame, %d mapping frames, %d localizer frames" % (len(tm_map), len(tm_loc))
    S.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg4")
    ap.add_argument("--cpu-iters", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--shard", action="store_true",
                    help="N > 1: shard the NAMED workload itself over the ranks (strong scaling; `--workload cfg5 --shard --gpus 8` is "
                         "BASELINE config 5: 1M poses / 50k cones by pose window across 8 GPUs) instead of N x the workload")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip roofline_by_config (cfg3 / cfg5 beside the timed workload)")
    args = ap.parse_args()

    import torch                                   # first: its HIP runtime is the one the process uses
    import numpy as np
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU: the GraphSLAM back-end has no CPU fallback")
    # GS_BENCH_BACKEND=gloo: rehearsal of the multi-rank launch on a ONE-GPU box (all ranks share device 0 and the
    # exchange buffer is all-reduced through host copies); the driver's 8-GPU run uses nccl (= RCCL over xGMI)
    backend = os.environ.get("GS_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local = 0
    torch.cuda.set_device(local)
    dist = None
    # GS_BENCH_FORCE_DIST=1 (tests/test_gpu_parity.py): run the multi-GPU branch — RCCL process group, torch side stream
    # adopted by the library, gs_dist_iterate_local / all_reduce / gs_dist_iterate_finish — at world_size 1, so that this
    # code has executed on hardware before the driver's 8-GPU run
    dist_mode = world > 1 or os.environ.get("GS_BENCH_FORCE_DIST") == "1"
    if dist_mode:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL and gloo print banners ("RCCL version : ...", "[Gloo] Rank 0 is connected ...") on STDOUT when the first communicator
        # comes up: send them to stderr — stdout carries exactly one line, the JSON record
        sys.stdout.flush(); saved_stdout = os.dup(1); os.dup2(2, 1)
        try:
            if backend == "gloo":
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            w0 = torch.zeros(1, dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
            dist.all_reduce(w0); dist.barrier()                  # the first collective creates the communicator
            if backend != "gloo":
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush(); os.dup2(saved_stdout, 1); os.close(saved_stdout)
    pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")

    Nw, Mw = pkg.track.CONFIGS[args.workload]     # one pose window = the single-GPU workload
    strong = args.shard and world > 1
    N, M = (Nw, Mw) if strong else (Nw * world, Mw * world)
    track = pkg.track.generate(N, M)
    fe = pkg.Graph(device=local)
    g = pkg.track.bench_graph(track, fe)           # A0 on the device
    fe.close()
    # the multi-GPU branch at world size 1 (GS_BENCH_FORCE_DIST): the top three levels of the tree are made a SHARED top
    # (gs_debug_options.force_shared_top), so that the exchange buffer the group of one all-reduces is not empty
    dbg = dict(force_shared_top=3) if (dist_mode and world == 1) else None
    G = pkg.Graph(device=local, debug=dbg)
    # N > 1: rank-local ingestion — a rank is given every vertex and odometry edge, the observation edges of its own pose window, of the windows'
    # first poses and of the fixed poses, and the whole graph's landmark windows (gs_dist_set_landmark_windows); GS_BENCH_FULL_INGEST=1: every
    # rank holds the whole graph and finds the windows out itself (the default of the library, rounds 2-3)
    local_ingest = dist_mode and world > 1 and os.environ.get("GS_BENCH_FULL_INGEST", "0") != "1"
    ingest_frac = 1.0
    if local_ingest:
        ingest_frac = float(G.load_bench_graph_shard(g, rank, world).mean())
    else:
        G.load_bench_graph(g)
    def library_communicator(H):
        # RCCL INSIDE the library (north_star: the host side stays C++): rank 0 makes the unique id (ncclGetUniqueId through the C-ABI),
        # torch.distributed only carries its 128 bytes to the other ranks, every rank calls ncclCommInitRank through the C-ABI
        box = [pkg.binding.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        H.dist_comm_init(box[0], rank, world)
    if dist_mode and not local_ingest:
        G.dist_configure(rank, world)
    G.initialize_optimization()                    # structure phase (iteration-0 work): plan + upload to HBM
    plan = G.stats()
    lib_rccl, xbuf, stream = False, None, None
    if dist_mode and backend != "gloo":
        # every rank must take the same path: a rank whose library communicator failed tells the others (MIN over the ranks)
        try:
            library_communicator(G); ok_t = torch.ones(1, device="cuda")
        except Exception as e:
            sys.stderr.write("bench: RCCL inside the library unavailable on rank %d (%s): torch.distributed all-reduce instead\n" % (rank, e)); ok_t = torch.zeros(1, device="cuda")
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN); lib_rccl = bool(ok_t.item() > 0)
        if not lib_rccl:                           # rounds 2-3's path: a torch side stream adopted by the library, the exchange buffer a torch tensor
            stream = torch.cuda.Stream(); G.set_stream(stream.cuda_stream)
            xbuf = torch.zeros(max(G.dist_exchange_doubles(), 1), dtype=torch.float64, device="cuda")
            if G.dist_exchange_doubles() > 0:
                G.dist_set_exchange_buffer(xbuf.data_ptr())

    def step():
        if not dist_mode:
            G.iterate()
        elif backend == "gloo":
            G.dist_iterate_local()
            xh = torch.from_numpy(G.dist_read_exchange())
            dist.all_reduce(xh, op=dist.ReduceOp.SUM)
            G.dist_write_exchange(xh.numpy()); G.dist_iterate_finish()
        elif lib_rccl:
            G.dist_iterate()                       # local half -> ncclAllReduce(sum, fp64) of the shared rows of Omega / xi over xGMI -> finish: all enqueued from C++ on the handle's stream
        else:
            with torch.cuda.stream(stream):
                G.dist_iterate_local()
                dist.all_reduce(xbuf, op=dist.ReduceOp.SUM)
                G.dist_iterate_finish()

    def barrier():
        G.synchronize()                            # the library's own stream
        if dist_mode:
            dist.barrier()
        torch.cuda.synchronize()

    # A sharded run whose whole-tree launch gives up on a front's flag (the bounded polls: ranks that SHARE one GPU in a rehearsal are time-sliced against
    # each other, a profiler can do it too) reports that on every rank with the same all-reduce — the failure code rides in the exchange buffer —, the rank it
    # happened on switches to one launch per level, nobody applied the update: the measurement starts again (at most world + 2 times).  One GPU: as before, no retry.
    attempts = 0
    while True:
        try:
            for _ in range(args.warmup):
                step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            dt = time.perf_counter() - t0
            break
        except pkg.GsError as e:
            attempts += 1
            if not dist_mode or world == 1 or attempts > world + 2:      # (each report moves ONE rank to one launch per level for good: at most `world` of them can come)
                raise
            sys.stderr.write("bench: rank %d: %s -- measuring again (attempt %d)\n" % (rank, e, attempts + 1))
            dist.barrier()
    if dist_mode:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt * 1e3 / args.steps
    # weak: pose-window iterations per second over the whole job; strong (--shard): iterations per second of the ONE sharded graph
    value = (1 if strong else world) * args.steps / dt
    # the handle that was timed must have WORKED: a zero pivot or a solver launch that gave up on a front applies no
    # update and is reported here (GsError -> non-zero exit, no JSON line), on every rank
    G.sync_estimates()
    timed_poses, timed_lms = G.poses(), G.landmarks()
    if not (np.isfinite(timed_poses).all() and np.isfinite(timed_lms).all()):
        raise SystemExit("bench: the timed handle holds non-finite estimates")

    # ---- roofline of the edge-linearisation kernel (HIP events on the library's stream, this process)
    lin_ms = G.time_linearize(50)                  # this rank's window when sharded
    alg_bytes = G.linearize_bytes() // world       # E_pp*152 + E_pl*96 + N*120 + M*64  (SURVEY §8d), per window
    achieved = alg_bytes / (lin_ms * 1e-3) / 1e9
    # HBM traffic per launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE): rocprofv3 collects them in separate
    # passes around the process (scripts/pmc_iter.sh), so this run cannot measure them itself.  The committed measurement
    # of the same kernel on the same workload is reported — and only while the kernel source still hashes to what the
    # counters were taken on (scripts/make_pmc_json.py); otherwise null.
    traffic, traffic_src = None, None
    try:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from make_pmc_json import kernel_hash
        for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if fn.endswith("_linearize_pmc.json"):
                pm = json.load(open(os.path.join(ROOT, "profiles", fn)))
                w = pm.get("workloads", {}).get(args.workload)
                if w and pm.get("kernel_source_sha256") == kernel_hash() and world == 1:
                    traffic = w["traffic_bytes_per_launch"]
                    traffic_src = "committed_measurement profiles/%s (kernel source sha256 %s...)" % (fn, pm["kernel_source_sha256"][:12])
                break
    except Exception:
        traffic = None
    roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                    traffic=traffic, traffic_source=traffic_src, kernel="k_linearize_ell: edge linearisation + assembly (A5-A7)", ms_per_launch=lin_ms,
                    algorithmic_bytes=alg_bytes, note="back-to-back launches of this rank's window (N > 1: no per-phase events inside the sharded step)")
    out = dict(metric="GraphSLAM Gauss-Newton iters/sec at N poses x M cones; pose RMSE vs ref",
               value=value, unit="GN iterations/s (whole %d-pose graph)" % N if strong else "GN iterations/s (%dk-pose windows)" % (Nw // 1000),
               n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=ms_per_step, higher_is_better=True, scaling="strong" if strong else "weak", vs_baseline=None, dtype="f64",
               data="synthetic",
               config=dict(workload=("%s sharded over %d pose windows: %d poses / %d cones closed synthetic cone track, K=8 observations per pose, "
                                     "gauge = first 2 poses + first 2 cones" % (args.workload, world, N, M)) if strong else
                                    ("%s x %d: %d poses / %d cones closed synthetic cone track, K=8 observations per pose, "
                                     "gauge = first 2 poses + first 2 cones" % (args.workload, world, N, M)),
                           n_poses=N, n_cones=M, n_odometry_edges=G.n_pp, n_observation_edges=G.n_pl,
                           unknowns=3 * plan.n_free_poses + 2 * plan.n_free_landmarks,
                           parallelism=("%d pose windows, one per GPU; one RCCL all-reduce (sum, fp64) of %d doubles (%d bytes) per iteration, enqueued by the library (gs_dist_iterate)"
                                        % (world, G.dist_exchange_doubles(), 8 * G.dist_exchange_doubles())) if world > 1 else "single GPU",
                           fronts=plan.n_fronts, levels=plan.n_levels, max_front=plan.max_front),
               roofline=roofline)
    if dist_mode and backend != "gloo" and not lib_rccl:
        out["exchange"] = dict(doubles=int(G.dist_exchange_doubles()), bytes=int(8 * G.dist_exchange_doubles()), ms_exchange=None,
                               note="the library's own RCCL communicator could not be created: torch.distributed.all_reduce on a torch side stream (rounds 2-3's path)")
    if dist_mode and backend != "gloo" and lib_rccl:
        # the collective alone: `reps` all-reduces of the exchange buffer back to back on the library's stream (every rank calls it)
        out["exchange"] = dict(doubles=int(G.dist_exchange_doubles()), bytes=int(8 * G.dist_exchange_doubles()), ms_exchange=G.time_exchange(50),
                               shared_fronts=int(plan.n_shared_fronts), own_fronts=int(plan.n_own_fronts),
                               note="ms_exchange = one ncclAllReduce(sum, fp64) of the exchange buffer, mean of 50 back to back (HIP events on the library's stream); inside the iteration it sits between the local half and the shared top")
    if dist_mode:
        # the structure phase of the sharded graph (once per graph change, outside the timed region): a rank plans its own window and the shared
        # top from per-landmark window masks; the slowest rank's time
        ts = torch.tensor([plan.ms_structure], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        out["structure_ms_slowest_rank"] = float(ts.item())
        out["ingestion"] = ("rank-local: %.3f of the observation edges per rank (own window + the windows' first poses + fixed poses), landmark windows handed over" % ingest_frac
                            if local_ingest else "every rank holds the whole graph")
    if dist_mode and world == 1:
        out["config"]["parallelism"] = ("single GPU through the multi-GPU code path (GS_BENCH_FORCE_DIST): RCCL group of 1 created inside the library, top 3 levels forced shared "
                                        "(%d doubles all-reduced per iteration), gs_dist_iterate" % G.dist_exchange_doubles())
    if world == 1 and not dist_mode:
        warm_clocks(G)                             # (untimed; the timed region above is over)
        phases = G.time_iterations(20)
        out["phases_ms"] = dict(linearize=phases.ms_linearize, factor=phases.ms_factor, backsolve=phases.ms_backsolve,
                                update=phases.ms_update, structure_once=plan.ms_structure)
        # the roofline entry is the kernel AS IT RUNS INSIDE the Gauss-Newton iteration (same launch sequence as the timed region: its
        # inputs are cold there).  Back-to-back launches of the same kernel find cfg4's 105 MB still in the 256 MB Infinity Cache and
        # are reported beside it, not as `achieved`.
        # Duration = HIP start / stop events attached to the kernel's OWN dispatch (hipExtLaunchKernelGGL, on the stream it is launched
        # on) in iterations that carry no other event — the launch sequence of the timed region: begin -> end as the command processor
        # stamps the dispatch, the quantity a rocprofv3 kernel trace reports.  Measured on one box (profiles/r03_linearize_duration_*):
        # attached 29.7 us, the trace's in-iteration launches 28.4 us (a mean over these and the phase-timed iterations), events
        # recorded around the phase 30.8 us (ms_event_to_event: also holds the hand-over from k_update), and 27.3 us when an event
        # boundary precedes the launch (the previous kernel has then drained before the dispatch is stamped).
        lin_in = phases.ms_linearize_kernel if phases.ms_linearize_kernel > 0 else phases.ms_linearize
        out["phases_ms"]["event_overhead"] = phases.ms_event_overhead
        # ---- the solver (A8) against its own bytes: every L and update-matrix double is written once and read once, every
        # block of H read once.  Dependent latency bounds it, not bandwidth; the entry says by how much, and how close the
        # measured HBM traffic (committed PMC measurement, same rule as roofline.traffic) is to that minimum.
        h_bytes = 8 * (9 * N + 9 * G.n_pp + 6 * G.n_pl + 5 * M)
        sol_alg = 2 * plan.factor_bytes + h_bytes
        sol_ms = phases.ms_factor + phases.ms_backsolve
        sol_traffic, sol_src = None, None
        try:
            from make_pmc_json import solver_hash
            for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
                if fn.endswith("_solver_pmc.json"):
                    pm = json.load(open(os.path.join(ROOT, "profiles", fn)))
                    w = pm.get("workloads", {}).get(args.workload)
                    if w and pm.get("solver_source_sha256") == solver_hash():
                        sol_traffic = w["traffic_bytes_per_iteration"]
                        sol_src = "committed_measurement profiles/%s (solver source sha256 %s...)" % (fn, pm["solver_source_sha256"][:12])
                    break
        except Exception:
            sol_traffic = None
        out["solver"] = dict(kernels="k_factor3 (leaf + tree launch) + k_backsolve3 (tree launch + leaf levels), A8", ms_per_iteration=sol_ms,
                             algorithmic_bytes=sol_alg, achieved=sol_alg / (sol_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=sol_alg / (sol_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=sol_traffic, traffic_source=sol_src,
                             bound="dependent latency of the elimination tree (levels in sequence), instruction issue in the leaf level; not HBM",
                             note="algorithmic_bytes = 2 x (L + update matrices) + H blocks read once")
        inside = alg_bytes / (lin_in * 1e-3) / 1e9
        out["roofline"].update(achieved=inside, frac=inside / HBM_PEAK_GBS, ms_per_launch=lin_in, ms_event_to_event=phases.ms_linearize,
                               frac_event_to_event=alg_bytes / (phases.ms_linearize * 1e-3) / 1e9 / HBM_PEAK_GBS,      # rounds 1-2 reported this definition as `frac`: comparable across rounds
                               achieved_back_to_back=achieved, ms_per_launch_back_to_back=lin_ms,
                               note="HIP start/stop events attached to the kernel's dispatch inside full iterations (cold inputs), mean of 20 launches; ms_event_to_event = events recorded around the phase (holds the hand-over from the previous kernel too); "
                                    "achieved_back_to_back = the same kernel launched 50x in a row (inputs cached)")
    if rank == 0 and world == 1 and not args.no_cpu:
        og, cb = cpu_baseline(pkg, g, args.cpu_iters)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_baseline"] = value / cb["value"]
        # parity of what was timed: the same number of iterations from the same initial estimates
        G2 = pkg.Graph(device=local, debug=dbg); G2.load_bench_graph(g)
        if dist_mode and lib_rccl:                 # the same sharded arithmetic through gs_dist_optimize (Slam's optimize(10) on a sharded graph)
            G2.dist_configure(rank, world); G2.initialize_optimization(); library_communicator(G2)
            opt2 = G2.dist_optimize
        else:
            opt2 = G2.optimize
        done, st = opt2(args.cpu_iters)
        P, Lm = G2.poses(), G2.landmarks()
        rms = float(np.sqrt((og.poses()[:, :2] ** 2).sum(1).mean()))
        out["pose_rmse_vs_oracle_rel"] = float(np.sqrt(((P[:, :2] - og.poses()[:, :2]) ** 2).sum(1).mean()) / rms)
        out["landmark_rmse_vs_oracle_rel"] = float(np.sqrt(((Lm - og.landmarks()) ** 2).sum(1).mean()) / rms)
        out["heading_max_abs_diff_vs_oracle"] = float(np.abs(P[:, 2] - og.poses()[:, 2]).max())
        out["parity_iterations"] = int(done)
        # what one call of the reference's optimizeGraph costs END TO END after the graph has changed: structure phase
        # (initializeOptimization + analyzePattern there; plan + upload here) + 10 Gauss-Newton iterations + read-back
        G3 = pkg.Graph(device=local); G3.load_bench_graph(g)
        t0 = time.perf_counter(); G3.optimize(10); e2e_gpu = (time.perf_counter() - t0) * 1e3
        st3 = G3.stats(); G3.close()
        cb2 = cb if args.cpu_iters == 10 else cpu_baseline(pkg, g, 10)[1]       # the CPU leg above already ran optimize(10) end to end
        out["optimize10_end_to_end_ms"] = dict(gpu=e2e_gpu, gpu_structure_ms=st3.ms_structure, cpu=cb2["wall_ms_end_to_end"],
                                               cpu_symbolic_ms=cb2["ms_symbolic"], ratio=cb2["wall_ms_end_to_end"] / e2e_gpu,
                                               note="fresh handle, graph already inserted; wall clock of gs_optimize(10) "
                                                    "including plan build, upload and estimate read-back vs the CPU path's "
                                                    "buildStructure + analyzePattern + 10 iterations")
        # ... and the handle that was TIMED ran exactly that arithmetic: the same number of iterations through
        # gs_optimize on the second handle must reproduce its estimates bit for bit (every sum has a fixed order)
        more = args.warmup + args.steps - int(done)
        if more > 0:
            opt2(more)
        same = bool(np.array_equal(G2.poses(), timed_poses) and np.array_equal(G2.landmarks(), timed_lms))
        out["timed_handle_bitwise_equals_optimize"] = same
        out["timed_handle_iterations"] = args.warmup + args.steps
        if not same:
            raise SystemExit("bench: the timed handle's estimates differ from gs_optimize(%d) on a second handle" % (args.warmup + args.steps))
        G2.close()
    else:
        out["cpu_baseline"] = None
    if rank == 0 and world == 1 and not dist_mode and not args.no_extra_configs:
        try:
            out["association"] = association_roofline(pkg, np, track, g, local)
        except Exception as e:                                  # an extra: never costs the bench line
            out["association"] = dict(error=str(e))
    if rank == 0 and world == 1 and not dist_mode and not args.no_cpu and args.workload == "cfg4":
        out["frame_latency"] = frame_latency(pkg, np)
    if rank == 0 and world == 1 and not dist_mode and not args.no_extra_configs and args.workload == "cfg4":
        # the roofline kernel where it is launch-bound (cfg3), at the headline size (cfg4 = the entry above) and where it is
        # HBM-bound (cfg5, 1.05 GB per pass): same kernel, same measurement (HIP events inside full iterations)
        rb = {"cfg4": dict(algorithmic_bytes=alg_bytes, ms_per_launch=out["roofline"]["ms_per_launch"], achieved=out["roofline"]["achieved"],
                           frac=out["roofline"]["frac"], ms_per_launch_back_to_back=lin_ms, frac_back_to_back=achieved / HBM_PEAK_GBS,
                           iteration_ms=ms_per_step, iterations_per_s=value)}
        # one more keyframe on the timed handle (reference src/slam.cpp:433-459, 537-550: a pose vertex, its odometry edge, observation
        # edges to the cones the last pose sees): the structure phase the next optimize() needs — the plan grows, nothing is rebuilt
        try:
            p_last = G.poses()[-1]; step = np.array([0.25, 0.0, 0.0]); c, s_ = np.cos(p_last[2]), np.sin(p_last[2])
            p_new = np.array([p_last[0] + c * step[0], p_last[1] + s_ * step[0], p_last[2]])
            seen = g["pl_l"][g["pl_p"] == N - 1]; L_xy = G.landmarks()[seen]
            z = np.stack([c * (L_xy[:, 0] - p_new[0]) + s_ * (L_xy[:, 1] - p_new[1]), -s_ * (L_xy[:, 0] - p_new[0]) + c * (L_xy[:, 1] - p_new[1])], axis=1)
            G.add_pose(N, p_new); G.add_odometry_edge(N - 1, N, step, np.asarray(g["pp_info"][0]).reshape(3, 3))
            G.add_observation_edges(np.full(len(seen), N), seen, z, np.tile(np.asarray(g["pl_info"][0]).reshape(1, 4), (len(seen), 1)))
            t0 = time.perf_counter(); G.initialize_optimization(); grow_ms = 1e3 * (time.perf_counter() - t0)
            grown = G.plan_growths(); it_g = G.time_iterations(10)
            out["growth"] = dict(what="one more pose + its odometry edge + %d observation edges on the timed handle, then gs_initialize_optimization" % len(seen),
                                 ms_structure_after_one_more_keyframe=grow_ms, plan_growths=int(grown), refusal=G.growth_refusal(),
                                 ms_structure_full=plan.ms_structure, iteration_ms_after=it_g.ms_total, iteration_ms_before=phases.ms_total)
        except Exception as e:                                  # an extra: never costs the bench line
            out["growth"] = dict(error=str(e))
        G.close(); G = None                                     # cfg5 wants the HBM to itself
        for name in ("cfg3", "cfg5"):
            rb[name] = linearize_roofline_of(pkg, name, local)
        out["roofline_by_config"] = rb
        # tracks with 16 / 24 cones in view (what the reference's coneMappingThreshold of 50 m lets a frame hold, src/slam.cpp:608)
        # at the headline size: separators of 35 / 51 scalars, fronts up to 105 / 153 — a workgroup per front, chosen per front
        wv = {}
        for K in (16, 24):
            Nk, Mk = pkg.track.CONFIGS["cfg4"]
            tk = pkg.track.generate(Nk, Mk, K)
            fe = pkg.Graph(device=local); gk = pkg.track.bench_graph(tk, fe); fe.close()
            Gk = pkg.Graph(device=local); Gk.load_bench_graph(gk); Gk.initialize_optimization(); stk = Gk.stats()
            ph = Gk.time_iterations(10)
            wv["K%d" % K] = dict(observations_per_pose=K, n_observation_edges=int(len(gk["pl_p"])), fronts=stk.n_fronts, big_fronts=stk.n_big_fronts, levels=stk.n_levels,
                                 max_front=stk.max_front, factor_variant=stk.factor_variant, iteration_ms=ph.ms_total, iterations_per_s=1e3 / ph.ms_total,
                                 phases_ms=dict(linearize=ph.ms_linearize, factor=ph.ms_factor, backsolve=ph.ms_backsolve, update=ph.ms_update),
                                 per_edge_rate_vs_K8=(1e3 / ph.ms_total * len(gk["pl_p"])) / (value * out["config"]["n_observation_edges"]), structure_ms=stk.ms_structure)
            Gk.close()
        out["wide_view_tracks"] = wv
    if rank == 0:
        print(json.dumps(out))
    if G is not None:
        G.close()
    if dist_mode:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
