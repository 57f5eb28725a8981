"""Random sharded plans replayed on the CPU (no GPU needed; as tests/test_shards_cpu.py does for fixed seeds): laps of 40-500 poses with every observation of a
few poses dropped and poses fixed in the middle of the chain (windows that fall apart), worlds 2 .. 64 (windows down to 4 poses), planned by windows and by the
general recursion; every rank's plan replayed in numpy (tests/plan_exec.py), the exchange buffers summed, the merged increment against the oracle's joint solve.
Checks: every edge has exactly one owner, every rank's exchange buffer has the same length, the merged increment is the oracle's to 1e-7.
usage: python scripts/stress_shards_cpu.py [first_seed] [count]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from plan_exec import Plan
from conftest import make_oracle_graph
pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
from oracle import pyoracle as po
fe=po.OracleFrontend(); bad=0; total=0; eng=0; nloc=0
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0; count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for seed in range(first, first + count):
    rng=np.random.default_rng(6000+seed)
    N=int(rng.integers(40,500)); M=max(30,N//int(rng.integers(3,7)))
    try: t=pkg.track.generate(N,M)
    except ValueError: continue
    g0=pkg.track.bench_graph(t,fe); g=dict(g0)
    # drop every observation of a few random poses; fix a few poses
    drop=set(rng.choice(N, int(rng.integers(0,6)), replace=False).tolist())
    keep=~np.isin(g["pl_p"], list(drop))
    for k in ("pl_p","pl_l","pl_z","pl_info"): g[k]=g[k][keep]
    # landmarks that lost all observers: fix them so the system stays regular
    seen=np.zeros(len(g["lm_est"]),bool); seen[g["pl_l"]]=True
    g["fixed_landmarks"]=np.array(sorted(set(g["fixed_landmarks"].tolist())|set(np.flatnonzero(~seen).tolist())),dtype=np.int32)
    g["fixed_poses"]=np.array(sorted(set([0,1]+rng.choice(N,int(rng.integers(0,4)),replace=False).tolist())),dtype=np.int32)
    og=make_oracle_graph(po,g); og.build_system()
    try: x=og.solve_ldlt(1)
    except Exception: continue
    og.apply_update(x); dp_o,dl_o=og.delta(); scale=max(np.abs(dp_o).max(),np.abs(dl_o).max())
    nfp=N-len(g["fixed_poses"])
    worlds=sorted(set([2,3,int(rng.integers(4,17)), min(64,max(2,nfp//4)), min(64,max(2,nfp//5))]))
    for world in worlds:
        for bw in (1,0):
            total+=1
            plans=[];locs=[];prim=[]; npl=len(g["pl_p"]); npp=len(g["pp_i"]); spl=np.zeros(npl,int); spp=np.zeros(npp,int); e=False
            try:
                for rank in range(world):
                    G=pkg.Graph(device=-2, debug=dict(shard_by_window=bw)); G.load_bench_graph(g); G.dist_configure(rank,world); G.plan_build_host()
                    P=Plan(G.plan_export()); P.check_invariants()
                    kp=P.pp_rank==rank; kl=P.pl_rank==rank; spl+=kl; spp+=kp; e = e or bool((P.pl_rank==-1).any())
                    sub=dict(g)
                    for k in ("pp_i","pp_j","pp_z","pp_info"): sub[k]=g[k][kp]
                    for k in ("pl_p","pl_l","pl_z","pl_info"): sub[k]=g[k][kl]
                    bs=make_oracle_graph(po,sub).linearize_blocks(); blocks=dict(bs)
                    blocks["Hpp_off"]=np.zeros((npp,9)); blocks["Hpp_off"][kp]=bs["Hpp_off"]
                    blocks["Hpl"]=np.zeros((npl,6)); blocks["Hpl"][kl]=bs["Hpl"]
                    X,ok=P.shard_local(blocks); assert ok
                    plans.append(P); locs.append(X); prim.append(G.dist_known()); G.close()
                eng+=e
                if bw == 1 and e:                                # rank-local ingestion (gs_dist_set_landmark_windows) must give the same plans, field by field
                    masks = pkg.binding.landmark_windows(g, world)
                    for rank in range(world):
                        L = pkg.Graph(device=-2); keep = L.load_bench_graph_shard(g, rank, world, masks); L.plan_build_host(); PL = Plan(L.plan_export()); L.close()
                        PF = plans[rank]
                        for name in ("npiv", "nbnd", "parent", "level", "owner", "piv0", "bnd_rows", "pose_gidx", "lm_gidx", "x_off", "pp_rank"):
                            assert np.array_equal(getattr(PL, name), getattr(PF, name)), ("rank-local plan differs", rank, name)
                        idx = np.flatnonzero(keep)
                        assert np.array_equal(PL.pl_rank == rank, (PF.pl_rank == rank)[idx]) and not (PF.pl_rank == rank)[~keep].any(), ("rank-local edges", rank)
                    nloc += 1
                assert (spl==1).all() and (spp==1).all(), "edge owners"
                assert len({len(x) for x in locs})==1, ("exchange sizes",[len(x) for x in locs])
                Xs=np.sum(locs,axis=0); dp=np.zeros_like(dp_o); dl=np.zeros_like(dl_o)
                for P,(pk,lk,pp_,lp_) in zip(plans,prim):
                    a,b,okk=P.shard_finish(Xs.copy()); assert okk
                    dp+=a*pp_[:,None]; dl+=b*lp_[:,None]
                err=max(np.abs(dp-dp_o).max(), np.abs(dl-dl_o).max())/scale
                assert err<1e-7, ("err",err)
            except Exception as ex:
                bad+=1; print("BAD seed",seed,"N",N,"world",world,"bw",bw,repr(ex)[:200], flush=True)
print("seeds %d..%d: cases %d, by-window engaged %d (rank-local ingestion compared in %d), bad %d" % (first, first + count - 1, total, eng, nloc, bad))
sys.exit(1 if bad else 0)
