"""Run by tests/test_gpu_parity.py::test_device_exchange_buffers_on_a_shared_torch_stream in its own process (GPU needed).

Two pose-window rank handles on ONE GPU share one torch side stream; their exchange buffers are torch tensors handed to
the library by pointer (gs_dist_set_exchange_buffer), and the all-reduce between gs_dist_iterate_local and
gs_dist_iterate_finish is a tensor add enqueued on that stream — the plumbing bench.py uses with RCCL on 8 GPUs."""
import importlib
import os
import sys

import torch                                   # first: its HIP runtime is the one the process uses (as in bench.py)
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_oracle_graph          # noqa: E402
from oracle import pyoracle as po               # noqa: E402

pkg = importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
assert torch.cuda.is_available()
t = pkg.track.generate(10000, 2000)
g = pkg.track.bench_graph(t, po.OracleFrontend())
stream = torch.cuda.Stream()
ranks, bufs = [], []
for r in range(2):
    H = pkg.Graph(); H.load_bench_graph(g); H.dist_configure(r, 2); H.set_stream(stream.cuda_stream); H.initialize_optimization()
    x = torch.zeros(H.dist_exchange_doubles(), dtype=torch.float64, device="cuda")
    H.dist_set_exchange_buffer(x.data_ptr()); ranks.append(H); bufs.append(x)
with torch.cuda.stream(stream):
    for _ in range(5):
        for H in ranks:
            H.dist_iterate_local()
        bufs[0].add_(bufs[1]); bufs[1].copy_(bufs[0])       # stands where dist.all_reduce(xbuf) stands in bench.py
        for H in ranks:
            H.dist_iterate_finish()
stream.synchronize()
N, Mg = len(g["pose_est"]), len(g["lm_est"])
P = np.zeros((N, 3)); L = np.zeros((Mg, 2))
for H in ranks:
    H.sync_estimates()
    pk, lk, pprim, lprim = H.dist_known()
    P += H.poses() * pprim[:, None]; L += H.landmarks() * lprim[:, None]
og = make_oracle_graph(po, g); og.optimize(5, ordering=1)
ep = np.abs(P - og.poses()).max() / np.abs(og.poses()).max(); el = np.abs(L - og.landmarks()).max() / np.abs(og.landmarks()).max()
assert ep < 1e-9 and el < 1e-9, (ep, el)
for H in ranks:
    H.set_stream(0); H.close()
print("ok: device exchange buffers on a shared torch stream, rel diff vs oracle %.3g %.3g" % (ep, el))
