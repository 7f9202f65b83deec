"""Probe: cost of the split step (graph replay of the collective-free half + eager second half) on ONE GPU, with a
stand-in for the row-parallel context whose collectives are no-ops -- what the N > 1 step pays besides RCCL."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from xdfm_amd import dist as xdist  # noqa: E402


class OneRank(object):
    world, rank, backend = 1, 0, "none"
    _n_global = None
    _replicated = set()

    def mark_replicated(self, params):
        pass

    def reduce_dense_grads(self, model):
        pass

    def exchange_rows(self, X, d_emb, d_dnn, d_lin):
        return [(X, d_emb, d_dnn, d_lin)]


cfg = bench.WORKLOADS["criteo_c2"]
dev = torch.device("cuda:0")
batches = [(torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)) for X, y in
           bench.synthetic_batches(8, cfg["batch"], [100000] * cfg["n_sparse"], cfg["n_dense"], seed=1)]
for mode in ("single", "split"):
    ctx = OneRank()
    ctx._n_global = cfg["batch"]
    xdist.current = (lambda: ctx) if mode == "split" else (lambda: None)
    model = bench.build_model(cfg, 100000, dev)
    model.train()
    for s in range(6):
        model.train_on_batch(*batches[s % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(40):
        model.train_on_batch(*batches[s % 8])
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    step = model.__dict__["_graphed_step"]
    print("%s: %.3f ms/step (host %.3f), replays %d" % (mode, (time.perf_counter() - t0) / 40 * 1e3, th / 40 * 1e3, step.replays), flush=True)
