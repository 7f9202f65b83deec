"""Where does `fit` lose time against the bare train step?  Times the replayed step with the pieces of the fit loop
added one at a time (GPU box): python tools/fit_loop_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
sys.path.insert(0, ROOT)
import torch
from bench import WORKLOADS, build_model, synthetic_batches

cfg = WORKLOADS["criteo_c2"]
dev = torch.device("cuda:0")
model = build_model(cfg, 100000, dev)
model.train()
rows, bs = 262144, 4096
(X, y), = synthetic_batches(1, rows, [100000] * cfg["n_sparse"], cfg["n_dense"], seed=1)
X_all = torch.as_tensor(X, dtype=torch.float32).to(dev)
Y_all = torch.as_tensor(y, dtype=torch.float32).reshape(-1, 1).to(dev)
order = torch.randperm(rows).to(dev)
x0, y0 = X_all[:bs].clone(), Y_all[:bs].clone()
for _ in range(6):
    model.train_on_batch(x0, y0)
log = torch.empty((rows // bs + 1, 2), device=dev)


def run(name, body, n=rows // bs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        body(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-46s %.3f ms/step (host enqueue %.3f)" % (name, (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3), flush=True)


def fixed(i):
    model.train_on_batch(x0, y0)


def sliced(i):
    model.train_on_batch(X_all[i * bs:(i + 1) * bs], Y_all[i * bs:(i + 1) * bs])


def shuffled(i):
    idx = order[i * bs:(i + 1) * bs]
    model.train_on_batch(X_all.index_select(0, idx), Y_all.index_select(0, idx))


def shuffled_logged(i):
    idx = order[i * bs:(i + 1) * bs]
    _, loss, total = model.train_on_batch(X_all.index_select(0, idx), Y_all.index_select(0, idx))
    log[i, 0:1].copy_(loss.detach().reshape(1))
    log[i, 1:2].copy_(total.detach().reshape(1))


def shuffled_item(i):
    idx = order[i * bs:(i + 1) * bs]
    _, loss, total = model.train_on_batch(X_all.index_select(0, idx), Y_all.index_select(0, idx))
    total.item()


for rep in range(2):
    run("fixed batch", fixed)
    run("contiguous slices", sliced)
    run("index_select batches", shuffled)
    run("index_select + loss log", shuffled_logged)
    run("index_select + .item() per step", shuffled_item)
