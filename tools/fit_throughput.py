#!/usr/bin/env python3
"""End-to-end `fit` throughput (host loop included): python tools/fit_throughput.py [rows] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
sys.path.insert(0, ROOT)
import torch
from bench import WORKLOADS, build_model, synthetic_batches

nums = [a for a in sys.argv[1:] if a.isdigit()]
rows = int(nums[0]) if len(nums) > 0 else 262144
bs = int(nums[1]) if len(nums) > 1 else 4096
cfg = WORKLOADS["criteo_c2"]
card = "card" in sys.argv
vocab = list(__import__("bench").CRITEO_CARD) if card else [100000] * cfg["n_sparse"]
model = build_model(cfg, vocab, torch.device("cuda:0"))
(X, y), = synthetic_batches(1, rows, vocab, cfg["n_dense"], seed=1)
names = list(model.feature_index.keys())
xs = {n: X[:, i] for i, n in enumerate(names)}
model.fit(xs, y, batch_size=bs, epochs=1, verbose=0)            # warm-up epoch (allocator, first launches)


def timed_fit(epochs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = model.fit(xs, y, batch_size=bs, epochs=epochs, verbose=0)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, h


t1, _ = timed_fit(1)
t3, h = timed_fit(3)
per_epoch = (t3 - t1) / 2          # what an epoch costs once the inputs are on the device (the one-off conversion of
print("fit: %d rows, batch %d: 3 epochs %.3f s = %.0f examples/s end to end; steady state %.3f s per epoch = %.0f examples/s "
      "(%.3f ms/step); one-off input conversion + first epoch %.3f s (loss %.5f -> %.5f)"
      % (rows, bs, t3, 3 * rows / t3, per_epoch, rows / per_epoch, per_epoch / ((rows + bs - 1) // bs) * 1e3, t1,
         h.history["loss"][0], h.history["loss"][-1]))
t0 = time.perf_counter()
p = model.predict(xs, batch_size=8192)
dt = time.perf_counter() - t0
print("predict: %d rows in %.3f s -> %.0f examples/s" % (rows, dt, rows / dt))
ts = []
for _ in range(3):                                   # again: the host-side conversion of the numpy inputs dominates and varies
    t0 = time.perf_counter()
    p = model.predict(xs, batch_size=8192)
    ts.append(time.perf_counter() - t0)
print("predict, 3 more calls: %s s -> best %.0f examples/s" % (", ".join("%.3f" % t for t in ts), rows / min(ts)))
