#!/usr/bin/env python3
"""End-to-end `fit` throughput (host loop included): python tools/fit_throughput.py [rows] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import WORKLOADS, build_model, synthetic_batches

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = WORKLOADS["criteo_c2"]
model = build_model(cfg, 100000, torch.device("cuda:0"))
(X, y), = synthetic_batches(1, rows, [100000] * cfg["n_sparse"], cfg["n_dense"], seed=1)
names = list(model.feature_index.keys())
xs = {n: X[:, i] for i, n in enumerate(names)}
model.fit(xs, y, batch_size=bs, epochs=1, verbose=0)            # warm-up epoch (allocator, first launches)
torch.cuda.synchronize()
t0 = time.perf_counter()
h = model.fit(xs, y, batch_size=bs, epochs=2, verbose=0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("fit: %d rows x 2 epochs, batch %d: %.3f s -> %.0f examples/s (loss %.5f -> %.5f)"
      % (rows, bs, dt, 2 * rows / dt, h.history["loss"][0], h.history["loss"][-1]))
t0 = time.perf_counter()
p = model.predict(xs, batch_size=8192)
dt = time.perf_counter() - t0
print("predict: %d rows in %.3f s -> %.0f examples/s" % (rows, dt, rows / dt))
