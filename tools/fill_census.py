#!/usr/bin/env python3
"""Dev: which host-side ops of one eager train step launch the small ATen kernels (fills, copies, adds) that survive in
the captured step?  Prints, per such device kernel, the chain of CPU ops above it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in bench.WORKLOADS else "criteo_c2"
cfg = bench.WORKLOADS[workload]
dev = torch.device("cuda:0")
vocab = bench.preset_vocab("criteo-card", cfg["n_sparse"]) if "card" in sys.argv else [100000] * cfg["n_sparse"]
os.environ["XDFM_HIP_GRAPH"] = "0"
model = bench.build_model(cfg, vocab, dev)
model.train()
X, y = bench.synthetic_batches(1, cfg["batch"], vocab, cfg["n_dense"], seed=1)[0]
X, y = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
for s in range(4):
    model.train_on_batch(X, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    model.train_on_batch(X, y)
    torch.cuda.synchronize()
n = 0
for e in prof.events():
    for k in getattr(e, "kernels", []) or []:
        if any(t in k.name for t in ("FillFunctor", "copyBuffer", "CUDAFunctor_add", "multi_tensor", "elementwise_kernel", "fillBuffer")):
            chain, p = [], e
            while p is not None and len(chain) < 6:
                chain.append(p.name[:48])
                p = p.cpu_parent
            print("%-46s %6.1f us  <- %s" % (k.name[:46], k.duration, " <- ".join(chain)))
            n += 1
print(n, "small ATen kernels in one eager step")
