"""cProfile of the host side of the train step (where does the enqueue time go?).  GPU box only."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

cfg = bench.WORKLOADS["criteo_c2"]
dev = torch.device("cuda:0")
model = bench.build_model(cfg, 100000, dev)
model.train()
batches = [(torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)) for X, y in
           bench.synthetic_batches(4, cfg["batch"], [100000] * 26, 13, seed=1)]
for s in range(5):
    model.train_on_batch(*batches[s % 4])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for s in range(20):
    model.train_on_batch(*batches[s % 4])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
