"""Dev check: capture the whole train step in a HIP graph (torch.cuda.CUDAGraph) and compare the
parameter trajectory and the step time with the eager loop.  GPU box only."""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

cfg = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "criteo_c2"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench.build_model(cfg, 100000, dev)
model.optim = torch.optim.Adam(model.parameters(), fused=True, capturable=True)
ref = bench.build_model(cfg, 100000, dev)
ref.load_state_dict(model.state_dict())
model.train(); ref.train()
batches = [(torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)) for X, y in
           bench.synthetic_batches(8, cfg["batch"], [100000] * cfg["n_sparse"], cfg["n_dense"], seed=1)]

sx, sy = torch.empty_like(batches[0][0]), torch.empty_like(batches[0][1])
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for s in range(3):
        sx.copy_(batches[s][0]); sy.copy_(batches[s][1])
        model.train_on_batch(sx, sy)
torch.cuda.current_stream().wait_stream(side)
for s in range(3):
    ref.train_on_batch(*batches[s])
torch.cuda.synchronize()
print("warm-up done, capturing", flush=True)
g = torch.cuda.CUDAGraph()
sx.copy_(batches[3][0]); sy.copy_(batches[3][1])
with torch.cuda.graph(g):
    out = model.train_on_batch(sx, sy)
torch.cuda.synchronize()
print("captured", flush=True)
# the capture itself does not execute: replay step 3.. and compare with eager
for s in range(3, 8):
    sx.copy_(batches[s][0]); sy.copy_(batches[s][1])
    g.replay()
    r = ref.train_on_batch(*batches[s])
    torch.cuda.synchronize()
    print("step %d: loss graph %.6f eager %.6f" % (s, float(out[2]), float(r[2])), flush=True)
worst = 0.0
for (k, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
    worst = max(worst, float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)))
print("max relative parameter difference after 8 steps: %.3e" % worst, flush=True)
for name, fn in (("graph", lambda s: (sx.copy_(batches[s % 8][0]), sy.copy_(batches[s % 8][1]), g.replay())),
                 ("eager", lambda s: ref.train_on_batch(*batches[s % 8]))):
    for s in range(3):
        fn(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(30):
        fn(s)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%s: %.3f ms/step (host %.3f)" % (name, (time.perf_counter() - t0) / 30 * 1e3, th / 30 * 1e3), flush=True)
