"""Dev check 2: do back-to-back graph replays (no host sync) stay correct?  Trap-free loss so a NaN does not abort."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import bench  # noqa: E402

cfg = bench.WORKLOADS["criteo_c2"]
dev = torch.device("cuda:0")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 0        # 0 = unlimited replays in flight


def safe_bce(p, y, reduction="sum"):
    return F.binary_cross_entropy(torch.nan_to_num(p, nan=0.5).clamp(0.0, 1.0), y, reduction=reduction)


def make():
    torch.manual_seed(0)
    m = bench.build_model(cfg, 100000, dev)
    m.optim = torch.optim.Adam(m.parameters(), fused=True, capturable=True)
    m.loss_func = safe_bce
    m.train()
    return m


model, ref = make(), make()
ref.load_state_dict(model.state_dict())
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
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = model.train_on_batch(sx, sy)
torch.cuda.synchronize()
losses = torch.zeros(64, device=dev)
events = []
t0 = time.perf_counter()
for s in range(3, 43):
    sx.copy_(batches[s % 8][0]); sy.copy_(batches[s % 8][1])
    g.replay()
    losses[s].copy_(out[2].detach().reshape(()))
    if depth:
        ev = torch.cuda.Event()
        ev.record()
        events.append(ev)
        if len(events) > depth:
            events.pop(0).synchronize()
th = time.perf_counter() - t0
torch.cuda.synchronize()
print("graph: %.3f ms/step (host %.3f), depth limit %d" % ((time.perf_counter() - t0) / 40 * 1e3, th / 40 * 1e3, depth), flush=True)
rl = []
for s in range(3, 43):
    rl.append(ref.train_on_batch(*batches[s % 8])[2].detach().reshape(()))
torch.cuda.synchronize()
gl = losses[3:43].cpu().tolist()
rl = [float(v) for v in rl]
bad = [i + 3 for i, (a, b) in enumerate(zip(gl, rl)) if not (abs(a - b) <= 2e-4 * abs(b))]
print("first losses graph", ["%.3f" % v for v in gl[:4]], "eager", ["%.3f" % v for v in rl[:4]])
print("last  losses graph", ["%.3f" % v for v in gl[-3:]], "eager", ["%.3f" % v for v in rl[-3:]])
print("steps whose loss differs from eager by > 2e-4 relative:", bad, flush=True)
