"""Dev check 3: long replay sequences, optionally interleaved with eager work, against an eager twin."""
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
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"        # plain | interleave | eager
NSTEP = int(sys.argv[2]) if len(sys.argv) > 2 else 400


def safe_bce(p, y, reduction="sum"):
    return F.binary_cross_entropy(torch.nan_to_num(p, nan=0.5).clamp(0.0, 1.0), y, reduction=reduction)


def make():
    torch.manual_seed(0)
    m = bench.build_model(cfg, 100000, dev)
    m.optim = torch.optim.Adam(m.parameters(), fused=True, capturable=True)
    for pg in m.optim.param_groups:
        pg["lr"] = 1e-4
    m.loss_func = safe_bce
    m.train()
    return m


model, twin, other = make(), make(), make()
twin.load_state_dict(model.state_dict())
batches = [(torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)) for X, y in
           bench.synthetic_batches(32, cfg["batch"], [100000] * cfg["n_sparse"], cfg["n_dense"], seed=1)]
sx, sy = torch.empty_like(batches[0][0]), torch.empty_like(batches[0][1])
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for s in range(3):
        sx.copy_(batches[s][0]); sy.copy_(batches[s][1])
        model.train_on_batch(sx, sy)
torch.cuda.current_stream().wait_stream(side)
for s in range(3):
    twin.train_on_batch(*batches[s])
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = model.train_on_batch(sx, sy)
torch.cuda.synchronize()
losses = torch.zeros(NSTEP + 3, device=dev)
nanflag = torch.zeros((), device=dev)
t0 = time.perf_counter()
for s in range(3, NSTEP + 3):
    sx.copy_(batches[s % 32][0]); sy.copy_(batches[s % 32][1])
    if mode == "eager":
        out = model.train_on_batch(sx, sy)                       # natural run-to-run divergence (atomics order)
    else:
        g.replay()
    losses[s].copy_(out[2].detach().reshape(()))
    nanflag += (~torch.isfinite(out[0].detach())).any().float()
    if mode == "interleave" and s % 7 == 0:
        other.train_on_batch(*batches[(s * 3) % 32])             # another model, eager
        with torch.no_grad():
            model.eval(); model(batches[(s * 5) % 32][0]); model.train()   # the same model, eager forward
        if s % 21 == 0:
            torch.cuda.synchronize()
torch.cuda.synchronize()
print("%s: %d replays, %.3f ms/step, non-finite predictions in %d steps" % (
    mode, NSTEP, (time.perf_counter() - t0) / NSTEP * 1e3, int(nanflag.item())), flush=True)
tl = []
for s in range(3, NSTEP + 3):
    tl.append(twin.train_on_batch(*batches[s % 32])[2].detach().reshape(()))
torch.cuda.synchronize()
gl = losses[3:].cpu()
tl = torch.stack(tl).cpu()
rel = ((gl - tl).abs() / tl.abs())
print("loss first/last graph %.4f %.4f twin %.4f %.4f; max rel diff %.2e at step %d; steps over 1e-3: %d" % (
    gl[0], gl[-1], tl[0], tl[-1], rel.max(), int(rel.argmax()) + 3, int((rel > 1e-3).sum())), flush=True)
