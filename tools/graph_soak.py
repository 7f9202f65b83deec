"""Soak: the product train step (HIP-graph replay, K7 with the L2 term, K8 head, f16x3 CIN) for many steps against an
eager twin fed the same batches.  Differences must stay at the run-to-run noise of the scatter atomics."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from xdfm_amd import graphstep  # noqa: E402

NSTEP = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
cfg = bench.WORKLOADS["criteo_c2"]
dev = torch.device("cuda:0")


def make(use_graph):
    torch.manual_seed(0)
    m = bench.build_model(cfg, 100000, dev)
    for pg in m.optim.param_groups:
        pg["lr"] = 1e-4
    m.train()
    step = graphstep.GraphedStep(m)
    step.disabled = not use_graph
    m.__dict__["_graphed_step"] = step
    return m, step


batches = [(torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)) for X, y in
           bench.synthetic_batches(32, cfg["batch"], [100000] * cfg["n_sparse"], cfg["n_dense"], seed=1)]
res = {}
for name, use_graph in (("graph", True), ("eager", False)):
    m, step = make(use_graph)
    losses = torch.zeros(NSTEP, device=dev)
    bad = torch.zeros((), device=dev)
    t0 = time.perf_counter()
    for s in range(NSTEP):
        out = m.train_on_batch(*batches[s % 32])
        losses[s].copy_(out[2].reshape(()))
        bad += (~torch.isfinite(out[0])).any().float()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[name] = (losses.cpu(), {k: v.detach().clone() for k, v in m.state_dict().items()})
    print("%s: %d steps, %.3f ms/step, replays %d, steps with non-finite predictions %d" % (
        name, NSTEP, dt / NSTEP * 1e3, step.replays, int(bad.item())), flush=True)
lg, le = res["graph"][0], res["eager"][0]
rel = ((lg - le).abs() / le.abs())
print("loss first/last: graph %.4f %.4f eager %.4f %.4f; max relative difference %.2e at step %d" % (
    lg[0], lg[-1], le[0], le[-1], rel.max(), int(rel.argmax())), flush=True)
worst = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)) for a, b in zip(res["graph"][1].values(), res["eager"][1].values()))
print("max relative parameter difference after %d steps: %.2e" % (NSTEP, worst), flush=True)
