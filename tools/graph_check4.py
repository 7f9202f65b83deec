"""Dev check 4: which tensor goes non-finite first under graph replay?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import bench  # noqa: E402
from xdfm_amd import _lib  # noqa: E402

cfg = bench.WORKLOADS["criteo_c2"]
dev = torch.device("cuda:0")
NSTEP = int(sys.argv[1]) if len(sys.argv) > 1 else 600
if len(sys.argv) > 2:
    _lib.set_option("cin_math", int(sys.argv[2]))


def safe_bce(p, y, reduction="sum"):
    return F.binary_cross_entropy(torch.nan_to_num(p, nan=0.5).clamp(0.0, 1.0), y, reduction=reduction)


torch.manual_seed(0)
model = bench.build_model(cfg, 100000, dev)
model.optim = torch.optim.Adam(model.parameters(), fused=True, capturable=True)
for pg in model.optim.param_groups:
    pg["lr"] = 1e-4
model.loss_func = safe_bce
model.train()
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
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = model.train_on_batch(sx, sy)
torch.cuda.synchronize()
named = [(n, p) for n, p in model.named_parameters()]
watch = [("y_pred", lambda: out[0].detach())]
for n, p in named:
    if "embedding_dict" in n and not n.endswith("C1.weight"):
        continue
    watch.append((n, (lambda p=p: p.detach())))
    watch.append((n + ".grad", (lambda p=p: p.grad)))
first = torch.full((len(watch),), 10 ** 9, dtype=torch.int64, device=dev)
for s in range(3, NSTEP + 3):
    sx.copy_(batches[s % 32][0]); sy.copy_(batches[s % 32][1])
    g.replay()
    bad = torch.stack([(~torch.isfinite(f())).any() for _, f in watch])
    first = torch.where(bad & (first == 10 ** 9), torch.full_like(first, s), first)
torch.cuda.synchronize()
first = first.cpu().tolist()
order = sorted(range(len(watch)), key=lambda i: first[i])
for i in order[:14]:
    print("%-60s first non-finite at step %s" % (watch[i][0], first[i] if first[i] < 10 ** 9 else "never"), flush=True)
