"""Dev: dump the captured train-step graph (dot) and list its non-kernel nodes."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

cfg = bench.WORKLOADS["criteo_c2"]
dev = torch.device("cuda:0")
model = bench.build_model(cfg, 100000, dev)
model.optim = torch.optim.Adam(model.parameters(), fused=True, capturable=True)
model.train()
batches = [(torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)) for X, y in
           bench.synthetic_batches(4, cfg["batch"], [100000] * cfg["n_sparse"], cfg["n_dense"], seed=1)]
sx, sy = batches[0][0].clone(), batches[0][1].clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for s in range(3):
        model.train_on_batch(sx, sy)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
g.enable_debug_mode()
with torch.cuda.graph(g):
    out = model.train_on_batch(sx, sy)
torch.cuda.synchronize()
path = os.path.join(ROOT, "gpurun_out", "train_step_graph.dot")
os.makedirs(os.path.dirname(path), exist_ok=True)
g.debug_dump(path)
txt = open(path).read()
labels = re.findall(r'label="([^"]*)"', txt)
print("nodes with labels:", len(labels))
kinds = {}
for lab in labels:
    k = lab.split("\\n")[0][:60]
    kinds[k] = kinds.get(k, 0) + 1
for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])[:60]:
    print("%4d  %s" % (v, k))
print("edges:", txt.count("->"))
