"""K1 / K2 alone over batch sizes: does the gather approach the HBM roofline once the launch is large enough?
Usage (GPU box): python tools/gather_scale.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
from xdfm_amd import _lib, ops  # noqa: E402

if len(sys.argv) > 1:
    _lib.set_option("dbg", int(sys.argv[1]))          # 512: 8 examples per workgroup, 256: 4

dev = torch.device("cuda:0")
m, nd, D, V = 26, 13, 16, int(sys.argv[2]) if len(sys.argv) > 2 else 100000
torch.manual_seed(0)
tabs = [torch.randn(V, D, device=dev).requires_grad_(True) for _ in range(m)]
lins = [torch.randn(V, 1, device=dev).requires_grad_(True) for _ in range(m)]
w = torch.randn(nd, 1, device=dev).requires_grad_(True)
plan = ops.EmbedPlan(list(range(m)), [V] * m, list(range(m, m + nd)), D)
for B in [int(b) for b in os.environ.get("XDFM_GATHER_B", "4096,16384,65536,262144,1048576").split(",")]:
    X = torch.cat([torch.randint(0, V, (B, m), device=dev).float(), torch.rand(B, nd, device=dev)], 1)
    fwd_bytes = B * (4 * (m + nd) + m * (4 * D + 4) + 4 * m * D + 4 * (m * D + nd) + 4)
    bwd_bytes = B * (4 * (m + nd) + 2 * 4 * m * D + 4 + 4 * m * (D + 1))
    outs = ops.EmbedGather.apply(X, w, plan, True, *tabs, *lins)
    gs = [torch.randn_like(o) for o in outs]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for it in range(3):
        e[0].record()
        for _ in range(5):
            outs = ops.EmbedGather.apply(X, w, plan, True, *tabs, *lins)
        e[1].record()
        for t in tabs + lins + [w]:
            t.grad = None
        e[2].record()
        torch.autograd.backward(outs, gs)
        e[3].record()
    torch.cuda.synchronize()
    tf, tb = e[0].elapsed_time(e[1]) / 5 * 1e-3, e[2].elapsed_time(e[3]) * 1e-3
    print("B=%8d  gather %8.1f us = %6.0f GB/s (%.2f of 8 TB/s)   scatter+zero-fill %8.1f us = %6.0f GB/s" % (
        B, tf * 1e6, fwd_bytes / tf / 1e9, fwd_bytes / tf / 8e12, tb * 1e6, bwd_bytes / tb / 1e9), flush=True)
