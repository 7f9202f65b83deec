"""Smallest program that launches the CIN kernels of the headline config (B=4096, m=26, D=16, layers 256,128,128)
a few times, for `rocprofv3 --pmc` passes (FETCH_SIZE and WRITE_SIZE in separate runs; see profiles/).
Usage (GPU box): rocprofv3 --kernel-trace --pmc FETCH_SIZE -d OUT -o pmc -f csv -- python3 tools/pmc_cin.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
from deepctr.layers import CIN  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
layer = CIN(26, (256, 128, 128), "relu", True, 0.0, 1024, device="cpu").to(dev)
x = (torch.randn(4096, 26, 16, device=dev) * 0.5).requires_grad_(True)
for it in range(4):
    out = layer(x)
    out.sum().backward()
torch.cuda.synchronize()
print("done", flush=True)
