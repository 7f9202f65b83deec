import os, sys, time
sys.path.insert(0, "/root/repo/xdeepfm-pytorch_amd")
import torch
from xdfm_amd import _lib, ops
from deepctr.layers import CIN
dev = torch.device("cuda:0")
layer = CIN(26, (256, 128, 128), "relu", True, 0.0, 1024, device="cpu").to(dev)
x = torch.randn(4096, 26, 16, device=dev) * 0.5
for dbg in (0, 64, 128, 192):
    _lib.set_option("dbg", dbg)
    ops.PROFILE = []
    with torch.no_grad():
        for _ in range(12):
            layer(x)
    torch.cuda.synchronize()
    per = {}
    idx = 0
    for name, work, e0, e1 in ops.PROFILE:
        if name == "cin_level_fwd":
            per.setdefault(idx % 3, []).append(e0.elapsed_time(e1) * 1e3)
            idx += 1
    ops.PROFILE = None
    print("dbg=%d (bit6 no stores, bit7 one block only): level us" % dbg, {k: round(sorted(v)[len(v) // 2], 1) for k, v in per.items()}, flush=True)
_lib.set_option("dbg", 0)
