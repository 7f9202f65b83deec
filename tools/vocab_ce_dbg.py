"""Dev: where the weight-gradient kernel of the SFG heads (vx_ws_kernel) spends its time -- the kernel with parts switched
off (library option "dbg", bits 20..23; the results are wrong in these runs):  python tools/vocab_ce_dbg.py [rows] [vocab] [fields]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
import torch
from xdfm_amd import _lib, ops

a = [int(x) for x in sys.argv[1:]]
R, V, F_ = (a + [1024, 100000, 26][len(a):])[:3]
K = 64
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
h = torch.randn(R, K, generator=g).relu().to(dev).requires_grad_(True)
Ws = [(torch.rand(V, K, generator=g) * 0.25 - 0.125).to(dev).requires_grad_(True) for _ in range(F_)]
bs = [(torch.rand(V, generator=g) * 0.25 - 0.125).to(dev).requires_grad_(True) for _ in range(F_)]
tgt = torch.randint(0, V, (F_, R), generator=g).to(dev)
names = {0: "full kernel", 1: "no exp / target test / split", 2: "no dW MFMAs", 4: "no z MFMAs", 6: "no MFMAs at all",
         8: "no tile ring traffic, no barriers", 9: "8 + 1", 14: "8 + no MFMAs", 15: "everything off"}
for dbg in (0, 1, 2, 4, 6, 8, 9, 14, 15):
    _lib.set_option("dbg", dbg << 20)
    ts = []
    for it in range(4):
        for t in [h] + Ws + bs:
            t.grad = None
        ops.PROFILE = []
        (ops.vocab_heads_ce(h, tgt, Ws, bs).sum() / R).backward()
        torch.cuda.synchronize()
        ts.append([e0.elapsed_time(e1) for n, w, e0, e1 in ops.PROFILE if n == "vocab_ce_bwd_w"][0])
        ops.PROFILE = None
    print("dbg %2d  %-36s  bwd_w %.3f ms" % (dbg, names[dbg], sorted(ts)[1]), flush=True)
_lib.set_option("dbg", 0)
