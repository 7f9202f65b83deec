"""A/B of forward-kernel variants selected by the "dbg" option, same process, same box: python tools/fwd_ab.py BIT"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
from xdfm_amd import _lib  # noqa: E402
from deepctr.layers import CIN  # noqa: E402

bit = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
torch.manual_seed(0)
layer = CIN(26, (256, 128, 128), "relu", True, 0.0, 1024, device="cpu").to(dev)
x = torch.randn(4096, 26, 16, device=dev) * 0.5
outs = {}
for rep in range(3):
    for opt in (0, bit):
        _lib.set_option("dbg", opt)
        with torch.no_grad():
            for _ in range(3):
                y = layer(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = layer(x)
            e1.record()
            torch.cuda.synchronize()
        outs[opt] = y
        print("dbg=%-5d forward of the CIN stack: %.1f us" % (opt, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
print("max |difference| between the variants:", float((outs[0] - outs[bit]).abs().max()))
_lib.set_option("dbg", 0)
