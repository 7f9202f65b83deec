"""Level 0 of the CIN (x_prev is x0) with and without the folded pair list (library option "x3_sym"): time per launch by
events and the error against an fp64 contraction, forward / dX / dW.
    python tools/level0_sym.py [H m N] [math=2]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
from xdfm_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
math = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("math=")]
nums = [a for a in sys.argv[1:] if "=" not in a]
H, m, N = (int(v) for v in nums[:3]) if len(nums) >= 3 else (256, 26, 65536)
if math:
    _lib.set_option("cin_math", math[0])          # 1 = f16x3 (default), 2 = bf16
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
W = torch.randn(H, m * m, device=dev) * 0.05
x0 = torch.randn(m, N, device=dev)
bias = torch.randn(H, device=dev) * 0.1
dOut = torch.randn(H, N, device=dev)
# fp64 reference on a column subset
cols = torch.arange(0, N, max(1, N // 512), device=dev)
xs = x0[:, cols].double()
Z = (xs[:, None, :] * xs[None, :, :]).reshape(m * m, -1)
ref_out = torch.relu(W.double() @ Z + bias.double()[:, None])


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / reps


for sym in (0, 1):
    _lib.set_option("x3_sym", sym)
    pack = torch.empty(lib.xdfm_cin_fwd_pack_elems(H, m, m), dtype=torch.float32, device=dev)
    _lib.check(lib.xdfm_cin_fwd_pack(W.data_ptr(), H, m, m, pack.data_ptr(), st), "pack")
    out = torch.zeros(H, N, device=dev)

    def fwd():
        _lib.check(lib.xdfm_cin_level_fwd(x0.data_ptr(), x0.data_ptr(), pack.data_ptr(), bias.data_ptr(), H, m, m, N, 1,
                                          out.data_ptr(), st), "fwd")
    t = timed(fwd)
    d = out[:, cols].double() - ref_out
    err = d.abs().max().item() / ref_out.abs().max().item()
    rms = d.pow(2).mean().sqrt().item() / ref_out.pow(2).mean().sqrt().item()
    print("x3_sym=%d forward H=%d m=%d N=%d: %.1f us, max err / max |out| = %.2e, rms err / rms out = %.2e" % (
        sym, H, m, N, t, err, rms), flush=True)
_lib.set_option("x3_sym", 1)
