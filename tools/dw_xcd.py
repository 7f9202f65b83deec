"""dW kernel of the f16x3 / bf16 CIN arithmetic with the workgroups of an n-split kept on one XCD (library option
"bww_xcd" = 1, default) or in launch order (0): time of the whole call and of the MFMA kernel alone, and that the two
give the same bits.
    python tools/dw_xcd.py [H Hp m N] [xcd=0|1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
from xdfm_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
nums = [int(a) for a in sys.argv[1:] if "=" not in a]
shapes = [tuple(nums[:4])] if len(nums) >= 4 else [(128, 128, 26, 65536), (128, 64, 26, 65536), (256, 26, 26, 65536)]
ONLY = [int(a[4:]) for a in sys.argv[1:] if a.startswith("xcd=")]      # xcd=0 / xcd=1: one setting only (counter runs)
st = torch.cuda.current_stream().cuda_stream


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


for H, Hp, m, N in shapes:
    torch.manual_seed(0)
    x0 = torch.randn(m, N, device=dev)
    xp = x0 if Hp == m else torch.randn(Hp, N, device=dev)
    dOut = torch.randn(H, N, device=dev)
    ws = torch.empty(lib.xdfm_cin_bwd_w_ws_elems(H, Hp, m, N), dtype=torch.float32, device=dev)
    res = {}
    for xcd in ONLY or (0, 1, 0, 1):
        _lib.set_option("bww_xcd", xcd)
        dW = torch.zeros(H, Hp * m, device=dev)

        def call():
            _lib.check(lib.xdfm_cin_level_bwd_w(dOut.data_ptr(), xp.data_ptr(), x0.data_ptr(), H, Hp, m, N, ws.data_ptr(),
                                                dW.data_ptr(), st), "bwd_w")
        t = timed(call)
        _lib.set_option("bww_phase", 2)
        t2 = timed(call)
        _lib.set_option("bww_phase", 0)
        call()
        torch.cuda.synchronize()
        same = "" if xcd not in res else ("  same bits as launch order: %s" % bool(torch.equal(dW, res[0])))
        res.setdefault(xcd, dW.clone())
        print("H=%d Hp=%d m=%d N=%d bww_xcd=%d: call %.1f us, MFMA kernel %.1f us%s" % (H, Hp, m, N, xcd, t, t2, same), flush=True)
_lib.set_option("bww_xcd", 1)
