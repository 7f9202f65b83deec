"""Level 0 of the CIN (x_prev is x0) with and without the folded pair list (library option "x3_sym"): time per launch by
events and the error against an fp64 contraction, forward / dX / dW.
    python tools/level0_sym.py [H m N] [math=2] [opt=name:value ...]"""
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
for a in sys.argv[1:]:
    if a.startswith("opt="):                      # any library option, e.g. opt=x3_fwd_mt:4 opt=x3_waves:4
        k, v = a[4:].split(":")
        _lib.set_option(k, int(v))
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

# dX: gradient of sum(dOut * (W Z)) wrt x0, both factors
xs_g = x0[:, cols].double().requires_grad_(True)
Zg = (xs_g[:, None, :] * xs_g[None, :, :]).reshape(m * m, -1)
((W.double() @ Zg) * dOut[:, cols].double()).sum().backward()
ref_g = xs_g.grad
Hc = min(H, 256)                                   # rows of the contraction per dX launch
Wc = W[:Hc].contiguous()
Zg = (xs_g[:, None, :] * xs_g[None, :, :]).reshape(m * m, -1)
xs_g.grad = None
((Wc.double() @ Zg) * dOut[:Hc, cols].double()).sum().backward()
ref_g = xs_g.grad
for sym in (0, 1):
    _lib.set_option("x3_sym", sym)
    wz = torch.empty(lib.xdfm_cin_bwd_pack_elems(Hc, m, m), dtype=torch.float32, device=dev)
    _lib.check(lib.xdfm_cin_bwd_pack(Wc.data_ptr(), Hc, m, m, wz.data_ptr(), st), "bwd pack")
    dxp = torch.full((m, N), 7.0, device=dev)
    dx0 = torch.full((m, N), 7.0, device=dev)
    dOc = dOut[:Hc].contiguous()

    def bwx():
        _lib.check(lib.xdfm_cin_level_bwd_x_ex(dOc.data_ptr(), x0.data_ptr(), x0.data_ptr(), wz.data_ptr(), Hc, m, m, N,
                                               dxp.data_ptr(), dx0.data_ptr(), 3, st), "bwd_x")
    t = timed(bwx)
    d = (dxp + dx0)[:, cols].double() - ref_g
    err = d.abs().max().item() / ref_g.abs().max().item()
    rms = d.pow(2).mean().sqrt().item() / ref_g.pow(2).mean().sqrt().item()
    # accumulate mode: a second call with flags 0 doubles the total
    _lib.check(lib.xdfm_cin_level_bwd_x_ex(dOc.data_ptr(), x0.data_ptr(), x0.data_ptr(), wz.data_ptr(), Hc, m, m, N,
                                           dxp.data_ptr(), dx0.data_ptr(), 0, st), "bwd_x")
    d2 = (dxp + dx0)[:, cols].double() - 2 * ref_g
    err2 = d2.abs().max().item() / ref_g.abs().max().item()
    print("x3_sym=%d dX H=%d m=%d N=%d: %.1f us, max err / max |g| = %.2e, rms err / rms g = %.2e; accumulated twice: %.2e" % (
        sym, Hc, m, N, t, err, rms, err2), flush=True)

# dW: sum_n dOut[h][n] x0[i][n] x0[j][n], all columns, fp64
xd = x0.double()
Zf = (xd[:, None, :] * xd[None, :, :]).reshape(m * m, N)
ref_w = dOut.double() @ Zf.t()
del Zf
for sym in (0, 1):
    _lib.set_option("x3_sym", sym)
    ws = torch.empty(lib.xdfm_cin_bwd_w_ws_elems(H, m, m, N), dtype=torch.float32, device=dev)
    dW = torch.full((H, m * m), 7.0, device=dev)

    def bww():
        _lib.check(lib.xdfm_cin_level_bwd_w(dOut.data_ptr(), x0.data_ptr(), x0.data_ptr(), H, m, m, N, ws.data_ptr(),
                                            dW.data_ptr(), st), "bwd_w")
    t = timed(bww)
    _lib.set_option("bww_phase", 2)
    t2 = timed(bww)
    _lib.set_option("bww_phase", 0)
    bww()
    d = dW.double() - ref_w
    err = d.abs().max().item() / ref_w.abs().max().item()
    rms = d.pow(2).mean().sqrt().item() / ref_w.pow(2).mean().sqrt().item()
    print("x3_sym=%d dW H=%d m=%d N=%d: %.1f us (MFMA kernel alone %.1f us), max err / max |dW| = %.2e, rms err / rms = %.2e, "
          "kernel %d" % (sym, H, m, N, t, t2, err, rms, _lib.get_option("last_bww_kernel")), flush=True)
_lib.set_option("x3_sym", 1)
