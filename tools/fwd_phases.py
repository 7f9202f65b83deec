"""Where a forward launch of the f16x3 CIN kernel spends its time: every workgroup reports shader-clock and 100 MHz
tick counts of its prologue (x0 / column maxima / first stages), main loop and epilogue (library option dbg bit 4,
with bit 1 = no output stores so the reports survive).  Also gives the sustained shader clock under the kernel's load.
    python tools/fwd_phases.py [H Hp m N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
from xdfm_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
exps = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("exp=")] or [0]      # kernel-side experiments (EXP bits)
nums = [a for a in sys.argv[1:] if not a.startswith("exp=")]
shapes = [tuple(int(v) for v in nums[:4])] if len(nums) >= 4 else \
    [(256, 26, 26, 65536), (128, 128, 26, 65536), (128, 64, 26, 65536)]
shapes = [sh + (e,) for sh in shapes for e in exps]
st = torch.cuda.current_stream().cuda_stream
for H, Hp, m, N, exp in shapes:
    torch.manual_seed(0)
    W = torch.randn(H, Hp * m, device=dev) * 0.05
    x0 = torch.randn(m, N, device=dev)
    xp = x0 if Hp == m else torch.randn(Hp, N, device=dev)
    bias = torch.zeros(H, device=dev)
    out = torch.zeros(H, N, device=dev)
    pack = torch.empty(lib.xdfm_cin_fwd_pack_elems(H, Hp, m), dtype=torch.float32, device=dev)
    _lib.check(lib.xdfm_cin_fwd_pack(W.data_ptr(), H, Hp, m, pack.data_ptr(), st), "pack")

    def launch():
        _lib.check(lib.xdfm_cin_level_fwd(xp.data_ptr(), x0.data_ptr(), pack.data_ptr(), bias.data_ptr(), H, Hp, m, N, 1,
                                          out.data_ptr(), st), "fwd")
    _lib.set_option("dbg", (exp << 3) << 6)          # the experiment's kernel, stores on: whole-launch time by events
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        launch()
    e1.record()
    torch.cuda.synchronize()
    t_us = e0.elapsed_time(e1) * 100.0
    _lib.set_option("dbg", (5 | (exp << 3)) << 6)
    out.zero_()
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    _lib.set_option("dbg", 0)
    nwg = (N + 255) // 256 if N >= 256 * 64 else (N + 127) // 128
    r = out.flatten()[:nwg * 8].view(nwg, 8).double().cpu()
    cyc, tick = r[:, 0:3], r[:, 3:6]
    mhz = cyc.sum(1) / (tick.sum(1) / 100.0)
    us = tick.mean(0) / 100.0
    steps = (Hp // 8) * (m // 2) + ((Hp % 8 + 1) // 2 * m + 7) // 8
    print("exp=%d " % exp, end="")
    print("H=%d Hp=%d m=%d N=%d: launch %.1f us | per workgroup: prologue %.1f us, loop %.1f us (%d steps, %.0f clocks/step), "
          "epilogue %.1f us | shader clock %.0f MHz (min %.0f max %.0f)" % (
              H, Hp, m, N, t_us, us[0], us[1], steps, float(cyc[:, 1].mean()) / steps, us[2], float(mhz.mean()),
              float(mhz.min()), float(mhz.max())), flush=True)
