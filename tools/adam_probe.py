"""K7 alone on Criteo-card-sized tables: GB/s of the 24.25 B per parameter for grid caps (library option adam_bx) and
cached vs non-temporal accesses (dbg bit 17).   python tools/adam_probe.py [mid]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
sys.path.insert(0, ROOT)
from xdfm_amd import _lib  # noqa: E402
import bench  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
vocab = [100000] * 26 if "mid" in sys.argv else list(bench.CRITEO_CARD)
D = 16
sizes = [v * D for v in vocab] + [v for v in vocab] + [429 * 256, 256 * 256, 256, 256, 256 * 676, 128 * 3328, 128 * 1664, 256, 128, 128, 320, 256, 13]
T = len(sizes)
ps = [torch.randn(n, device=dev) * 0.01 for n in sizes]
ms = [torch.zeros(n, device=dev) for n in sizes]
vs = [torch.zeros(n, device=dev) for n in sizes]
gs = [torch.zeros(n, device=dev) for n in sizes]
marks = [torch.zeros((n + 3) // 4, dtype=torch.uint8, device=dev) for n in sizes]
steps = [torch.ones((), device=dev) for _ in sizes]
arr = (_lib.AdamTensor * T)()
for k in range(T):
    arr[k].param, arr[k].grad, arr[k].exp_avg, arr[k].exp_avg_sq = ps[k].data_ptr(), gs[k].data_ptr(), ms[k].data_ptr(), vs[k].data_ptr()
    arr[k].step, arr[k].numel, arr[k].l2 = steps[k].data_ptr(), sizes[k], 1e-5
    arr[k].grad_marks, arr[k].flags = marks[k].data_ptr(), 0
ws = torch.zeros(lib.xdfm_adam_step_ws_elems(T), device=dev)
l2v = torch.zeros(1, device=dev)
st = torch.cuda.current_stream().cuda_stream
total = sum(sizes)
print("%d tensors, %.1f M parameters" % (T, total / 1e6))
for dbg in (0, 1 << 17):
    for cap in (128, 512, 1024, 2048, 4096):
        _lib.set_option("adam_bx", cap)
        _lib.set_option("dbg", dbg)

        def launch():
            _lib.check(lib.xdfm_adam_step_lr(arr, T, 1e-3, None, 0.9, 0.999, 1e-8, ws.data_ptr(), l2v.data_ptr(), st), "adam")
        for _ in range(2):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms_ = e0.elapsed_time(e1) / 5
        print("%s cap %4d: %.3f ms  %.0f GB/s" % ("cached      " if dbg else "non-temporal", cap, ms_, total * 24.25 / ms_ / 1e6), flush=True)
_lib.set_option("adam_bx", 0)
_lib.set_option("dbg", 0)
