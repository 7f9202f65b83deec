"""A/B of CIN kernel variants selected by library options, same process, same box, at the headline shape (or --c5):
    python tools/cin_ab.py x3_waves=0 x3_waves=4          (any library option; several per variant: a=1,b=2)
Prints per-launch device times (HIP events, median of 7 runs) of forward and backward and checks every variant's
outputs / gradients against the first one."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
from xdfm_amd import _lib, ops  # noqa: E402
from deepctr.layers import CIN  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
c5 = "--c5" in sys.argv
configs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) for a in args] or [{}]
dev = torch.device("cuda:0")
torch.manual_seed(0)
m, D, ls, B = (22, 32, (512, 256, 256, 128), 4096) if c5 else (26, 16, (256, 128, 128), 4096)
layer = CIN(m, ls, "relu", True, 0.0, 1024, device="cpu").to(dev)
x = (torch.randn(B, m, D, device=dev) * 0.5).requires_grad_(True)
ref = None
for rep in range(2):
    for cfg in configs:
        for k, v in cfg.items():
            if k == "lean":                      # pseudo-option: the forward's fused direct sums / sign bits (ops.CINStack)
                os.environ["XDFM_CIN_LEAN"] = str(v)
            else:
                _lib.set_option(k, v)
        for _ in range(3):
            layer.zero_grad(); x.grad = None
            y = layer(x); y.sum().backward()
        torch.cuda.synchronize()
        acc = {}
        for _ in range(7):
            ops.PROFILE = []
            layer.zero_grad(); x.grad = None
            y = layer(x); y.sum().backward()
            torch.cuda.synchronize()
            for k, (name, work, a, b) in enumerate(ops.PROFILE):
                acc.setdefault((k, name), []).append(a.elapsed_time(b) * 1e3)
            ops.PROFILE = None
        med = {kn: sorted(v)[3] for kn, v in acc.items()}
        tot = {}
        for (k, n), v in med.items():
            tot[n] = tot.get(n, 0.0) + v
        res = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
        if ref is None:
            ref = res
        err = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(res, ref))
        print("%-28s %s | %s | max rel diff vs first %.2e" % (
            ",".join("%s=%d" % kv for kv in cfg.items()),
            "  ".join("%s %.0f" % (n.replace("cin_level_", ""), v) for n, v in sorted(tot.items())),
            " ".join("%s:%.0f" % (n.replace("cin_level_", "")[:5], v) for (k, n), v in sorted(med.items()) if "passes" not in n), err), flush=True)
