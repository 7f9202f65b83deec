"""Device time of the SFG heads' cross-entropy, fused (ops.VocabHeadsCE) against tiled (ops.VocabSoftmaxCE), forward and
backward, at the shapes of the pro step:  python tools/vocab_ce_probe.py [rows] [vocab] [fields]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
import torch
from xdfm_amd import ops

a = [int(x) for x in sys.argv[1:] if x.isdigit()]
FUSED_ONLY = "fused" in sys.argv
R, V, F_ = (a + [1024, 100000, 4][len(a):])[:3]
K = 64
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
h = torch.randn(R, K, generator=g).relu().to(dev).requires_grad_(True)
Ws = [(torch.rand(V, K, generator=g) * 0.25 - 0.125).to(dev).requires_grad_(True) for _ in range(F_)]
bs = [(torch.rand(V, generator=g) * 0.25 - 0.125).to(dev).requires_grad_(True) for _ in range(F_)]
tgt = torch.randint(0, V, (F_, R), generator=g).to(dev)


def fused():
    return ops.vocab_heads_ce(h, tgt, Ws, bs).sum() / R


def tiled():
    return sum(ops.vocab_softmax_ce(h, Ws[f], bs[f], tgt[f]).sum() for f in range(F_)) / R


def timed(fn):
    for _ in range(2):
        fn().backward()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    n = 5
    for _ in range(n):
        for t in [h] + Ws + bs:
            t.grad = None
        ev[0].record(); loss = fn(); ev[1].record(); loss.backward(); ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]) / n
        tb += ev[1].elapsed_time(ev[2]) / n
    return tf, tb, float(loss), [t.grad.clone() for t in [h] + Ws + bs]


rf = timed(fused)
if FUSED_ONLY:
    print("fused: fwd %.3f ms  bwd %.3f ms   loss %.6f" % rf[:3])
    sys.exit(0)
rt = timed(tiled)
print("rows %d, vocab %d x %d fields, K %d" % (R, V, F_, K))
print("fused: fwd %.3f ms  bwd %.3f ms   loss %.6f" % rf[:3])
print("tiled: fwd %.3f ms  bwd %.3f ms   loss %.6f" % rt[:3])
for name, a_, b_ in zip(["dh"] + ["dW%d" % f for f in range(F_)] + ["db%d" % f for f in range(F_)], rf[3], rt[3]):
    print("  %-5s max |fused - tiled| / max |tiled| = %.2e" % (name, float((a_ - b_).abs().max() / (b_.abs().max() + 1e-30))))
flops = 2.0 * R * V * K * F_
print("fused fp32-equivalent rate: fwd %.1f TFLOP/s, bwd (3 products + recompute x2) %.1f TFLOP/s" % (flops / rf[0] / 1e9, 4 * flops / rf[1] / 1e9))
