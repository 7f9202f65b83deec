"""Dev check of the f16x3 CIN path: error of both arithmetic modes against an fp64 reference, and timing.
Usage (GPU box): python tools/x3_check.py [fwd|all]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
from xdfm_amd import _lib, ops  # noqa: E402
from deepctr.layers import CIN  # noqa: E402


def ref64(x, Ws, Bs):
    """CIN forward/backward in fp64 with torch ops (deepctr/layers/interaction.py:207-248)."""
    x = x.double().detach().requires_grad_(True)
    Ws = [w.double().detach().requires_grad_(True) for w in Ws]
    Bs = [b.double().detach().requires_grad_(True) for b in Bs]
    B, m, D = x.shape
    hidden, finals = x, []
    for i, (w, b) in enumerate(zip(Ws, Bs)):
        z = torch.einsum("bhd,bmd->bhmd", hidden, x).reshape(B, hidden.shape[1] * m, D)
        cur = torch.relu(torch.nn.functional.conv1d(z, w, b))
        if i != len(Ws) - 1:
            hidden, direct = cur[:, : w.shape[0] // 2], cur[:, w.shape[0] // 2:]
        else:
            direct = cur
        finals.append(direct)
    return torch.cat(finals, 1).sum(-1), x, Ws, Bs


def run(B, m, D, ls, scale=0.5, spread=False, do_bwd=True, seed=0):
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    layer = CIN(m, ls, "relu", True, 0.0, 1024, device="cpu").to(dev)
    x = torch.randn(B, m, D, device=dev) * scale
    if spread:                                         # per-example magnitudes over 8 decades
        x = x * torch.pow(10.0, torch.rand(B, 1, 1, device=dev) * 8 - 6)
    Ws = [c.weight for c in layer.conv1ds]
    Bs = [c.bias for c in layer.conv1ds]
    want, x64, W64, B64 = ref64(x, Ws, Bs)
    gout = torch.randn(want.shape, device=dev)
    if do_bwd:
        (want * gout.double()).sum().backward()
    res = {}
    for mode in (0, 1):
        _lib.set_option("cin_math", mode)
        xg = x.clone().requires_grad_(True)
        for c in layer.conv1ds:
            c.weight.grad = None
            c.bias.grad = None
        out = layer(xg)
        den = want.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
        e = ((out.double() - want) / den).abs()
        line = "fwd max %.2e rms %.2e" % (e.max().item(), e.pow(2).mean().sqrt().item())
        if do_bwd:
            (out * gout).sum().backward()
            def rel(a, b):
                return ((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()
            line += "  dx %.2e" % rel(xg.grad, x64.grad)
            line += "  dW " + " ".join("%.2e" % rel(c.weight.grad, w.grad) for c, w in zip(layer.conv1ds, W64))
        res[mode] = line
        print("B=%d m=%d D=%d ls=%s spread=%s mode=%d: %s" % (B, m, D, ls, spread, mode, line), flush=True)
    _lib.set_option("cin_math", 0)


def timing(B, m, D, ls, do_bwd=True):
    dev = torch.device("cuda:0")
    layer = CIN(m, ls, "relu", True, 0.0, 1024, device="cpu").to(dev)
    x = (torch.randn(B, m, D, device=dev) * 0.5).requires_grad_(True)
    flops = sum(2.0 * H * Hp * m * B * D for (H, Hp, *_r) in ops.cin_geometry(m, ls, True)[0])
    for mode in (0, 1):
        _lib.set_option("cin_math", mode)
        for it in range(3):
            out = layer(x)
            if do_bwd:
                out.sum().backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 10
        for it in range(K):
            out = layer(x)
            if do_bwd:
                out.sum().backward()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print("timing mode=%d: %.3f ms  -> %.1f TFLOP/s (fp32-equivalent, %s)" % (
            mode, dt * 1e3, flops * (3 if do_bwd else 1) / dt / 1e12, "fwd+bwd" if do_bwd else "fwd"), flush=True)
    _lib.set_option("cin_math", 0)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "time":                      # full-size launches only (for rocprofv3 --pmc runs)
        timing(4096, 26, 16, (256, 128, 128), do_bwd=True)
        sys.exit(0)
    bwd = what != "fwd"
    run(130, 26, 16, (64, 32, 32), do_bwd=bwd)
    run(257, 26, 8, (128, 128), do_bwd=bwd)
    run(33, 26, 16, (256,), do_bwd=bwd)
    run(45, 22, 32, (136, 96), do_bwd=bwd)
    run(6, 22, 32, (512, 256, 256, 128), do_bwd=bwd)
    run(512, 26, 16, (256, 128, 128), do_bwd=bwd)
    run(512, 26, 16, (256, 128, 128), scale=1e-4, do_bwd=bwd)
    run(512, 26, 16, (256, 128, 128), spread=True, do_bwd=bwd)
    timing(4096, 26, 16, (256, 128, 128), do_bwd=False)
    if bwd:
        timing(4096, 26, 16, (256, 128, 128), do_bwd=True)
