"""Probe: does the HBM-bound Adam sweep (K7) overlap with the MFMA-bound CIN kernels when issued on a second
stream -- eagerly, and as a parallel branch of a captured HIP graph?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
import torch  # noqa: E402
from deepctr.layers import CIN  # noqa: E402
from xdfm_amd.optim import TableAdam  # noqa: E402

dev = torch.device("cuda:0")
layer = CIN(26, (256, 128, 128), "relu", True, 0.0, 1024, device="cpu").to(dev)
x = (torch.randn(4096, 26, 16, device=dev) * 0.5).requires_grad_(True)
tabs = [torch.nn.Parameter(torch.randn(100000, 16, device=dev) * 0.01) for _ in range(26)]
flat = torch.randn(26 * 1600000, device=dev) * 1e-3
for k, t in enumerate(tabs):
    t.grad = flat[k * 1600000:(k + 1) * 1600000].view(100000, 16)
opt = TableAdam(tabs, lr=1e-3)


def cin():
    out = layer(x)
    out.sum().backward()


def adam():
    opt.step()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


side = torch.cuda.Stream()


def both_two_streams():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        adam()
    cin()
    cur.wait_stream(side)


from xdfm_amd import _lib  # noqa: E402
if len(sys.argv) > 1:
    _lib.set_option("adam_bx", int(sys.argv[1]))
print("adam_bx", _lib.get_option("adam_bx"))
print("eager: cin fwd+bwd %.3f ms, adam sweep %.3f ms, serial %.3f ms, two streams %.3f ms" % (
    timeit(cin), timeit(adam), timeit(lambda: (cin(), adam())), timeit(both_two_streams)), flush=True)

# the same as a captured graph: serial chain vs forked branch
cap = torch.cuda.Stream()
for name, body in (("graph serial", lambda: (cin(), adam())), ("graph forked", None)):
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        for _ in range(2):
            cin(); adam()
    torch.cuda.current_stream().wait_stream(cap)
    torch.cuda.synchronize()
    for c in layer.conv1ds:
        c.weight.grad = c.bias.grad = None
    x.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        if body is not None:
            body()
        else:
            side.wait_stream(cap)
            with torch.cuda.stream(side):
                adam()
            cin()
            cap.wait_stream(side)
    torch.cuda.synchronize()
    print("%s: %.3f ms" % (name, timeit(g.replay)), flush=True)
