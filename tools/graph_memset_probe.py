"""Probe: is a memset node inside a captured HIP graph ordered against its neighbouring kernel nodes?
aten::sum over dim 0 of a [4096, 256] tensor zeroes a semaphore buffer with hipMemsetAsync before its
multi-block reduce kernel.  Replay the captured op back to back and compare with the eager result."""
import torch

dev = torch.device("cuda:0")
torch.manual_seed(0)
xs = [torch.randn(4096, 256, device=dev) for _ in range(8)]
want = [x.sum(0) for x in xs]
sx = xs[0].clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        y = (sx * 1.0).sum(0)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    t = sx * 1.0                       # a kernel node right before the memset + reduce
    y = t.sum(0)
    z = y * 2.0                        # and one right after
torch.cuda.synchronize()
worst = torch.zeros((), device=dev)
nbad = torch.zeros((), device=dev)
for it in range(4000):
    sx.copy_(xs[it % 8])
    g.replay()
    err = (z - 2.0 * want[it % 8]).abs().max()
    worst = torch.maximum(worst, err)
    nbad += (err > 1e-2).float()
torch.cuda.synchronize()
print("4000 replays of [mul, sum(0) (memset + reduce), mul]: worst abs error %.3e, wrong results in %d replays"
      % (worst.item(), int(nbad.item())))

# ---- same question for a device-to-device memcpy node (aten::copy_ of a contiguous tensor -> hipMemcpyAsync)
g2 = torch.cuda.CUDAGraph()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        u = torch.empty_like(sx); u.copy_(sx * 1.0)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
with torch.cuda.graph(g2):
    t2 = sx * 1.0
    u2 = torch.empty_like(t2)
    u2.copy_(t2)                       # memcpy node
    z2 = u2 * 2.0
torch.cuda.synchronize()
worst = torch.zeros((), device=dev)
nbad = torch.zeros((), device=dev)
for it in range(4000):
    sx.copy_(xs[it % 8])
    g2.replay()
    err = (z2 - 2.0 * xs[it % 8]).abs().max()
    worst = torch.maximum(worst, err)
    nbad += (err > 0).float()
torch.cuda.synchronize()
print("4000 replays of [mul, copy_ (memcpy node), mul]: worst abs error %.3e, wrong results in %d replays"
      % (worst.item(), int(nbad.item())))
