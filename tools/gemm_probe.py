"""DNN GEMM shapes of config 2 (fp32): hipBLASLt default heuristic vs TunableOp's pick.  GPU box: python tools/gemm_probe.py"""
import time
import torch

dev = torch.device("cuda:0")
B = 4096
shapes = [("fwd1 x[B,429] W[256,429]^T", (B, 429), (256, 429), "nt"),
          ("fwd2 x[B,256] W[256,256]^T", (B, 256), (256, 256), "nt"),
          ("dx2  g[B,256] W[256,256]", (B, 256), (256, 256), "nn"),
          ("dW2  g[B,256]^T x[B,256]", (B, 256), (B, 256), "tn"),
          ("dx1  g[B,256] W[256,429]", (B, 256), (256, 429), "nn"),
          ("dW1  g[B,256]^T x[B,429]", (B, 256), (B, 429), "tn")]


def run(tag):
    tot = 0.0
    for name, sa, sb, mode in shapes:
        a = torch.randn(sa, device=dev)
        b = torch.randn(sb, device=dev)
        f = {"nt": lambda: a @ b.t(), "nn": lambda: a @ b, "tn": lambda: a.t() @ b}[mode]
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        tot += us
        print("%-8s %-30s %7.1f us" % (tag, name, us), flush=True)
    print("%-8s total %.1f us" % (tag, tot), flush=True)


run("default")
t0 = time.time()
torch.cuda.tunable.enable(True)
torch.cuda.tunable.set_max_tuning_duration(20)
torch.cuda.tunable.set_max_tuning_iterations(20)
run("tunable")
print("tuning + run took %.1f s" % (time.time() - t0))
