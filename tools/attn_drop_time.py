"""K5 launch times with and without attention dropout at BASELINE config 3's size (B=4096, S=320, D=16, 4 heads)."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "xdeepfm-pytorch_amd"))
import torch
from xdfm_amd import ops
from deepctr.layers.cin_attention import MultiHeadSelfAttention, AttentionPooling

dev = torch.device("cuda:0")
B, S, D, nh = 4096, 320, 16, 4
torch.manual_seed(0)
fm = torch.randn(S, B * D, device=dev)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0      # token magnitude (the bench's tokens are ~1e-3)
fm = fm * scale
for p in (0.0, 0.1, 0.0):
    torch.manual_seed(1)
    att = MultiHeadSelfAttention(D, nh, p, dev).to(dev).train()
    ln = torch.nn.LayerNorm(D).to(dev)
    pool = AttentionPooling(D, D, dev).to(dev)
    x = fm.clone().requires_grad_(True)
    for _ in range(3):
        out = ops.attn_pool(x, B, D, [att], [ln], pool, True)
        out.sum().backward()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(10):
        ev[0].record()
        out = ops.attn_pool(x, B, D, [att], [ln], pool, True)
        ev[1].record()
        g = torch.ones_like(out)
        out.backward(g)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1])
        tb += ev[1].elapsed_time(ev[2])
    print("p_drop=%.1f  fwd %.1f us  bwd %.1f us" % (p, tf * 100, tb * 100), flush=True)
