"""Train-step time of xDeepFMPro (SFG heads: one vocabulary-wide softmax per sparse field) on synthetic Criteo-shaped
batches:  python tools/pro_step.py [mid|card] [light]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
sys.path.insert(0, ROOT)
import torch
import bench
from deepctr.inputs import DenseFeat, SparseFeat
from deepctr.xdeepfm_pro import xDeepFMPro, xDeepFMProLight

dev = torch.device("cuda:0")
preset = "card" if "card" in sys.argv else "mid"
vocab = bench.preset_vocab("criteo-card" if preset == "card" else "mid", 26)
D, B = 16, 4096
cols = [SparseFeat("C%d" % (i + 1), v, D) for i, v in enumerate(vocab)] + [DenseFeat("I%d" % (i + 1), 1) for i in range(13)]
cls = xDeepFMProLight if "light" in sys.argv else xDeepFMPro
model = cls(cols, cols, cin_layer_size=(256, 128, 128), l2_reg_dnn=1e-5, device=dev)
model.compile("adam", "binary_crossentropy", metrics=[])
model.train()
n_params = sum(p.numel() for p in model.parameters())
batches = [(torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)) for X, y in bench.synthetic_batches(4, B, vocab, 13, seed=5)]
for s in range(3):
    model.train_on_batch(*batches[s % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 8
for s in range(K):
    out = model.train_on_batch(*batches[s % 4])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("%s, vocabulary %s (%.1f M rows), %.0f M parameters: %.2f ms/step = %.0f examples/s (loss %.4f); peak memory %.1f GB"
      % (cls.__name__, preset, sum(vocab) / 1e6, n_params / 1e6, dt * 1e3, B / dt, float(out[1]) / B,
         torch.cuda.max_memory_allocated() / 2**30))
