import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from deepctr.inputs import DenseFeat, SparseFeat
from deepctr.models import xDeepFM
from conftest import load_golden
mode = sys.argv[1]
dev = torch.device("cuda:0")
g = load_golden("fit_history")
vocab, nd, D = [int(v) for v in g["vocab"]], int(g["n_dense"]), int(g["emb_dim"])
print("vocab", vocab, "nd", nd, "D", D, "rows", g["X"].shape, flush=True)
cols = [SparseFeat("C%d" % (i + 1), v, D) for i, v in enumerate(vocab)] + [DenseFeat("I%d" % (i + 1), 1) for i in range(nd)]
model = xDeepFM(cols, cols, dnn_hidden_units=(8,), cin_layer_size=(6, 4), l2_reg_dnn=1e-5, device=dev)
model.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
names = list(model.feature_index.keys())
X, y = g["X"], g["y"]
if mode == "direct":
    Xd, yd = torch.from_numpy(X.astype(np.float32)).to(dev), torch.from_numpy(y.astype(np.float32)).reshape(-1, 1).to(dev)
    model.train()
    for s in range(5):
        out = model.train_on_batch(Xd[:64].contiguous(), yd[:64].contiguous())
        torch.cuda.synchronize()
        print("step", s, float(out[2]), flush=True)
elif mode == "views":
    Xd, yd = torch.from_numpy(X.astype(np.float32)).to(dev), torch.from_numpy(y.astype(np.float32)).reshape(-1, 1).to(dev)
    model.train()
    for s in range(5):
        out = model.train_on_batch(Xd[64 * (s % 2):64 * (s % 2) + 64], yd[64 * (s % 2):64 * (s % 2) + 64])
        print("step", s, float(out[2]), flush=True)
else:
    hist = model.fit({n: X[:, i] for i, n in enumerate(names)}, y, batch_size=64, epochs=2, verbose=0 if mode == "fit0" else 2, shuffle=False)
    print(hist.history, flush=True)
print("done", mode, model.__dict__["_graphed_step"].replays, flush=True)
