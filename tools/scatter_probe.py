"""K2 alone at the headline shape, with the kernel's phases switched off one at a time (xdfm option "dbg" bits 12..16;
timing only -- the results of the ablated runs are wrong by construction).  python tools/scatter_probe.py [B] [vocab]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
import numpy as np, torch
from xdfm_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
V = sys.argv[2] if len(sys.argv) > 2 else "100000"
m, D, nd = 26, 16, 13
CARD = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10,
        5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
VS = CARD if V == "card" else [int(V)] * m
rng = np.random.default_rng(0)
X = np.zeros((B, m + nd), dtype=np.float32)
X[:, :m] = np.floor(np.asarray(VS)[None, :] * rng.random((B, m)) ** 3)
X[:, m:] = rng.random((B, nd))
sizes = [v * D for v in VS] + list(VS)
offs, off = [], 0
for n in sizes:
    offs.append(off); off += (n + 3) // 4 * 4
total = off
flat = torch.zeros(total + nd + 3, device=dev)
marks = torch.zeros(flat.numel() // 4 + 2, dtype=torch.uint8, device=dev)
i32 = dict(dtype=torch.int32, device=dev)
cols, voc, dcols = torch.arange(m, **i32), torch.tensor(VS, **i32), torch.arange(m, m + nd, **i32)
off_dev = torch.tensor(offs, dtype=torch.int64, device=dev)
Xd = torch.from_numpy(X).to(dev)
de = torch.randn(m, B * D, device=dev); dd = torch.randn(B, m * D + nd, device=dev); dl = torch.randn(B, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def call():
    _lib.check(lib.xdfm_embed_scatter_bwd_marked(P(Xd), Xd.stride(0), B, P(cols), P(voc), m, D, P(dcols), nd, P(de), P(dd), 0,
                                                P(dl), 0, P(flat), P(off_dev[:m]), P(off_dev[m:]), P(flat[total:]), P(marks), st), "scatter")
def timeit(n=20):
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
names = {0: "full", 1: "no grouping", 2: "no row loads", 4: "no RMW", 8: "no cross-window phases", 16: "no X loads",
         1 | 2 | 4 | 8 | 16: "nothing but key build + window logic", 2 | 4: "no loads, no RMW", 1 | 16: "no grouping, no X"}
for bits, name in names.items():
    _lib.set_option("dbg", bits << 12)
    print("%-40s %8.1f us" % (name, timeit()), flush=True)
_lib.set_option("dbg", 0)
