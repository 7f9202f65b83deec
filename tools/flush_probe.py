#!/usr/bin/env python3
"""Cost of the deferred table update's flush (K7d, csrc/adam.hip) at the bench's Criteo-card vocabulary: K train steps, then
the flush that replays them for every untouched chunk, timed with events on the launch stream.
usage (GPU box): python tools/flush_probe.py [steps ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    steps = [int(a) for a in sys.argv[1:]] or [8, 32, 64]
    dev = torch.device("cuda:0")
    cfg = bench.WORKLOADS["criteo_c2"]
    vocab = bench.preset_vocab("criteo-card", cfg["n_sparse"])
    model = bench.build_model(cfg, vocab, dev)
    model.train()
    batches = [(torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)) for X, y in
               bench.synthetic_batches(8, cfg["batch"], vocab, cfg["n_dense"], seed=2025)]
    for s in range(6):
        model.train_on_batch(*batches[s % 8])
    model.optim.flush()
    n_table = sum(p.numel() for k, p in model.named_parameters() if "embedding_dict" in k)
    for K in steps:
        for s in range(K):
            model.train_on_batch(*batches[s % 8])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        model.optim.flush()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print("flush after %3d steps: %8.3f ms = %.4f ms per replayed step, %.2f T element-updates/s (%.0f M table parameters)" % (
            K, ms, ms / K, n_table * K / ms / 1e9, n_table / 1e6), flush=True)


if __name__ == "__main__":
    main()
