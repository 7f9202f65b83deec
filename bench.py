#!/usr/bin/env python3
"""Headline benchmark: xDeepFM training throughput (examples/sec) on BASELINE.json config 2
-- Criteo-1TB shape (26 sparse + 13 dense fields), per-GPU batch 4096, emb_dim 16,
cin_layer_size (256,128,128), dnn (256,256), fp32, synthetic data, random-init weights.

A "step" is one pass of the hot path over one resident batch, exactly the per-batch sequence of
the reference's fit loop (deepctr/models/basemodel.py:245-262): forward -> BCE(sum) -> L2 term ->
backward -> Adam step.  Inputs are in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  The timed K steps run the way `fit` runs them: replayed from the HIP graph
the model captured on its third step (single process; eager launches when row-parallel).  `roofline`
describes the dominant hand-written kernel by summed device time, measured live right after the timed
region over up to 10 of the same steps issued eagerly with HIP events around every launch on the launch
stream (events cannot be recorded inside a graph replay; `profiles/` holds the rocprofv3 summary of the
same command); `other_arithmetic` repeats the K steps with the other CIN arithmetic; `cpu_baseline` is
the CPU oracle (a port of the reference's op sequence) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "xdeepfm-pytorch_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
F16_MFMA_PEAK_TFLOPS = 2516.6      # dense f16/bf16 MFMA: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz (same guide, ~2.5 PF)
# cin_math = 1 ("f16x3"): every fp32 product of the contraction is three f16 MFMAs (hi*hi + hi*lo + lo*hi,
# fp32 accumulate), so the fp32-equivalent ceiling of that path is a third of the dense f16 peak.
CIN_MATH = {0: ("f32mfma", FP32_MFMA_PEAK_TFLOPS, "v_mfma_f32_32x32x2_f32 on fp32 operands"),
            1: ("f16x3", F16_MFMA_PEAK_TFLOPS / 3.0,
                "fp32 operands split into fp16 hi+lo, 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulate; "
                "peak = dense f16 MFMA peak / 3"),
            2: ("bf16", F16_MFMA_PEAK_TFLOPS,
                "operands rounded to bf16, 1 x v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate (BASELINE config 5's "
                "arithmetic; tolerance 2e-2 / 4e-2 of a tensor's largest magnitude, tests/test_gpu_parity.py)")}
HBM_PEAK_GBS = 8000.0
# K5 (attention block of config 3): scores and P.V as fp32 MFMAs (v_mfma_f32_32x32x2_f32) -- priced against the fp32 matrix peak
ATTN_PEAK_TFLOPS = FP32_MFMA_PEAK_TFLOPS
ATTN_NOTE = ("FLOPs of the reference's attention block per launch (Q K^T, P V and their gradients: 2 * 3 D S^2 forward, "
             "2 * 7 D S^2 backward per example and layer) against the fp32 MFMA peak")

# SURVEY.md 8(d) vocabulary presets.  "criteo-card": the 26 public Criteo-Kaggle cardinalities (C1..C26), all below the
# 2^24 - 1 limit of ids that travel as fp32 (deepctr/models/basemodel.py:242) -- 33.8 M rows, 574 M parameters; the
# ids themselves stay synthetic.  "mid": 1e5 rows per field (round 1's bench).  "small": 1e3 (BASELINE config 1).
CRITEO_CARD = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10,
               5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]


def preset_vocab(preset, n_sparse):
    if preset == "criteo-card":
        if n_sparse != len(CRITEO_CARD):
            raise SystemExit("the criteo-card preset has %d fields, the workload %d" % (len(CRITEO_CARD), n_sparse))
        return [min(v, (1 << 24) - 1) for v in CRITEO_CARD]
    return [{"mid": 100000, "small": 1000}[preset]] * n_sparse

WORKLOADS = {
    # BASELINE.json configs[1]
    "criteo_c2": dict(n_sparse=26, n_dense=13, emb_dim=16, cin=(256, 128, 128), dnn=(256, 256), batch=4096),
    # BASELINE.json configs[0] (CPU plumbing shape), handy for quick runs
    "criteo_c1": dict(n_sparse=26, n_dense=13, emb_dim=8, cin=(128, 128), dnn=(256, 256), batch=4096),
    # BASELINE.json configs[2]: xDeepFMAttention, same Criteo shape (script-default cin (256,128) -> 256 tokens)
    # BASELINE.json configs[4] shape on one GPU: Avazu (22 sparse fields, no dense), emb_dim 32, deep CIN, in the bf16
    # MFMA arithmetic the config names (cin_math 2; --cin-math 1 runs it in f16x3)
    "avazu_c5": dict(n_sparse=22, n_dense=0, emb_dim=32, cin=(512, 256, 256, 128), dnn=(256, 256), batch=4096),
    # xDeepFMPro = config 2's model + the SFG decoder (one vocabulary-wide softmax head per sparse field over
    # the positive rows); 1e5 rows per field unless a preset is named
    "criteo_pro": dict(n_sparse=26, n_dense=13, emb_dim=16, cin=(256, 128, 128), dnn=(256, 256), batch=4096, model="xDeepFMPro",
                       preset="mid"),
    "criteo_c3_attn": dict(n_sparse=26, n_dense=13, emb_dim=16, cin=(256, 128), dnn=(256, 256), batch=4096,
                           model="xDeepFMAttention"),
}


def synthetic_batches(n_batches, batch, vocab, n_dense, seed, uniform=False):
    """Criteo-shaped batches: Zipf-like ids floor(V*u^3) -- or uniform ids floor(V*u), the worst case for
    every cache between the tables and the gather -- dense ~ U(0,1), y ~ Bern(0.25)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_batches):
        ids = np.floor(np.asarray(vocab)[None, :] * rng.random((batch, len(vocab))) ** (1 if uniform else 3))
        ids = np.minimum(ids, np.asarray(vocab)[None, :] - 1)
        dense = rng.random((batch, n_dense))
        X = np.concatenate([ids, dense], axis=1).astype(np.float32)
        y = (rng.random((batch, 1)) < 0.25).astype(np.float32)
        out.append((X, y))
    return out


def build_model(cfg, vocab, device, lazy_rows=False):
    from deepctr.inputs import DenseFeat, SparseFeat
    from deepctr import models
    cols = [SparseFeat("C%d" % (i + 1), v, cfg["emb_dim"]) for i, v in enumerate(vocab)]
    cols += [DenseFeat("I%d" % (i + 1), 1) for i in range(cfg["n_dense"])]
    if cfg.get("model") == "xDeepFMPro":
        from deepctr.xdeepfm_pro import xDeepFMPro as cls
    else:
        cls = getattr(models, cfg.get("model", "xDeepFM"))
    model = cls(cols, cols, dnn_hidden_units=cfg["dnn"], cin_layer_size=cfg["cin"], l2_reg_dnn=1e-5, device=device)
    if lazy_rows:
        from xdfm_amd.optim import TableAdam
        model.compile(TableAdam(model.parameters(), lazy_rows=True), "binary_crossentropy", metrics=[])
    else:
        model.compile("adam", "binary_crossentropy", metrics=[])
    return model


def train_step(model, xb, yb, dp):
    """The body of the reference's batch loop (basemodel.py:245-262) as the model's own fit runs it,
    without the two .item() host syncs (losses stay on the device)."""
    return model.train_on_batch(xb, yb)[1]


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return min(n, 64)


def log(msg):
    print("[bench %7.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def cpu_baseline(cfg, vocab, rows, steps):
    """The CPU oracle (torch-CPU port of the reference's op sequence) on `rows` rows of the same
    workload: 1 warm-up + up to `steps` timed train steps (stops after ~25 s) on the usable cores."""
    from oracle import xdeepfm_oracle as orc
    threads = usable_cores()
    torch.set_num_threads(threads)
    names = ["C%d" % (i + 1) for i in range(cfg["n_sparse"])]
    dnames = ["I%d" % (i + 1) for i in range(cfg["n_dense"])]
    variant = {"xDeepFM": "sum", "xDeepFMAttention": "attn", "xDeepFMAttentionV2": "attn_v2"}[cfg.get("model", "xDeepFM")]
    spec = orc.Spec(names, list(vocab), dnames, cfg["emb_dim"], tuple(cfg["cin"]), True, "relu",
                    tuple(cfg["dnn"]), variant, l2_reg_dnn=1e-5)
    state = orc.init_state(spec)
    batches = [(torch.from_numpy(X), torch.from_numpy(y)) for X, y in
               synthetic_batches(steps + 1, rows, list(vocab), cfg["n_dense"], seed=7)]
    params = [p.requires_grad_(True) for p in state.values()]
    opt = torch.optim.Adam(params)

    def step(X, y):
        tot, _, _ = orc.total_loss(X, y, state, spec)
        opt.zero_grad()
        tot.backward()
        opt.step()
    step(*batches[0])
    log("cpu baseline warm-up step done (%d threads)" % threads)
    t0 = time.perf_counter()
    done = 0
    for X, y in batches[1:]:
        step(X, y)
        done += 1
        if time.perf_counter() - t0 > 25.0:
            break
    dt = time.perf_counter() - t0
    steps = done
    return dict(value=rows * steps / dt, unit="examples/sec", cores=threads, kind="port",
                sample="%d train steps of %d rows (same model/config, fp32, torch-CPU oracle), %.2f s/step"
                       % (steps, rows, dt / steps))


PMC_KERNEL = {"cin_level_bwd_x": ("cin_bwd_x3_kernel", "cin_bwd_x3_sym_kernel"), "cin_level_bwd_w": ("cin_bwd_w_x3_kernel",),
              "cin_level_fwd": ("cin_fwd_x3_kernel",)}


def pmc_traffic(bracket, workload, math_mode):
    """HBM-side bytes per launch of the roofline kernel.  Performance counters cannot be read from inside this process,
    so the figure comes from the committed rocprofv3 --pmc passes over THIS script (tools/profile_round.sh: bench.py with
    eager launches, XDFM_HIP_GRAPH=0 -- counter collection over the graph-replayed step hung in round 1; FETCH_SIZE and
    WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) --
    profiles/r03_pmc_bench_traffic.json, profiles/r03_pmc_traffic.md.  null for any other workload or arithmetic."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_pmc_bench_traffic.json")
    if workload != "criteo_c2" or math_mode != 1 or bracket not in PMC_KERNEL or not os.path.exists(path):
        return {}
    rows = [v for k, v in json.load(open(path)).items() if k.startswith(PMC_KERNEL[bracket])]
    n = sum(v["launches"] for v in rows)
    if not n:
        return {}
    total = sum(v["total_bytes"] * v["launches"] for v in rows) / n
    return dict(traffic=round(total), traffic_unit="bytes/launch (2 x FETCH_SIZE + WRITE_SIZE, mean over the levels)",
                traffic_source="profiles/r03_pmc_traffic.md (rocprofv3 --pmc over bench.py, eager launches)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="criteo_c2", choices=sorted(WORKLOADS))
    ap.add_argument("--vocab-preset", default=None, choices=["criteo-card", "mid", "small"],
                    help="SURVEY 8d vocabulary preset (default: criteo-card for the 26-field workloads, mid otherwise)")
    ap.add_argument("--vocab", type=int, default=0, help="uniform rows per embedding table (overrides the preset)")
    ap.add_argument("--no-extras", action="store_true", help="skip the mid-vocabulary and lazy-Adam side measurements")
    ap.add_argument("--cpu-rows", type=int, default=1024)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the second timed region with the other CIN arithmetic")
    ap.add_argument("--option", action="append", default=[], help="libxdfm tuning knob key=value")
    ap.add_argument("--cin-math", type=int, default=None, choices=[0, 1, 2],
                    help="CIN arithmetic: 0 fp32 MFMA, 1 f16x3 (default), 2 bf16 (default for --workload avazu_c5)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only "
                    "to rehearse the multi-rank path with several ranks on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ranks_seen = None
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run)"
                  % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        n_dev = torch.cuda.device_count()
        local_rank = local_rank % max(n_dev, 1)
        torch.cuda.set_device(local_rank)
        # fail fast and loudly: RCCL initialisation and one real collective before anything is built or timed
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(args.backend)
            probe = torch.ones(1, device=torch.device("cuda", local_rank) if args.backend == "nccl" else "cpu")
            dist.all_reduce(probe)
            if args.backend == "nccl":
                torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError("all-reduce of ones over %d ranks returned %r" % (world, probe.item()))
            seen = [None] * world
            dist.all_gather_object(seen, {"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(local_rank),
                                          "pid": os.getpid()})
            ranks_seen = seen
        except Exception as exc:      # noqa: BLE001
            print("bench.py: rank %d: %s backend failed at initialisation / first collective: %r" % (rank, args.backend, exc),
                  file=sys.stderr, flush=True)
            sys.exit(3)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from xdfm_amd import _lib, ops
    from xdfm_amd import dist as xdist
    for kv in args.option:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    cfg = WORKLOADS[args.workload]
    B = cfg["batch"]
    if args.cin_math is not None or args.workload == "avazu_c5":
        _lib.set_option("cin_math", args.cin_math if args.cin_math is not None else 2)    # config 5 names the bf16 MFMA path
    preset = args.vocab_preset or cfg.get("preset") or ("criteo-card" if cfg["n_sparse"] == len(CRITEO_CARD) else "mid")
    vocab = [args.vocab] * cfg["n_sparse"] if args.vocab > 0 else preset_vocab(preset, cfg["n_sparse"])
    vocab_name = ("%d rows/field" % args.vocab) if args.vocab > 0 else preset
    dp = xdist.current()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    class Run(object):
        """One model + its resident batches; every rank draws its own shard of the global batch (weak scaling)."""

        def __init__(self, vocab, lazy_rows=False, uniform=False, cfg_=None):
            self.model = build_model(cfg_ or cfg, vocab, device, lazy_rows)
            self.model.train()
            self.n_res = 8
            self.batches = [(torch.from_numpy(X).to(device), torch.from_numpy(y).to(device)) for X, y in
                            synthetic_batches(self.n_res, B, vocab, cfg["n_dense"], seed=2025 + rank, uniform=uniform)]
            if dp is not None:
                dp._n_global = B * world          # the scatter exchange needs the global split (equal shards)
            # untimed: the first steps of a batch shape run eagerly and the third one captures the HIP graph the
            # later steps are replayed from (xdfm_amd/graphstep.py); keep that out of the W + K steps
            for s in range(4):
                train_step(self.model, *self.batches[s % self.n_res], dp)
            self.gstep = self.model.__dict__.get("_graphed_step")

        def timed(self, steps, warmup, kernel_events=False):
            """W untimed steps, then exactly K steps between barrier + synchronize; MAX over ranks.
            kernel_events: bracket every heavy launch with HIP events on the launch stream (ops.PROFILE); the
            step then runs from eager launches (events cannot be recorded inside a replayed HIP graph)."""
            for s in range(warmup):
                train_step(self.model, *self.batches[s % self.n_res], dp)
            flush = getattr(self.model.optim, "flush", None)
            if flush is not None:
                flush()                                      # deferred table update: the region starts with nothing owed ...
            barrier()
            ops.PROFILE = [] if kernel_events else None     # (name, work, start_event, end_event) per heavy launch
            t0 = time.perf_counter()
            for s in range(steps):
                loss = train_step(self.model, *self.batches[s % self.n_res], dp)
            if flush is not None:
                flush()                                      # ... and pays everything its K steps deferred before the clock stops
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            self.dt_local = time.perf_counter() - t0          # this rank's own time, before it waits for the others
            barrier()
            dt = time.perf_counter() - t0
            prof, ops.PROFILE = ops.PROFILE, None
            if world > 1:
                import torch.distributed as dist
                t = torch.tensor([dt], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            if not np.isfinite(float(loss.item())):
                raise RuntimeError("non-finite loss in the benchmark loop")
            return dt, t_host, prof

    math_mode = _lib.get_option("cin_math")
    log("building model (vocabulary: %s, %.1f M rows)" % (vocab_name, sum(vocab) / 1e6))
    run = Run(vocab)
    log("warm-up + timed region (cin_math=%s)" % CIN_MATH[math_mode][0])
    dt, t_host, _ = run.timed(args.steps, args.warmup)
    replayed = run.gstep is not None and run.gstep.replays > 0
    log("timed region done: %.3f ms/step (host enqueue %.3f ms/step, %s)" % (
        dt / args.steps * 1e3, t_host / args.steps * 1e3,
        "HIP graph replay, %d nodes" % max(e.nodes for e in run.gstep.entries.values()) if replayed else "eager launches"))
    dt_local, host_local = run.dt_local, t_host
    # per-kernel device times for the roofline: the same steps from eager launches, HIP events around each launch
    _, _, prof = run.timed(min(args.steps, 10), 2, kernel_events=True)
    per_rank = None
    if world > 1:
        # every rank's own view of the timed region, so that a scaling run explains itself: its wall time before the final
        # barrier, its host time to enqueue a step (the second half of the row-parallel step is launched eagerly), and the
        # device time of ITS scatter (K2 runs over the rows of all ranks on every rank) and gather launches
        import torch.distributed as dist
        mine = {"rank": rank, "ms_per_step": round(dt_local / args.steps * 1e3, 4), "host_ms_per_step": round(host_local / args.steps * 1e3, 4)}
        agg = {}
        for name, work, e0, e1 in prof:
            if name.startswith("embed_"):
                agg.setdefault(name.replace("[bytes]", ""), []).append(e0.elapsed_time(e1))
        nprof = min(args.steps, 10)
        for k, v in agg.items():
            mine[k + "_ms_per_step"] = round(sum(v) / nprof, 4)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    alt = None
    if world == 1 and not args.no_alt:
        # the same K steps with the other arithmetic of the CIN contraction, for reference in the same line
        other = 1 - math_mode if math_mode in (0, 1) else 1          # beside bf16: the f16x3 (fp32-grade) measurement
        _lib.set_option("cin_math", other)
        adt, _, _ = run.timed(args.steps, 4)
        _lib.set_option("cin_math", math_mode)
        alt = dict(cin_math=CIN_MATH[other][0], value=round(B * args.steps / adt, 1), unit="examples/sec",
                   ms_per_step=round(adt / args.steps * 1e3, 4), arithmetic=CIN_MATH[other][2])
        log("other arithmetic (%s): %.3f ms/step" % (CIN_MATH[other][0], adt / args.steps * 1e3))
    extras = {}
    if world == 1 and not args.no_extras:
        # (0a) one LONG timed region of the same replayed step (400 steps, the periodic flushes of the deferred update inside
        # it and one at its end): the figure a 20-step region can only sample
        ldt, _, _ = run.timed(400, 0)
        extras["long_region"] = dict(steps=400, value=round(B * 400 / ldt, 1), unit="examples/sec", ms_per_step=round(ldt / 400 * 1e3, 4))
        log("long region (400 steps): %.3f ms/step" % (ldt / 400 * 1e3))
        # (0b) the reference's own loop around the step (basemodel.py:137-309 / :325-352): `fit` over 64 steps per epoch with
        # the epoch-end flush and loss read-back (per epoch: the difference of a 3-epoch and a 1-epoch call, which removes
        # the one-off host conversion of the inputs), and `predict` at the scripts' batch size 8192 (second call)
        names = list(run.model.feature_index.keys())
        Xf, yf = synthetic_batches(1, 64 * B, vocab, cfg["n_dense"], seed=77)[0]
        data = {n: Xf[:, i] for i, n in enumerate(names)}

        def fit_time(epochs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):
                run.model.fit(data, yf, batch_size=B, epochs=epochs, verbose=0, shuffle=False)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        fit_time(1)
        t1, t3 = fit_time(1), fit_time(3)
        per_epoch = max((t3 - t1) / 2.0, 1e-9)
        extras["fit_examples_per_sec"] = dict(value=round(64 * B / per_epoch, 1), unit="examples/sec", steps_per_epoch=64,
                                              ms_per_step=round(per_epoch / 64 * 1e3, 4),
                                              end_to_end_3_epochs=round(3 * 64 * B / t3, 1),
                                              note="BaseModel.fit (verbose=0, shuffle=False): per epoch = (3-epoch call - 1-epoch call) / 2, "
                                                   "epoch-end flush and loss read-back included; end_to_end also pays the host-side "
                                                   "conversion of the numpy inputs")
        npred = 262144
        pdata = {n: Xf[:npred, i] for i, n in enumerate(names)}
        run.model.predict(pdata, 8192)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run.model.predict(pdata, 8192)
        torch.cuda.synchronize()
        pt = time.perf_counter() - t0
        extras["predict_examples_per_sec"] = dict(value=round(npred / pt, 1), unit="examples/sec", rows=npred, batch_size=8192,
                                                  note="BaseModel.predict from numpy inputs to the float64 [N, 1] result on the host")
        run.model.train()
        log("fit: %.3f ms/step per epoch (%.0f ex/s), predict: %.0f ex/s" % (per_epoch / 64 * 1e3, 64 * B / per_epoch, npred / pt))
        del Xf, yf, data, pdata
        # (0c) the headline step on UNIFORM ids (the worst case for the gather / scatter / row-keyed optimizer)
        uni = Run(vocab, uniform=True)
        udt, _, _ = uni.timed(args.steps, args.warmup)
        extras["uniform_ids"] = dict(value=round(B * args.steps / udt, 1), unit="examples/sec", ms_per_step=round(udt / args.steps * 1e3, 4),
                                     note="same workload, ids ~ floor(V * u) instead of the Zipf-like floor(V * u^3)")
        log("uniform ids: %.3f ms/step" % (udt / args.steps * 1e3))
        del uni
        torch.cuda.empty_cache()
    if world == 1 and not args.no_extras and args.vocab <= 0 and preset == "criteo-card":
        # (1) the same step at the mid vocabulary (1e5 rows per field: round 1's headline): what the step costs when the
        # dense-Adam table sweep is small;  (2) the criteo-card step with the OPT-IN row-sparse ("lazy") Adam, which is
        # NOT the reference's arithmetic (untouched rows keep their moments): what the sweep costs the reference's way
        del run.batches
        run_keep_model = run.model                      # keep alive until the line is printed (state_dict sizes)
        mid = Run(preset_vocab("mid", cfg["n_sparse"]))
        mdt, _, _ = mid.timed(args.steps, args.warmup)
        extras["mid_vocab"] = dict(value=round(B * args.steps / mdt, 1), unit="examples/sec", ms_per_step=round(mdt / args.steps * 1e3, 4),
                                   vocab="1e5 rows per field (2.6 M rows, 44 M parameters)",
                                   optimizer="deferred table update" if getattr(mid.model.optim, "_def", None) is not None
                                   else "dense Adam sweep (TableAdam picks it below 64 M table parameters: same bits, faster there)")
        log("mid vocabulary: %.3f ms/step" % (mdt / args.steps * 1e3))
        del mid
        if args.workload == "criteo_c2":
            # (1b) SURVEY 8(f2): the same model with the SFG decoder (xDeepFMPro: a vocabulary-wide softmax head per sparse
            # field over the positive rows, csrc/vocab_ce_x3.hip), at the mid vocabulary; `--workload criteo_pro` is the full line
            pro = Run(preset_vocab("mid", cfg["n_sparse"]), cfg_=WORKLOADS["criteo_pro"])
            pdt, _, _ = pro.timed(10, 3)
            extras["sfg_pro_step"] = dict(value=round(B * 10 / pdt, 1), unit="examples/sec", ms_per_step=round(pdt / 10 * 1e3, 4),
                                          model="xDeepFMPro, 26 heads of 1e5 x 64, positive rows only (about a quarter of the batch)",
                                          parameters_M=round(sum(p.numel() for p in pro.model.parameters()) / 1e6, 1))
            log("xDeepFMPro (mid vocabulary): %.3f ms/step" % (pdt / 10 * 1e3))
            del pro
        lazy = Run(vocab, lazy_rows=True)
        ldt, _, _ = lazy.timed(args.steps, args.warmup)
        extras["lazy_adam_opt_in"] = dict(
            value=round(B * args.steps / ldt, 1), unit="examples/sec", ms_per_step=round(ldt / args.steps * 1e3, 4),
            note="xdfm_amd.optim.TableAdam(lazy_rows=True): rows a batch does not touch are not updated (no moment decay, no L2 "
                 "pull) -- a deviation from the reference's dense Adam (basemodel.py:452 over sparse=False tables), off by default")
        log("lazy-Adam opt-in: %.3f ms/step" % (ldt / args.steps * 1e3))
        del lazy
        torch.cuda.empty_cache()
        # (3) the dense Adam sweep over every table row in every step (XDFM_ADAM_DEFERRED=0): the same bits as the default
        # deferred update (tests/test_gpu_host.py), 24 bytes per table parameter and step of HBM traffic instead of ALU work
        dense = Run(vocab)
        dense.model.optim.flush()
        dense.model.optim.deferred = False
        dense.model.optim._invalidate()
        ddt, _, dprof = dense.timed(args.steps, args.warmup)
        extras["dense_adam_sweep"] = dict(
            value=round(B * args.steps / ddt, 1), unit="examples/sec", ms_per_step=round(ddt / args.steps * 1e3, 4),
            note="TableAdam(deferred=False): every row of every table updated in every step by the streaming kernel (K7); "
                 "bit-identical parameters and moments")
        log("dense Adam sweep: %.3f ms/step" % (ddt / args.steps * 1e3))
        del dense
        torch.cuda.empty_cache()

    if rank == 0:
        # Every step issues the same launch sequence, so launch k of a name is the same kernel on the same shape in
        # every profiled step: take the MEDIAN over the steps of each such launch (a host hiccup between the start
        # event and the launch -- allocator, GC -- would otherwise be billed to the kernel), then add them up.
        prof_steps = min(args.steps, 10)
        by_name = {}
        for name, work, e0, e1 in prof:
            by_name.setdefault(name, []).append((e0.elapsed_time(e1) * 1e-3, work))
        per_kernel = {}
        for name, ev in by_name.items():
            per_step = len(ev) // prof_steps if len(ev) % prof_steps == 0 and len(ev) >= prof_steps else 0
            if per_step:
                secs = sum(float(np.median([ev[s * per_step + k][0] for s in range(prof_steps)])) for k in range(per_step))
                work = sum(ev[k][1] for k in range(per_step))
                per_kernel[name] = [secs * prof_steps, work * prof_steps, len(ev)]
            else:
                per_kernel[name] = [sum(t for t, _ in ev), sum(w for _, w in ev), len(ev)]
        # the DOMINANT bracketed kernel by summed device time, whatever it is (CIN contraction, attention block, gather,
        # scatter, optimizer), against the roofline that bounds it
        cands = {k: v for k, v in per_kernel.items() if v[1] > 0 and not k.endswith("passes")}
        roof = None
        if cands:
            name, (secs, work, n) = max(cands.items(), key=lambda kv: kv[1][0])
            if name.endswith("[bytes]"):
                achieved, peak = work / secs / 1e9, HBM_PEAK_GBS
                roof = dict(kernel=name.replace("[bytes]", ""), bound="hbm", achieved=round(achieved, 1), peak=peak, unit="GB/s",
                            frac=round(achieved / peak, 4), traffic=None, launches=n, avg_ms=round(secs / n * 1e3, 4),
                            note="algorithmic bytes per launch (SURVEY.md 8d) / measured launch time")
            elif name.startswith("vocab_ce"):
                achieved, peak = work / secs / 1e12, CIN_MATH[1][1]
                roof = dict(kernel=name, bound="mfma", achieved=round(achieved, 2), peak=round(peak, 1), unit="TFLOP/s",
                            frac=round(achieved / peak, 4), traffic=None, launches=n, avg_ms=round(secs / n * 1e3, 4),
                            note="fp32-equivalent FLOPs of the reference's product (2 * positive rows * K * vocabulary rows of all "
                                 "heads per launch; the backward kernels recompute the logits on top of their own product, which "
                                 "is not counted); f16x3 on the matrix pipe, " + CIN_MATH[1][2])
            elif name.startswith("cin_attn_pool"):
                achieved, peak = work / secs / 1e12, ATTN_PEAK_TFLOPS
                roof = dict(kernel=name, bound="mfma", achieved=round(achieved, 2), peak=round(peak, 1), unit="TFLOP/s",
                            frac=round(achieved / peak, 4), traffic=None, launches=n, avg_ms=round(secs / n * 1e3, 4), note=ATTN_NOTE)
            else:
                achieved = work / secs / 1e12
                peak = CIN_MATH[math_mode][1]
                roof = dict(kernel=name, bound="mfma", achieved=round(achieved, 2), peak=round(peak, 1),
                            unit="TFLOP/s", frac=round(achieved / peak, 4), traffic=None,
                            launches=n, avg_ms=round(secs / n * 1e3, 4),
                            note="fp32-equivalent FLOPs of the reference's contraction (2*H*Hp*m*N per launch); level 0 "
                                 "(x_prev is x0) contracts over the pairs i <= j with folded weights and issues about half (50-54 %) "
                                 "of the MFMAs counted here for it; " + CIN_MATH[math_mode][2])
                roof.update(pmc_traffic(name, args.workload, math_mode))
        kernels = {}
        calls_per_step = {k: v[2] / prof_steps for k, v in per_kernel.items()}
        for k, v in sorted(per_kernel.items()):
            rate = v[1] / v[0] if v[0] > 0 else 0.0
            kernels[k.replace("[bytes]", "")] = dict(
                ms_per_step=round(v[0] / max(v[2], 1) * calls_per_step[k] * 1e3, 4),
                **({"GBps": round(rate / 1e9, 1), "frac_hbm": round(rate / 1e9 / HBM_PEAK_GBS, 4)}
                   if k.endswith("[bytes]") else ({} if (k.endswith("passes") or v[1] <= 0) else {"TFLOPs": round(rate / 1e12, 2)})))
        if getattr(run.model.optim, "_def", None) is not None and "adam_step" in kernels:
            # deferred table update: no sweep to price in GB/s -- these are the step's own optimizer kernels (dense weights,
            # small-table mark scan, big tables by the batch's rows); the catch-up before the gather and the periodic flush
            # are part of the line's ms_per_step
            kernels["adam_step"] = {"ms_per_step": kernels["adam_step"]["ms_per_step"],
                                    "note": "deferred table update: per-step optimizer kernels only (see `optimizer`)"}
        m_, nd_, D_ = cfg["n_sparse"], cfg["n_dense"], cfg["emb_dim"]
        gather_bytes = 4 * (m_ + nd_) + m_ * (4 * D_ + 4) + 4 * m_ * D_ + 4 * (m_ * D_ + nd_) + 4
        n_params = sum(p.numel() for p in run.model.parameters())
        out = {
            "metric": "examples/sec (xDeepFM train step, Criteo-shape synthetic, bs=4096 per GPU)",
            "value": round(B * world * args.steps / dt, 1),
            "unit": "examples/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if math_mode == 2 else "f32", "data": "synthetic",
            "dtype_detail": ("fp32 inputs, weights, accumulators, outputs and optimizer state in both CIN arithmetics; "
                             + CIN_MATH[math_mode][2] + (". Error against fp64 <= that of the fp32-MFMA kernels "
                             "(tests/test_gpu_parity.py::test_cin_f16x3_is_as_accurate_as_fp32_mfma); the strict fp32-MFMA "
                             "measurement of the same K steps is `other_arithmetic`" if math_mode == 1 else "")),
            "config": {"workload": "%s: %d sparse + %d dense, emb_dim %d, cin %s, dnn %s, vocabulary %s (%.1f M rows, %.0f M "
                                   "parameters), per-GPU batch %d, dense Adam + L2 as the reference, fp32" % (
                                       args.workload, cfg["n_sparse"], cfg["n_dense"], cfg["emb_dim"],
                                       list(cfg["cin"]), list(cfg["dnn"]), vocab_name, sum(vocab) / 1e6, n_params / 1e6, B),
                       "vocab_preset": vocab_name, "global_batch": B * world, "parallelism": "dp%d" % world},
            # the step's throughput against the HBM-gather roofline alone (5 308 algorithmic bytes per
            # example forward at this shape, 8 TB/s): what an embedding-only model could reach per GPU
            "hbm_gather_roofline": {"bytes_per_example": gather_bytes, "bound_examples_per_sec": round(HBM_PEAK_GBS * 1e9 / gather_bytes * world, 1),
                                    "frac": round(B * world * args.steps / dt / (HBM_PEAK_GBS * 1e9 / gather_bytes * world), 6)},
            "cin_math": CIN_MATH[math_mode][0],
            "launch": "hip_graph_replay" if replayed else "eager",
            "optimizer": ("Adam, the reference's dense update of every table row in every step, computed DEFERRED: rows are brought "
                          "up to date before a batch gathers them, when a gradient arrives, and every %d steps for all rows "
                          "(bit-identical parameters and moments, tests/test_gpu_host.py); every update the timed K steps owe is "
                          "paid inside the timed region (flush before the clock stops); `dense_adam_sweep` = the same step with "
                          "the per-step HBM sweep" % run.model.optim.flush_every)
            if getattr(run.model.optim, "_def", None) is not None else "Adam, dense sweep over every parameter in every step",
            "roofline": roof,
            "kernels": kernels,
        }
        out.update(extras)
        if ranks_seen is not None:
            out["ranks_seen"] = ranks_seen
            out["backend"] = args.backend
        if per_rank is not None:
            out["per_rank"] = per_rank
        if alt is not None:
            out["other_arithmetic"] = alt
        if world == 1 and not args.no_cpu_baseline:
            if "mid_vocab" in extras:
                del run                              # the CPU oracle needs the host memory and cores, not the GPU model
            out["cpu_baseline"] = cpu_baseline(cfg, vocab, args.cpu_rows, args.cpu_steps)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
