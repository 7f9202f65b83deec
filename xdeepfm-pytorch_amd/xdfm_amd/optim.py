"""torch.optim.Adam whose step is the library's streaming kernel (K7, `xdfm_adam_step`): one launch per 52
tensors for every fp32 CUDA parameter -- at BASELINE config 2 the embedding / linear tables are 44 M of the 45 M
parameters, ATen's multi-tensor kernel moves them at 3.2 TB/s, K7 at 5.7 TB/s, and the ~15 small tensors stop
costing a 47 us latency-bound launch of their own.  State layout (`step`, `exp_avg`, `exp_avg_sq` per
parameter, device-resident fp32 step counters) and hyper-parameters are torch's, so `state_dict()` /
`load_state_dict()` and code that edits `param_groups` keep working; anything the kernel does not implement
(amsgrad, weight decay, maximize, tensor learning rates, non-fp32 parameters) falls back to
`torch.optim.Adam.step`.  The update rule is the one basemodel.py:452 selects (torch.optim.Adam, defaults).

The model's own train step may *arm* an L2 term for one step (`arm_l2`): K7 then adds 2*l2*w to the gradients
while it streams the weights and returns the term's value (`l2_value`), which removes the regulariser's own
passes over the parameters (basemodel.py:412-428) from the step.

Gradients that are views of a kept gradient buffer (`ops.GradArena`, registered through `grad_sources`) are read by
their chunk marks: K7 skips the untouched rows of the dense table gradients and re-zeroes the touched ones."""
import ctypes
import os

import torch

from . import _lib

DEFER_CAP = 256          # steps the clock's constant table holds (the `last` bytes count steps since the last flush)
DEFER_MIN_NUMEL = int(os.environ.get("XDFM_ADAM_DEFER_MIN_NUMEL", 1 << 26))   # "auto": tables of fewer parameters in total take the dense sweep
ROWS_MIN_NUMEL = 1 << 20  # tables at least this large get the step's update by the batch's rows (XDFM_ADAM_ROWS_MIN_NUMEL overrides)


class TableAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, lazy_rows=False, deferred=None, flush_every=64):
        super().__init__(params, lr=lr, betas=betas, eps=eps, fused=True, capturable=True)
        # Deferred (exact) update of the tables, include/xdfm.h "K7d": same bits as the dense sweep, but a row is brought
        # up to date when a batch gathers it, when a gradient arrives for it, and every `flush_every` steps for all rows
        # -- the sweep's 24 bytes per table parameter are moved once per `flush_every` steps instead of every step.
        # Applies, like the marks, inside the model's own train step in a single process.  XDFM_ADAM_DEFERRED=0 turns it off.
        # `deferred`: True / False, or "auto" (default; XDFM_ADAM_DEFERRED = 1 / 0 / auto): deferred when the gathers' tables
        # hold at least DEFER_MIN_NUMEL parameters.  Both ways give the same bits; the deferred path costs ~0.27 ms per
        # step whatever the tables' size (catch-up and update by rows, the amortised flush), the sweep 5 us per million
        # parameters: 1.50 against 1.55 ms per step at 44 M table parameters, 4.1 against 2.13 ms at 575 M.
        env = os.environ.get("XDFM_ADAM_DEFERRED", "auto")
        self.deferred = (False if env == "0" else True if env == "1" else "auto") if deferred is None else \
            (deferred if deferred == "auto" else bool(deferred))
        self.flush_every = max(1, min(int(os.environ.get("XDFM_ADAM_FLUSH_EVERY", flush_every)), DEFER_CAP - 8))
        self._def = None            # clock, constants, per-table `last` bytes, backlog (built by the first deferred step)
        self._since = 0             # steps since the last flush (host count of what the device clock holds)
        self.path_counts = {"rows": 0, "scan": 0}      # deferred steps issued (or captured) by path: keyed by the batch's rows / by the mark bytes
        # OPT-IN deviation from the reference: rows of the embedding tables that a batch does not touch
        # are not updated at all (no moment decay, no L2 pull) -- "lazy" Adam.  The reference's dense Adam updates every
        # row every step; with `lazy_rows` the step's cost follows the batch instead of the vocabulary.  Applies only to
        # gradients that arrive through the kept, marked gradient buffer (the model's own train step).
        self.lazy_rows = bool(lazy_rows)
        self._armed = None          # id(parameter) -> L2 strength, for the next step only
        self._desc = {}             # group index -> (key, ctypes array of xdfm_adam_tensor)
        self.l2_value = None        # [1] device tensor: value of the armed L2 term at the last step
        self.grad_sources = []      # objects with .arenas() -> [ops.GradArena]: gradients K7 may read by their marks
        self._lr_dev = {}           # group index -> (host value, [1] float64 device tensor K7 reads the rate from)
        self.generation = 0         # bumped whenever state tensors may have been replaced (part of the graph key)

    # K7 takes the learning rate from device memory, so a captured train step follows `param_groups[i]["lr"]` edits
    # (schedules, the reference's lr override xdftrain.py:283-284) without a new capture.  The scalar is rewritten
    # OUTSIDE any capture: GraphedStep calls sync_lr() before every replay, step() calls it for eager launches.
    def sync_lr(self):
        for gi, group in enumerate(self.param_groups):
            lr = group["lr"]
            if not isinstance(lr, float) or not group["params"]:
                continue
            hit = self._lr_dev.get(gi)
            dev = group["params"][0].device
            if hit is None or hit[1].device != dev:
                if dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
                    continue
                hit = self._lr_dev[gi] = [None, torch.empty(1, dtype=torch.float64, device=dev)]
            if hit[0] != lr:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("xdfm TableAdam: learning rate changed inside a HIP-graph capture")
                hit[1].fill_(lr)
                hit[0] = lr

    def load_state_dict(self, state_dict):
        self.flush()                 # rows that still owe replayed steps get them from the OLD moments, before those go
        out = super().load_state_dict(state_dict)
        self._invalidate()
        return out

    def add_param_group(self, param_group):
        if hasattr(self, "_desc"):
            self.flush()
        out = super().add_param_group(param_group)
        if hasattr(self, "_desc"):
            self._invalidate()
        return out

    def _invalidate(self):
        """State tensors (exp_avg, exp_avg_sq, step) may have been replaced: forget the cached descriptors, and make
        every captured graph that baked their addresses stale (graphstep._signature hashes `generation`)."""
        self._desc = {}
        self._lr_dev = {}
        self.generation += 1
        self._auto_numel = None      # the "auto" decision follows the param groups
        self._drop_deferred()

    # ------------------------------------------------------------------ deferred update of the tables
    def _drop_deferred(self):
        d = self.__dict__.get("_def")
        if d is not None:
            for plan in d["plans"]:
                plan.catchup = None
        self._def = None
        self._since = 0

    def _deferred_state(self, dev, steps_done):
        if self._def is None:
            clock = torch.zeros(2, dtype=torch.int32, device=dev)
            clock[1] = int(steps_done)
            self._def = dict(clock=clock, consts=torch.zeros(4 * DEFER_CAP, dtype=torch.float32, device=dev),
                             backlog=torch.zeros(1, dtype=torch.int64, device=dev), last={}, l2={}, rows={}, plans=[],
                             tensors={})
            self._def["clk"] = _lib.AdamClock(clock.data_ptr(), self._def["consts"].data_ptr(), DEFER_CAP)
            for src in self.grad_sources:             # the gathers whose rows must be current before they are read
                if hasattr(src, "catchup"):
                    src.catchup = self._catchup
                    self._def["plans"].append(src)
            self.generation += 1                       # a step captured without the catch-up launch is stale
        return self._def

    def _rows_for_apply(self, d, params, grads, arr, deferred_now):
        """(plan, X, emb rows, lin rows, indices of the tensors K7 proper still handles) when the step's update of the BIG
        deferred tables can be keyed by the batch of the one gather that feeds them -- a single process, one gather whose
        fields are exactly the deferred tables, at least one table of ROWS_MIN_NUMEL elements -- else None (every deferred
        table is updated by the scan of its mark bytes)."""
        from . import dist as xdist
        # Big tables only (ROWS_MIN_NUMEL): in a small table an id occurs hundreds of times per batch and every occurrence
        # contends for the claim of the same `last` word (all tables by rows: 0.36 ms per step at the Criteo-card
        # benchmark against 0.18 ms for the scan); small tables stay with the step's mark scan, where they cost nothing.
        if xdist.current() is not None or len(d["plans"]) != 1 or os.environ.get("XDFM_ADAM_ROWS", "1") == "0":
            return None
        plan = d["plans"][0]
        if plan.last_gather is None:
            return None
        X, emb_tables, lin_tables = plan.last_gather
        index = {params[k].data_ptr(): k for k in deferred_now}
        fields = list(emb_tables) + list(lin_tables)
        if len(fields) != len(index) or any(t.data_ptr() not in index for t in fields) or X.shape[0] <= 0:
            return None
        min_numel = int(os.environ.get("XDFM_ADAM_ROWS_MIN_NUMEL", ROWS_MIN_NUMEL))
        key = ("apply", min_numel, tuple(t.data_ptr() for t in fields), tuple(grads[index[t.data_ptr()]].data_ptr() for t in fields))
        hit = d["rows"].get(key)
        if hit is None:
            dev = X.device
            mk = lambda vals: torch.tensor(vals, dtype=torch.int64, device=dev)

            by_rows_k = set()

            def table_of(ts):
                if not ts:
                    return None, None
                ent = [d["tensors"][t.data_ptr()] for t in ts]
                ks = [index[t.data_ptr()] for t in ts]
                big = [t.numel() >= min_numel for t in ts]
                by_rows_k.update(k for k, b in zip(ks, big) if b)
                arrs = (mk([t.data_ptr() if b else 0 for t, b in zip(ts, big)]), mk([e[0].data_ptr() for e in ent]),
                        mk([e[1].data_ptr() for e in ent]), mk([e[2].data_ptr() for e in ent]),
                        torch.tensor([e[3] for e in ent], dtype=torch.float32, device=dev),
                        mk([grads[k].data_ptr() for k in ks]), mk([arr[k].grad_marks for k in ks]))
                return _lib.AdamRows(*[a.data_ptr() for a in arrs]), arrs
            e_struct, e_keep = table_of(list(emb_tables))
            l_struct, l_keep = table_of(list(lin_tables))
            hit = d["rows"][key] = (e_struct, l_struct, e_keep, l_keep, frozenset(by_rows_k))
        if "cell" not in d:
            d["cell"] = torch.zeros(1, dtype=torch.int64, device=X.device)
        if not hit[4]:
            return None                                 # no table is big enough: everything by the scan
        rest = [k for k in range(len(params)) if k not in hit[4]]      # K7 proper: dense tensors + the small deferred tables
        if not rest:
            return None
        return plan, X, hit[0], hit[1], rest

    def _last_bytes(self, p):
        d = self._def
        key = p.data_ptr()
        hit = d["last"].get(key)
        if hit is None:
            hit = d["last"][key] = torch.zeros(p.numel() // 4 + 8, dtype=torch.uint8, device=p.device)
        return hit

    def _catchup(self, plan, X, emb_tables, lin_tables):
        """Called by the gather (ops.EmbedGather.forward) before it reads the rows of X."""
        d = self._def
        if d is None:
            return
        if self._since == 0 and not torch.cuda.is_current_stream_capturing():
            return
        key = (tuple(t.data_ptr() for t in emb_tables), tuple(t.data_ptr() for t in lin_tables))
        rows = d["rows"].get(key)
        if rows is None:
            def table_of(ts):
                if not ts:
                    return None, None
                ent = [d["tensors"].get(t.data_ptr()) for t in ts]
                if any(e is None for e in ent):
                    return None, None
                dev = ts[0].device
                mk = lambda vals: torch.tensor(vals, dtype=torch.int64, device=dev)
                arrs = (mk([t.data_ptr() for t in ts]), mk([e[0].data_ptr() for e in ent]), mk([e[1].data_ptr() for e in ent]),
                        mk([e[2].data_ptr() for e in ent]), torch.tensor([e[3] for e in ent], dtype=torch.float32, device=dev))
                return _lib.AdamRows(*([a.data_ptr() for a in arrs] + [None, None])), arrs
            e_struct, e_keep = table_of(emb_tables)
            l_struct, l_keep = table_of(lin_tables)
            rows = d["rows"][key] = (e_struct, l_struct, e_keep, l_keep)
        e_struct, l_struct = rows[0], rows[1]
        if e_struct is None or (lin_tables and l_struct is None):
            return                                      # tables this optimizer does not update by deferral
        group = self.param_groups[0]
        beta1, beta2 = group["betas"]
        cols, vocab, _, _ = plan.on(X.device)
        lib = _lib.load()
        _lib.check(lib.xdfm_adam_catchup_rows(
            X.data_ptr(), X.stride(0), X.shape[0], cols.data_ptr(), vocab.data_ptr(), plan.m, plan.D, ctypes.byref(e_struct),
            ctypes.byref(l_struct) if l_struct is not None else None, ctypes.byref(d["clk"]), float(beta1), float(beta2),
            float(group["eps"]), d["backlog"].data_ptr(), torch.cuda.current_stream(X.device).cuda_stream),
            "adam_catchup_rows")

    @torch.no_grad()
    def flush(self):
        """Every deferred chunk up to date; afterwards parameters and moments are what the dense sweep would hold."""
        d = self.__dict__.get("_def")
        if d is None or self._since == 0:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("xdfm TableAdam: flush inside a HIP-graph capture")
        ent = list(d["tensors"].items())
        if ent:
            arr = (_lib.AdamTensor * len(ent))()
            for k, (ptr, (m, v, last, l2, numel)) in enumerate(ent):
                arr[k].param, arr[k].exp_avg, arr[k].exp_avg_sq, arr[k].last = ptr, m.data_ptr(), v.data_ptr(), last.data_ptr()
                arr[k].numel, arr[k].l2, arr[k].flags = numel, l2, 2
            group = self.param_groups[0]
            beta1, beta2 = group["betas"]
            dev = d["clock"].device
            _lib.check(_lib.load().xdfm_adam_flush(ctypes.cast(arr, ctypes.c_void_p), len(ent), ctypes.byref(d["clk"]), float(beta1),
                                                   float(beta2), float(group["eps"]), d["backlog"].data_ptr(),
                                                   torch.cuda.current_stream(dev).cuda_stream), "adam_flush")
        self._since = 0

    def take_backlog(self):
        """L2 value of the replayed steps since the last call (a host float; syncs).  Over an epoch, the per-step L2 values
        plus this equal the dense path's sum."""
        d = self.__dict__.get("_def")
        if d is None:
            return 0.0
        v = float(d["backlog"].item()) / float(1 << 40)
        d["backlog"].zero_()
        return v

    def note_replay(self):
        """Called before a captured step is replayed (its Python does not run): periodic flush, step count."""
        if self.__dict__.get("_def") is not None:
            if self._since >= self.flush_every:
                self.flush()
            self._since += 1

    def state_dict(self):
        self.flush()
        return super().state_dict()

    def __getstate__(self):
        state = super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__.copy()
        state = dict(state)
        self.flush()
        for k in ("_desc", "_lr_dev", "_armed", "l2_value", "_def"):      # ctypes descriptors / device scalars: rebuilt on use
            state[k] = {} if k in ("_desc", "_lr_dev") else None
        state["_since"] = 0
        state["grad_sources"] = []
        state["lazy_rows"] = self.lazy_rows
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.setdefault("_desc", {})
        self.__dict__.setdefault("_lr_dev", {})
        self.__dict__.setdefault("_armed", None)
        self.__dict__.setdefault("l2_value", None)
        self.__dict__.setdefault("grad_sources", [])
        self.__dict__.setdefault("lazy_rows", False)
        self.__dict__.setdefault("deferred", False)
        self.__dict__.setdefault("flush_every", 64)
        self.__dict__["_def"] = None
        self.__dict__["_since"] = 0
        self.__dict__["_auto_numel"] = None
        self.generation = self.__dict__.get("generation", 0) + 1

    def owns(self, tensors):
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        return all(id(t) in mine for t in tensors)

    def arm_l2(self, tensors, coeffs):
        """The next step() adds the gradient of sum_t coeffs[t] * sum(tensors[t]^2) itself and reports its value."""
        self._armed = {}
        for t, c in zip(tensors, coeffs):
            self._armed[id(t)] = self._armed.get(id(t), 0.0) + float(c)

    def _plain(self, group):
        return (not group["amsgrad"] and group["weight_decay"] == 0 and not group["maximize"] and
                not group["differentiable"] and not group.get("decoupled_weight_decay", False) and
                isinstance(group["lr"], float) and all(isinstance(b, float) for b in group["betas"]) and
                getattr(self, "grad_scale", None) is None and getattr(self, "found_inf", None) is None)

    def _l2_by_hand(self, armed):
        """Fallback path: apply the armed term with ATen ops before torch's own step."""
        value = None
        for group in self.param_groups:
            for p in group["params"]:
                c = armed.get(id(p), 0.0)
                if c and p.grad is not None:
                    p.grad.add_(p.detach(), alpha=2.0 * c)
                    term = c * p.detach().square().sum()
                    value = term if value is None else value + term
        self.l2_value = None if value is None else value.reshape(1)

    @torch.no_grad()
    def step(self, closure=None):
        armed, self._armed = self._armed, None
        self.l2_value = None
        if closure is not None or not all(self._plain(g) for g in self.param_groups):
            if armed:
                self._l2_by_hand(armed)
            return super().step(closure)
        self._cuda_graph_capture_health_check()
        self.sync_lr()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            has_complex = self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            if not params:
                continue
            ok = not has_complex and all(
                p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous() and
                g.is_contiguous() and g.device == p.device for p, g in zip(params, grads))
            if not ok:
                raise RuntimeError("xdfm TableAdam: parameters and gradients must be contiguous fp32 CUDA tensors")
            beta1, beta2 = group["betas"]
            T = len(params)
            l2 = tuple(armed.get(id(p), 0.0) for p in params) if armed else None
            key = (tuple(p.data_ptr() for p in params), tuple(m.data_ptr() for m in exp_avgs), l2)
            hit = self._desc.get(gi)
            if hit is None or hit[0] != key:
                arr = (_lib.AdamTensor * T)()
                for k in range(T):
                    arr[k].param, arr[k].exp_avg, arr[k].exp_avg_sq = params[k].data_ptr(), exp_avgs[k].data_ptr(), exp_avg_sqs[k].data_ptr()
                    arr[k].step, arr[k].numel = steps[k].data_ptr(), params[k].numel()
                    arr[k].l2 = l2[k] if l2 is not None else 0.0
                hit = self._desc[gi] = (key, arr)
            arr = hit[1]
            arenas = [a for src in self.grad_sources for a in src.arenas() if a.pending]
            capturing = torch.cuda.is_current_stream_capturing()
            # row-parallel runs included: a rank's own rows are brought up to date before its gather, the rows the other
            # ranks touched arrive with their marks and are replayed inside the step -- every replica ends with the same bits
            defer_ok = bool(self.deferred) and not self.lazy_rows and gi == 0
            # the tables of the gathers that feed this optimizer (their rows are what a batch touches)
            table_ptrs = set()
            for src in self.grad_sources:
                lg = getattr(src, "last_gather", None)
                if lg is not None:
                    table_ptrs.update(t.data_ptr() for t in lg[1])
                    table_ptrs.update(t.data_ptr() for t in lg[2])
            if defer_ok and self.deferred == "auto":
                if self.__dict__.get("_auto_numel") is None:           # the tables' sizes do not change: decided once
                    self._auto_numel = sum(p.numel() for p in params if p.data_ptr() in table_ptrs)
                defer_ok = self._auto_numel >= DEFER_MIN_NUMEL
            deferred_now = []
            for k in range(T):
                gp = grads[k].data_ptr()
                arr[k].grad, arr[k].grad_marks, arr[k].flags, arr[k].last = gp, None, 0, None
                for a in arenas:                       # a view of a kept gradient buffer: read it by its marks
                    mp = a.marks_ptr(gp)
                    if mp is not None and params[k].data_ptr() % 16 == 0:
                        arr[k].grad_marks = mp
                        table = params[k].dim() == 2 and params[k].shape[0] > 1
                        arr[k].flags = 1 if (self.lazy_rows and table) else 0
                        if defer_ok and params[k].data_ptr() in table_ptrs:
                            deferred_now.append(k)
                        a.consumed(gp)
                        break
            dev = params[0].device
            d = self._def
            if gi != 0:
                # the deferred tables live in group 0 (`defer_ok`); a later group has nothing to do with their clock: it
                # must neither flush them nor tick `clock[1]` a second time for the same step
                pass
            elif d is not None and len(deferred_now) != len(d["tensors"]):
                # tables that were deferred arrive without marks (a user-driven loop, a row-parallel run): bring
                # everything up to date and take this step densely; the clock only notes that a step passed
                self.flush()
                deferred_now = []
                d["clock"][1:2].add_(1)
            elif deferred_now:
                if d is None:
                    if capturing:
                        deferred_now = []              # state is built by an eager step, never inside a capture
                    else:
                        d = self._deferred_state(dev, int(round(float(steps[deferred_now[0]].item()))))
                if d is not None:
                    if not capturing and self._since >= self.flush_every:
                        self.flush()
                    for k in deferred_now:
                        ptr = params[k].data_ptr()
                        ent = d["tensors"].get(ptr)
                        lk = float(l2[k]) if l2 is not None else 0.0
                        if ent is None or ent[3] != lk:
                            if ent is not None:
                                self.flush()           # the L2 strength of a table changed: the replays assumed the old one
                            ent = d["tensors"][ptr] = (exp_avgs[k], exp_avg_sqs[k], self._last_bytes(params[k]), lk, params[k].numel())
                            d["rows"] = {}
                        arr[k].flags, arr[k].last = 2, ent[2].data_ptr()
            torch._foreach_add_(steps, 1)
            ws = val = None
            if l2 is not None and any(l2):
                ws = torch.empty(lib.xdfm_adam_step_ws_elems(T), dtype=torch.float32, device=dev)
                val = torch.empty(1, dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            lr_dev = self._lr_dev.get(gi)
            from . import ops                          # per-kernel timing hook of bench.py (HIP events on the launch stream)
            nbytes = sum(params[k].numel() * (0.0625 if arr[k].flags == 2 else (24.25 if arr[k].grad_marks else 28.0)) for k in range(T))
            if deferred_now and d is not None:
                by_rows = self._rows_for_apply(d, params, grads, arr, deferred_now)
                self.__dict__.setdefault("path_counts", {"rows": 0, "scan": 0})["rows" if by_rows is not None else "scan"] += 1
                if by_rows is None:
                    # the mark bytes say which chunks have a gradient (row-parallel runs: rows of every rank)
                    _lib.check(ops._run("adam_step[bytes]", nbytes, lambda: lib.xdfm_adam_step_deferred(
                        ctypes.cast(arr, ctypes.c_void_p), T, ctypes.byref(d["clk"]), float(group["lr"]),
                        lr_dev[1].data_ptr() if lr_dev is not None else None, float(beta1), float(beta2), float(group["eps"]),
                        ws.data_ptr() if ws is not None else None, val.data_ptr() if val is not None else None, stream)),
                        "adam_step_deferred")
                else:
                    # single process: the chunks with a gradient are the rows of the batch -- no scan of the mark bytes
                    plan, X, e_struct, l_struct, rest = by_rows
                    arr2 = (_lib.AdamTensor * max(len(rest), 1))()
                    for j, k in enumerate(rest):
                        ctypes.memmove(ctypes.byref(arr2[j]), ctypes.byref(arr[k]), ctypes.sizeof(_lib.AdamTensor))
                    cols, vocab, _, _ = plan.on(X.device)

                    def launch():
                        rc = lib.xdfm_adam_step_deferred(
                            ctypes.cast(arr2, ctypes.c_void_p), len(rest), ctypes.byref(d["clk"]), float(group["lr"]),
                            lr_dev[1].data_ptr() if lr_dev is not None else None, float(beta1), float(beta2), float(group["eps"]),
                            ws.data_ptr() if ws is not None else None, val.data_ptr() if val is not None else None, stream)
                        if rc:
                            return rc
                        return lib.xdfm_adam_apply_rows(
                            X.data_ptr(), X.stride(0), X.shape[0], cols.data_ptr(), vocab.data_ptr(), plan.m, plan.D,
                            ctypes.byref(e_struct), ctypes.byref(l_struct) if l_struct is not None else None, ctypes.byref(d["clk"]),
                            float(beta1), float(beta2), float(group["eps"]), d["cell"].data_ptr(),
                            val.data_ptr() if val is not None else None, stream)
                    _lib.check(ops._run("adam_step[bytes]", nbytes, launch), "adam_step_deferred (rows)")
                if not capturing:
                    self._since += 1
            else:
                _lib.check(ops._run("adam_step[bytes]", nbytes, lambda: lib.xdfm_adam_step_lr(ctypes.cast(arr, ctypes.c_void_p), T, float(group["lr"]),
                                             lr_dev[1].data_ptr() if lr_dev is not None else None, float(beta1),
                                             float(beta2), float(group["eps"]), ws.data_ptr() if ws is not None else None,
                                             val.data_ptr() if val is not None else None, stream)), "adam_step")
            if val is not None:
                self.l2_value = val if self.l2_value is None else self.l2_value + val
        return None
