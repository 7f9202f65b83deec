"""torch.optim.Adam whose step is the library's streaming kernel (K7, `xdfm_adam_step`): one launch per 40
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

import torch

from . import _lib


class TableAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, lazy_rows=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, fused=True, capturable=True)
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
        out = super().load_state_dict(state_dict)
        self._invalidate()
        return out

    def add_param_group(self, param_group):
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

    def __getstate__(self):
        state = super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__.copy()
        state = dict(state)
        for k in ("_desc", "_lr_dev", "_armed", "l2_value"):      # ctypes descriptors / device scalars: rebuilt on use
            state[k] = {} if k in ("_desc", "_lr_dev") else None
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
            for k in range(T):
                gp = grads[k].data_ptr()
                arr[k].grad, arr[k].grad_marks, arr[k].flags = gp, None, 0
                for a in arenas:                       # a view of a kept gradient buffer: read it by its marks
                    mp = a.marks_ptr(gp)
                    if mp is not None and params[k].data_ptr() % 16 == 0:
                        arr[k].grad_marks = mp
                        arr[k].flags = 1 if (self.lazy_rows and params[k].dim() == 2 and params[k].shape[0] > 1) else 0
                        a.consumed(gp)
                        break
            torch._foreach_add_(steps, 1)
            dev = params[0].device
            ws = val = None
            if l2 is not None and any(l2):
                ws = torch.empty(lib.xdfm_adam_step_ws_elems(T), dtype=torch.float32, device=dev)
                val = torch.empty(1, dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            lr_dev = self._lr_dev.get(gi)
            from . import ops                          # per-kernel timing hook of bench.py (HIP events on the launch stream)
            nbytes = sum(params[k].numel() * (24.25 if arr[k].grad_marks else 28.0) for k in range(T))
            _lib.check(ops._run("adam_step[bytes]", nbytes, lambda: lib.xdfm_adam_step_lr(ctypes.cast(arr, ctypes.c_void_p), T, float(group["lr"]),
                                             lr_dev[1].data_ptr() if lr_dev is not None else None, float(beta1),
                                             float(beta2), float(group["eps"]), ws.data_ptr() if ws is not None else None,
                                             val.data_ptr() if val is not None else None, stream)), "adam_step")
            if val is not None:
                self.l2_value = val if self.l2_value is None else self.l2_value + val
        return None
