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
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, fused=True, capturable=True)
        self._armed = None          # id(parameter) -> L2 strength, for the next step only
        self._desc = {}             # group index -> (key, ctypes array of xdfm_adam_tensor)
        self.l2_value = None        # [1] device tensor: value of the armed L2 term at the last step
        self.grad_sources = []      # objects with .arenas() -> [ops.GradArena]: gradients K7 may read by their marks

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
                arr[k].grad, arr[k].grad_marks = gp, None
                for a in arenas:                       # a view of a kept gradient buffer: read it by its marks
                    mp = a.marks_ptr(gp)
                    if mp is not None and params[k].data_ptr() % 16 == 0:
                        arr[k].grad_marks = mp
                        a.consumed(gp)
                        break
            torch._foreach_add_(steps, 1)
            dev = params[0].device
            ws = val = None
            if l2 is not None and any(l2):
                ws = torch.empty(lib.xdfm_adam_step_ws_elems(T), dtype=torch.float32, device=dev)
                val = torch.empty(1, dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.xdfm_adam_step(ctypes.cast(arr, ctypes.c_void_p), T, float(group["lr"]), float(beta1),
                                          float(beta2), float(group["eps"]), ws.data_ptr() if ws is not None else None,
                                          val.data_ptr() if val is not None else None, stream), "adam_step")
            if val is not None:
                self.l2_value = val if self.l2_value is None else self.l2_value + val
        return None
