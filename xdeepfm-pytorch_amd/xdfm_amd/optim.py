"""torch.optim.Adam whose step sends the gather's tables (44 M of the 45 M parameters at BASELINE config 2)
through one streaming launch of the library (K7, `xdfm_adam_tables`) and every other tensor through ATen's
fused kernel.  State layout (`step`, `exp_avg`, `exp_avg_sq` per parameter, device-resident fp32 step counters)
and hyper-parameters are torch's, so `state_dict()` / `load_state_dict()` and code that edits `param_groups`
keep working; anything the kernel does not implement (amsgrad, weight decay, maximize, tensor learning rates)
falls back to `torch.optim.Adam.step`.  The update rule is the one basemodel.py:452 selects (torch.optim.Adam
with default arguments).

The model's own train step may *arm* the tables' L2 term for one step (`arm_table_l2`): K7 then adds 2*l2*w to
the gradients while it streams the weights and returns the term's value (`table_l2_value`), which saves the two
table-sized passes the regulariser would otherwise need (basemodel.py:412-428)."""
import torch
from torch.optim.adam import adam as _functional_adam

from . import _lib


class TableAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, tables=None):
        super().__init__(params, lr=lr, betas=betas, eps=eps, fused=True, capturable=True)
        self._table_ids = {id(t): k for k, t in enumerate(tables or [])}
        self._cache = {}            # group index -> (key, device pointer tables)
        self._armed = None          # per-table L2 strengths for the next step only
        self.table_l2_value = None  # [1] device tensor: value of the armed L2 term at the last step

    def owns(self, tensors):
        return len(self._table_ids) > 0 and all(id(t) in self._table_ids for t in tensors)

    def arm_table_l2(self, tensors, coeffs):
        """The next step() adds the gradient of sum_t coeffs[t] * sum(tensors[t]^2) itself and reports its value."""
        self._armed = {id(t): float(c) for t, c in zip(tensors, coeffs)}

    def _plain(self, group):
        return (not group["amsgrad"] and group["weight_decay"] == 0 and not group["maximize"] and
                not group["differentiable"] and not group.get("decoupled_weight_decay", False) and
                isinstance(group["lr"], float) and all(isinstance(b, float) for b in group["betas"]) and
                getattr(self, "grad_scale", None) is None and getattr(self, "found_inf", None) is None)

    @staticmethod
    def _l2_by_hand(params, grads, armed):
        """Fallback when the tables cannot take the streaming kernel: apply the armed term with ATen ops."""
        value = None
        for p, g in zip(params, grads):
            c = armed.get(id(p), 0.0)
            if c:
                g.add_(p.detach(), alpha=2.0 * c)
                term = c * p.detach().square().sum()
                value = term if value is None else value + term
        return value

    @torch.no_grad()
    def step(self, closure=None):
        armed, self._armed = self._armed, None
        self.table_l2_value = None
        if closure is not None or not all(self._plain(g) for g in self.param_groups):
            if armed:
                for group in self.param_groups:
                    ps = [p for p in group["params"] if p.grad is not None]
                    v = self._l2_by_hand(ps, [p.grad for p in ps], armed)
                    if v is not None:
                        self.table_l2_value = v.reshape(1) if self.table_l2_value is None else self.table_l2_value + v
            return super().step(closure)
        self._cuda_graph_capture_health_check()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            has_complex = self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            beta1, beta2 = group["betas"]
            big = [i for i, p in enumerate(params) if id(p) in self._table_ids]
            ok = len(big) > 0 and not has_complex and all(
                params[i].is_cuda and params[i].dtype == torch.float32 and grads[i].dtype == torch.float32 and
                params[i].is_contiguous() and grads[i].is_contiguous() for i in big)
            # the gradients must be views of ONE buffer (the gather's flat table gradients): their offsets from the
            # buffer start are then the same every step and the device tables are uploaded once
            ok = ok and len({grads[i].untyped_storage().data_ptr() for i in big}) == 1
            if not ok:
                big = []
            if big:
                base = min(grads[i].data_ptr() for i in big)
                l2 = tuple(armed.get(id(params[i]), 0.0) for i in big) if armed else None
                key = (tuple(params[i].data_ptr() for i in big), tuple(exp_avgs[i].data_ptr() for i in big),
                       tuple(grads[i].data_ptr() - base for i in big), l2)
                tab = self._cache.get(gi)
                if tab is None or tab[0] != key:
                    dev = params[big[0]].device
                    i64 = dict(dtype=torch.int64, device=dev)
                    tab = (key, torch.tensor(key[0], **i64), torch.tensor(key[1], **i64),
                           torch.tensor([exp_avg_sqs[i].data_ptr() for i in big], **i64),
                           torch.tensor([steps[i].data_ptr() for i in big], **i64),
                           torch.tensor([params[i].numel() for i in big], **i64),
                           torch.tensor([o // 4 for o in key[2]], **i64),
                           torch.tensor(l2, dtype=torch.float32, device=dev) if l2 is not None else None)
                    self._cache[gi] = tab
                torch._foreach_add_([steps[i] for i in big], 1)
                _, p_t, m_t, v_t, s_t, n_t, o_t, l2_t = tab
                dev = p_t.device
                ws = val = None
                if l2_t is not None:
                    ws = torch.empty(lib.xdfm_adam_tables_ws_elems(len(big)), dtype=torch.float32, device=dev)
                    val = torch.empty(1, dtype=torch.float32, device=dev)
                stream = torch.cuda.current_stream(dev).cuda_stream
                _lib.check(lib.xdfm_adam_tables(
                    p_t.data_ptr(), m_t.data_ptr(), v_t.data_ptr(), s_t.data_ptr(), n_t.data_ptr(), len(big), base,
                    o_t.data_ptr(), float(group["lr"]), float(beta1), float(beta2), float(group["eps"]),
                    l2_t.data_ptr() if l2_t is not None else None, ws.data_ptr() if ws is not None else None,
                    val.data_ptr() if val is not None else None, stream), "adam_tables")
                if val is not None:
                    self.table_l2_value = val if self.table_l2_value is None else self.table_l2_value + val
            bigset = set(big)
            small = [i for i in range(len(params)) if i not in bigset]
            if small:
                if armed:
                    v = self._l2_by_hand([params[i] for i in small], [grads[i] for i in small], armed)
                    if v is not None:
                        self.table_l2_value = v.reshape(1) if self.table_l2_value is None else self.table_l2_value + v
                _functional_adam([params[i] for i in small], [grads[i] for i in small], [exp_avgs[i] for i in small],
                                 [exp_avg_sqs[i] for i in small], [], [steps[i] for i in small],
                                 amsgrad=False, has_complex=has_complex, beta1=beta1, beta2=beta2, lr=group["lr"],
                                 weight_decay=0, eps=group["eps"], maximize=False, foreach=group["foreach"],
                                 capturable=True, differentiable=False, fused=True, grad_scale=None, found_inf=None,
                                 decoupled_weight_decay=False)
        return None
