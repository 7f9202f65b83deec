"""torch.optim.Adam whose step sends the large tensors (the embedding / linear tables: 44 M of the 45 M
parameters at BASELINE config 2) through one streaming launch of the library (K7, `xdfm_adam_tables`) and the
small ones through ATen's fused kernel.  State layout (`step`, `exp_avg`, `exp_avg_sq` per parameter, device-
resident fp32 step counters) and hyper-parameters are torch's, so `state_dict()` / `load_state_dict()` and
code that edits `param_groups` keep working; anything the kernel does not implement (amsgrad, weight decay,
maximize, tensor learning rates, non-fp32 / non-CUDA parameters) falls back to `torch.optim.Adam.step`.
The update rule is the one basemodel.py:452 selects (torch.optim.Adam with default arguments)."""
import torch
from torch.optim.adam import adam as _functional_adam

from . import _lib

BIG = 1 << 16          # tensors with at least this many elements take the streaming kernel


class TableAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, fused=True, capturable=True)
        self._tables = {}          # group index -> cached device pointer tables

    def _plain(self, group):
        return (not group["amsgrad"] and group["weight_decay"] == 0 and not group["maximize"] and
                not group["differentiable"] and not group.get("decoupled_weight_decay", False) and
                isinstance(group["lr"], float) and all(isinstance(b, float) for b in group["betas"]) and
                getattr(self, "grad_scale", None) is None and getattr(self, "found_inf", None) is None)

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or not all(self._plain(g) for g in self.param_groups):
            return super().step(closure)
        self._cuda_graph_capture_health_check()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            has_complex = self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            beta1, beta2 = group["betas"]
            big = [i for i, (p, g) in enumerate(zip(params, grads))
                   if p.numel() >= BIG and p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and
                   p.is_contiguous() and g.is_contiguous() and not has_complex]
            if big:
                # only gradients that are views of ONE buffer (the gather's flat table gradients): their offsets
                # from the buffer start are the same every step, so the device tables are uploaded once
                stor = {}
                for i in big:
                    stor.setdefault(grads[i].untyped_storage().data_ptr(), []).append(i)
                big = max(stor.values(), key=lambda idx: sum(params[i].numel() for i in idx))
                if len(big) < 2:
                    big = []
            if big:
                base = min(grads[i].data_ptr() for i in big)
                key = (tuple(params[i].data_ptr() for i in big), tuple(exp_avgs[i].data_ptr() for i in big),
                       tuple(grads[i].data_ptr() - base for i in big))
                tab = self._tables.get(gi)
                if tab is None or tab[0] != key:
                    dev = params[big[0]].device
                    i64 = dict(dtype=torch.int64, device=dev)
                    tab = (key, torch.tensor(key[0], **i64), torch.tensor(key[1], **i64),
                           torch.tensor([exp_avg_sqs[i].data_ptr() for i in big], **i64),
                           torch.tensor([steps[i].data_ptr() for i in big], **i64),
                           torch.tensor([params[i].numel() for i in big], **i64),
                           torch.tensor([o // 4 for o in key[2]], **i64))
                    self._tables[gi] = tab
                torch._foreach_add_([steps[i] for i in big], 1)
                _, p_t, m_t, v_t, s_t, n_t, o_t = tab
                stream = torch.cuda.current_stream(p_t.device).cuda_stream
                _lib.check(lib.xdfm_adam_tables(p_t.data_ptr(), m_t.data_ptr(), v_t.data_ptr(), s_t.data_ptr(),
                                                n_t.data_ptr(), len(big), base, o_t.data_ptr(), float(group["lr"]),
                                                float(beta1), float(beta2), float(group["eps"]), stream), "adam_tables")
            bigset = set(big)
            small = [i for i in range(len(params)) if i not in bigset]
            if small:
                _functional_adam([params[i] for i in small], [grads[i] for i in small], [exp_avgs[i] for i in small],
                                 [exp_avg_sqs[i] for i in small], [], [steps[i] for i in small],
                                 amsgrad=False, has_complex=has_complex, beta1=beta1, beta2=beta2, lr=group["lr"],
                                 weight_decay=0, eps=group["eps"], maximize=False, foreach=group["foreach"],
                                 capturable=True, differentiable=False, fused=True, grad_scale=None, found_inf=None,
                                 decoupled_weight_decay=False)
        return None
