"""Replay of the train step from a captured HIP graph.

The step of `BaseModel.train_on_batch` is ~230 kernel launches issued from Python (forward, loss, L2,
backward on the autograd thread, fused Adam); once the CIN runs in f16x3 arithmetic the GPU finishes them
faster than the host can enqueue them.  After two eager steps on a given batch shape the step is captured
once (`torch.cuda.CUDAGraph`: all library launches go to the capturing stream, nothing in the step
synchronises or allocates outside torch's caching allocator) and every later step of that shape is two
device-to-device copies into the static input buffers plus one graph launch.

Guards:
  * the captured graph is inspected (`xdfm_graph_node_census`): it must contain no memset node -- on this
    stack (ROCm 7.2, gfx950) a memset node is not ordered against its neighbouring kernel nodes
    (tools/graph_memset_probe.py: ATen's sum(0) inside a graph is wrong in 3 of 4 replays).  The library
    itself issues no memset on this path and `ops.Dense` keeps ATen's semaphore memset out of the step;
  * the graph is keyed on the batch shape and on a signature of everything that is baked into it
    (parameter storage, requires_grad pattern, optimizer hyper-parameters, loss function, train flags); a
    change re-captures;
  * anything unexpected (capture error, unsupported optimizer, row-parallel run, profiling hooks)
    falls back to the eager step, permanently for that model after a capture error.
`XDFM_HIP_GRAPH=0` disables the replay.
"""
import ctypes
import gc
import os
import warnings
import weakref

import torch

from . import _lib, ops
from . import dist as xdist

EAGER_STEPS_BEFORE_CAPTURE = 2
MAX_GRAPHS = 6


def census(graph: "torch.cuda.CUDAGraph"):
    """(nodes, memset nodes, unexpected nodes) of a graph captured with keep_graph=True."""
    n, ms, other = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(_lib.load().xdfm_graph_node_census(ctypes.c_void_p(graph.raw_cuda_graph()), ctypes.byref(n),
                                                   ctypes.byref(ms), ctypes.byref(other)), "graph_node_census")
    return n.value, ms.value, other.value


class _Entry(object):
    __slots__ = ("sig", "eager", "graph", "sx", "sy", "out", "nodes")

    def __init__(self, sig):
        self.sig, self.eager, self.graph, self.sx, self.sy, self.out, self.nodes = sig, 0, None, None, None, None, 0


class GraphedStep(object):
    def __init__(self, model):
        # weak: the model owns this object; a reference cycle would leave old graphs to the cycle collector,
        # which may then destroy one (hipGraphExecDestroy + pool release) in the middle of a later capture
        self._model = weakref.ref(model)
        self.entries = {}
        self.disabled = os.environ.get("XDFM_HIP_GRAPH", "1") == "0"
        self.replays = 0
        self.stream = None      # the eager steps that precede a capture and the capture share one side stream

    @property
    def model(self):
        return self._model()

    # -- what the captured graph depends on besides the batch ---------------------------------------
    def _signature(self):
        m = self.model
        opt = m.optim
        # the learning rate is NOT baked in when the optimizer hands it to K7 through a device scalar (TableAdam.sync_lr)
        lr_free = hasattr(opt, "sync_lr")
        hyper = tuple(tuple((k, (v if isinstance(v, (int, float, bool, str, tuple, type(None))) else id(v)))
                            for k, v in sorted(pg.items()) if k != "params" and not (lr_free and k == "lr" and isinstance(v, float)))
                      for pg in opt.param_groups)
        ptrs, req = 0, 0
        for p in m.parameters():
            ptrs = (ptrs * 1000003 + p.data_ptr()) & 0xFFFFFFFFFFFFFFF
            req = (req << 1 | int(p.requires_grad)) & 0xFFFFFFFFFFFFFFF
        # the optimizer's state tensors are baked into the captured K7 launches by address: `generation` changes
        # whenever they may have been replaced (load_state_dict, unpickling, add_param_group)
        return (id(opt), getattr(opt, "generation", 0), hyper, ptrs, req, id(m.loss_func), m.training, id(m.aux_loss),
                _lib.get_option("cin_math"))

    def eligible(self, x, y):
        m = self.model
        if self.disabled or not (x.is_cuda and y.is_cuda) or ops.PROFILE is not None or not torch.is_grad_enabled() \
                or not getattr(m, "_optim_capturable", False):
            return False
        dp = xdist.current()
        if dp is None:
            return True
        # row-parallel: only the collective-free first half of the split step is captured; it needs equal shards
        # (static shapes on every rank) and the L2 term in the optimizer
        if os.environ.get("XDFM_HIP_GRAPH_DP", "1") == "0" or not m._can_split_step(dp, m._l2_fusion()):
            return False
        n = dp._n_global
        return n is not None and n % dp.world == 0 and x.shape[0] == n // dp.world

    def __call__(self, x, y):
        m = self.model
        if not self.eligible(x, y):
            return m._train_step_eager(x, y)
        if hasattr(m.optim, "sync_lr"):
            m.optim.sync_lr()              # outside any capture: the device scalar K7 reads follows param_groups["lr"]
        dp = xdist.current()
        key = (tuple(x.shape), tuple(y.shape), x.dtype, y.dtype, x.device.index, self._signature(),
               None if dp is None else dp.world)
        ent = self.entries.get(key)
        if ent is None:
            if len(self.entries) >= MAX_GRAPHS:             # each graph owns the activations of one step
                self.entries.pop(next(iter(self.entries)))
            ent = self.entries[key] = _Entry(key)
        if ent.graph is None:
            if ent.eager < EAGER_STEPS_BEFORE_CAPTURE:
                ent.eager += 1
                return self._eager_on_side_stream(x, y)
            if not self._capture(ent, x, y):
                return m._train_step_eager(x, y)
        ent.sx.copy_(x)
        ent.sy.copy_(y)
        if dp is None and hasattr(m.optim, "note_replay"):
            m.optim.note_replay()          # deferred table update: periodic flush, step count (the step's Python does not run)
        ent.graph.replay()
        self.replays += 1
        if dp is not None:          # captured: the first half; exchange, scatter, all-reduce and optimizer follow eagerly
            y_pred, loss, stash = ent.out
            return m._split_step_second(y_pred, loss, stash, m._l2_fusion())
        return ent.out

    def _eager_on_side_stream(self, x, y):
        """The warm-up steps run on the stream the capture will use (the pattern torch documents for whole-step
        capture): autograd then creates the parameters' AccumulateGrad nodes on that stream, never on the
        legacy default stream, which must not take part in a capture."""
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=x.device)
        cur = torch.cuda.current_stream(x.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            out = self.model._train_step_eager(x, y)
        cur.wait_stream(self.stream)
        for t in out:
            t.record_stream(cur)
        return out

    def _capture(self, ent, x, y):
        m = self.model
        gc.collect()                       # destroy unreachable graphs / pools now, not inside the capture
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            ent.sx, ent.sy = x.clone(), y.clone()
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph(keep_graph=True)
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=x.device)
            if xdist.current() is None:
                with torch.cuda.graph(g, stream=self.stream):
                    out = m._train_step_eager(ent.sx, ent.sy)
            else:
                # thread_local: the process group's watchdog thread polls events while we capture; it must not be
                # able to invalidate the capture (nothing it touches is part of it)
                with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
                    out = m._split_step_first(ent.sx, ent.sy)
            n, n_memset, n_other = census(g)
            if n_memset or n_other:
                raise RuntimeError("captured train step holds %d memset and %d unexpected nodes of %d"
                                   % (n_memset, n_other, n))
            g.instantiate()
            ent.graph, ent.out, ent.nodes = g, out, n
            return True
        except Exception as exc:      # noqa: BLE001 -- any failure means: keep training eagerly
            warnings.warn("xdfm: HIP-graph capture of the train step failed (%s); continuing with eager launches" % (exc,))
            self.disabled = True
            ent.graph = ent.sx = ent.sy = ent.out = None
            if m._plan is not None:
                m._plan.reg_defer = None
            return False
        finally:
            if gc_was_enabled:
                gc.enable()
