"""ctypes binding of libxdfm_hip.so (the C ABI declared in include/xdfm.h).

There is deliberately no fallback: if the shared library is missing or lacks a symbol the
import of any op fails with an explicit error, and every op refuses non-CUDA tensors.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

# torch must be imported BEFORE the library is dlopen'ed: the torch wheel bundles its own
# libamdhip64 and the process must end up with ONE HIP runtime (the one that owns torch's device
# context and streams).  Loaded first, torch's copy satisfies our DT_NEEDED by soname.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XDFM_LIB", os.path.join(_HERE, "libxdfm_hip.so"))

P = c_void_p
# name -> (restype, argtypes); mirrors include/xdfm.h one to one
SIGNATURES = {
    "xdfm_abi_version": (c_int, []),
    "xdfm_last_error": (c_char_p, []),
    "xdfm_device_count": (c_int, []),
    "xdfm_set_option": (c_int, [c_char_p, c_int]),
    "xdfm_get_option": (c_int, [c_char_p]),
    "xdfm_graph_node_census": (c_int, [P, P, P, P]),
    "xdfm_set_ticket_board": (c_int, [P, c_int]),
    "xdfm_embed_gather_fwd": (c_int, [P, c_long, c_int, P, P, P, P, c_int, c_int, P, P, c_int, P, P, P, P, P]),
    "xdfm_embed_scatter_bwd": (c_int, [P, c_long, c_int, P, P, c_int, c_int, P, c_int, P, P, P, P, P, P, P, P]),
    "xdfm_embed_scatter_bwd_marked": (c_int, [P, c_long, c_int, P, P, c_int, c_int, P, c_int, P, P, c_long, P, c_long,
                                              P, P, P, P, P, P]),
    "xdfm_cin_fwd_pack_elems": (c_size_t, [c_int, c_int, c_int]),
    "xdfm_cin_fwd_pack": (c_int, [P, c_int, c_int, c_int, P, P]),
    "xdfm_cin_pack_all_supported": (c_int, [c_int, c_int, c_int]),
    "xdfm_cin_pack_all": (c_int, [P, c_int, P]),
    "xdfm_cin_level_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_long, c_int, P, P]),
    "xdfm_cin_direct_sum": (c_int, [P, c_int, c_int, c_int, c_int, P, c_long, c_int, P]),
    "xdfm_cin_dout": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, c_int, P, c_int, c_long, c_int, c_int,
                              c_int, P, P, P]),
    "xdfm_cin_dout_ws_elems": (c_size_t, [c_int, c_int, c_int]),
    "xdfm_cin_dout_det": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, c_int, P, c_int, c_long, c_int, c_int,
                                  c_int, P, P, P, P]),
    "xdfm_cin_bwd_prep_ws_elems": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "xdfm_cin_bwd_prep": (c_int, [P, P, c_long, c_int, c_int, c_int, c_int, P, c_int, c_int, P, c_int, c_long, c_int, c_int, c_int, P, P, P,
                                  P, P, c_int, c_int, P, P, P]),
    "xdfm_cin_level_fwd_ex_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "xdfm_cin_level_fwd_ex": (c_int, [P, P, P, P, c_int, c_int, c_int, c_long, c_int, P, c_int, P, c_long, c_int, c_int, c_int,
                                      P, c_long, P]),
    "xdfm_cin_bwd_nodout_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "xdfm_cin_level_bwd_x_src": (c_int, [P, c_long, P, c_int, P, c_int, c_long, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, c_int,
                                         c_int, c_long, P, P, c_int, P]),
    "xdfm_cin_level_bwd_w_prepared": (c_int, [P, P, P, c_int, c_int, c_int, c_long, P, P, P]),
    "xdfm_cin_bwd_pack_elems": (c_size_t, [c_int, c_int, c_int]),
    "xdfm_cin_bwd_pack": (c_int, [P, c_int, c_int, c_int, P, P]),
    "xdfm_cin_level_bwd_x": (c_int, [P, P, P, P, c_int, c_int, c_int, c_long, P, P, P]),
    "xdfm_cin_level_bwd_x_ex": (c_int, [P, P, P, P, c_int, c_int, c_int, c_long, P, P, c_int, P]),
    "xdfm_cin_bwd_x_is_folded": (c_int, [c_int, c_int, c_int, c_int]),
    "xdfm_cin_bwd_w_ws_elems": (c_size_t, [c_int, c_int, c_int, c_long]),
    "xdfm_cin_level_bwd_w": (c_int, [P, P, P, c_int, c_int, c_int, c_long, P, P, P]),
    "xdfm_cin_attn_theta_elems": (c_size_t, [c_int, c_int, c_int]),
    "xdfm_cin_attn_pool_fwd": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, c_float, P, P]),
    "xdfm_cin_attn_pool_bwd": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, c_float,
                                       P, P]),
    "xdfm_cin_attn_pool_bwd_ws_elems": (c_size_t, [c_int, c_int, c_int, c_int]),
    "xdfm_cin_attn_pool_bwd_det": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, c_float,
                                           P, P]),
    "xdfm_cin_attn_dropout_mask": (c_int, [c_int, c_int, c_int, c_int, c_float, P, P, P]),
    "xdfm_head_ws_elems": (c_size_t, [c_int, c_int]),
    "xdfm_head_fwd": (c_int, [P, P, P, c_int, P, P, c_int, P, P, c_int, P, P, P, P]),
    "xdfm_head_bwd": (c_int, [P, P, P, P, P, c_int, P, P, c_int, c_int, P, P, P, P, P, P]),
    "xdfm_adam_step_ws_elems": (c_size_t, [c_int]),
    "xdfm_adam_step": (c_int, [P, c_int, c_double, c_double, c_double, c_double, P, P, P]),
    "xdfm_adam_step_lr": (c_int, [P, c_int, c_double, P, c_double, c_double, c_double, P, P, P]),
    "xdfm_adam_step_deferred": (c_int, [P, c_int, P, c_double, P, c_double, c_double, c_double, P, P, P]),
    "xdfm_adam_catchup_rows": (c_int, [P, c_long, c_int, P, P, c_int, c_int, P, P, P, c_double, c_double, c_double, P, P]),
    "xdfm_adam_apply_rows": (c_int, [P, c_long, c_int, P, P, c_int, c_int, P, P, P, c_double, c_double, c_double, P, P, P]),
    "xdfm_adam_flush": (c_int, [P, c_int, P, c_double, c_double, c_double, P, P]),
    "xdfm_adam_selftest": (c_int, [c_int, ctypes.c_ulonglong, ctypes.c_ulonglong, c_double, c_double, c_double, c_double, P, P]),
    "xdfm_vocab_lse_update": (c_int, [P, c_long, c_int, c_int, P, P, P]),
    "xdfm_vocab_softmax_grad": (c_int, [P, c_long, c_int, c_int, P, P, P]),
    "xdfm_vocab_ce_x3_supported": (c_int, [c_int]),
    "xdfm_vocab_ce_pack_elems": (c_long, [c_int, c_int]),
    "xdfm_vocab_ce_rows_padded": (c_long, [c_int]),
    "xdfm_vocab_ce_plan": (c_long, [c_int, P, c_int, c_int, P, P, c_long, P, P]),
    "xdfm_vocab_ce_pack_hidden": (c_int, [P, c_long, c_int, c_int, P, P]),
    "xdfm_vocab_ce_fwd": (c_int, [P, P, c_long, c_int, c_int, P, c_int, P, c_long, P, P, P, P, P, P]),
    "xdfm_vocab_ce_pack_g": (c_int, [P, c_int, c_int, P, P]),
    "xdfm_vocab_ce_bwd_h": (c_int, [P, c_int, c_int, P, c_int, P, c_long, P, P, P, P, P, P, P, c_long, P]),
    "xdfm_vocab_ce_bwd_w": (c_int, [P, c_int, c_int, P, c_int, c_int, P, P, P, P, P]),
    "xdfm_colsum_ws_elems": (c_size_t, [c_int]),
    "xdfm_colsum": (c_int, [P, c_long, c_int, c_long, P, P, P]),
    "xdfm_relu_bwd_colsum": (c_int, [P, P, c_long, c_int, c_long, c_long, P, P, P, P]),
    "xdfm_l2_reg_fwd": (c_int, [P, P, P, c_int, P, P, P]),
    "xdfm_l2_reg_bwd": (c_int, [P, P, P, c_int, P, P, P, c_int, P]),
}

class PackJob(ctypes.Structure):
    """xdfm_cin_pack_job of include/xdfm.h"""
    _fields_ = [("W", c_void_p), ("H", c_int), ("Hp", c_int), ("m", c_int), ("fwd_pack", c_void_p), ("bwd_pack", c_void_p)]


class AdamTensor(ctypes.Structure):
    """xdfm_adam_tensor of include/xdfm.h"""
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
                ("step", c_void_p), ("numel", c_long), ("l2", ctypes.c_float), ("grad_marks", c_void_p), ("flags", c_int),
                ("last", c_void_p)]


class AdamClock(ctypes.Structure):
    """xdfm_adam_clock of include/xdfm.h"""
    _fields_ = [("clock", c_void_p), ("consts", c_void_p), ("cap", c_int)]


class AdamRows(ctypes.Structure):
    """xdfm_adam_rows of include/xdfm.h (device pointer tables of one gather's fields)"""
    _fields_ = [("param", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("last", c_void_p), ("l2", c_void_p),
                ("grad", c_void_p), ("marks", c_void_p)]


ABI_VERSION = 8
_lib = None


class XdfmError(RuntimeError):
    pass


def load():
    """dlopen the library once, bind every symbol of include/xdfm.h, check the ABI version."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XdfmError(
            "libxdfm_hip.so not found at %s -- build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback for the xDeepFM hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise XdfmError("libxdfm_hip.so at %s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if lib.xdfm_abi_version() != ABI_VERSION:
        raise XdfmError("libxdfm_hip.so ABI %d != binding ABI %d" % (lib.xdfm_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().xdfm_last_error().decode("utf-8", "replace")
        if rc == 1:
            raise ValueError("xdfm %s: %s" % (what, msg))
        raise XdfmError("xdfm %s failed (code %d): %s" % (what, rc, msg))


_BOARDS = {}


def ticket_board(device):
    """OPT-IN (XDFM_TICKETS=1): register (once per device) the zeroed ticket array that lets the kernels' last block do the
    work of the separate "finish" launches (include/xdfm.h, xdfm_set_ticket_board).  Measured on MI355X (round 3): the
    captured step shrinks from 48 to 40 nodes and gets SLOWER, 2.05 against 1.97 ms -- a block that draws a ticket first
    waits for its own stores, the device-scope atomic is a ~2 us round trip, and the finishing block's acquire invalidates
    its XCD's L2 under the blocks still running (the fused dOut pass: 44 -> 83 us).  Off by default; the tests run both."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _BOARDS or os.environ.get("XDFM_TICKETS", "0") != "1":
        return
    if torch.cuda.is_current_stream_capturing():
        return                                 # never allocate inside a capture; the eager steps before it come here first
    with torch.cuda.device(idx):
        board = torch.zeros(2048, dtype=torch.int32, device=device)
        check(load().xdfm_set_ticket_board(c_void_p(board.data_ptr()), 2048), "set_ticket_board")
    _BOARDS[idx] = board


def set_option(key, value):
    check(load().xdfm_set_option(key.encode(), int(value)), "set_option")


def get_option(key):
    return load().xdfm_get_option(key.encode())
