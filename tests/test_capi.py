"""CPU: the C-ABI library loads and exports exactly what include/xdfm.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from conftest import PKG, ROOT

HEADER = os.path.join(ROOT, "include", "xdfm.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(xdfm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from xdfm_amd import _lib
    lib = _lib.load()
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libxdfm_hip.so lacks %s" % n
    assert sorted(_lib.SIGNATURES.keys()) == names, "python binding and header disagree"


def test_abi_version_and_error_text():
    from xdfm_amd import _lib
    lib = _lib.load()
    assert lib.xdfm_abi_version() == _lib.ABI_VERSION
    assert isinstance(lib.xdfm_last_error(), bytes)
    # argument validation happens before any device work, so it can be exercised without a GPU
    rc = lib.xdfm_cin_level_fwd(None, None, None, None, 8, 4, 4, 64, 1, None, None)
    assert rc == 1 and b"null pointer" in lib.xdfm_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "cin_level_fwd")
    assert lib.xdfm_set_option(b"no_such_key", 1) == 1
    assert lib.xdfm_set_option(b"fwd_nf", 2) == 0 and lib.xdfm_get_option(b"fwd_nf") == 2
    assert lib.xdfm_set_option(b"fwd_nf", 1) == 0


def test_workspace_size_queries():
    from xdfm_amd import _lib
    lib = _lib.load()
    old = _lib.get_option("cin_math")
    try:
        _lib.set_option("cin_math", 0)            # fp32-MFMA kernels
        # forward pack: MB * (Tpad + 4) * 64 * MT floats
        assert lib.xdfm_cin_fwd_pack_elems(128, 128, 26) == 1 * (128 * 13 + 4) * 64 * 4
        assert lib.xdfm_cin_fwd_pack_elems(256, 26, 26) == 1 * (340 + 4) * 64 * 8
        assert lib.xdfm_cin_fwd_pack_elems(512, 26, 22) == 2 * (288 + 4) * 64 * 8
        assert lib.xdfm_cin_bwd_pack_elems(128, 128, 26) == (4 * 26 * 16 + 2 * 16 + 3) * 256
        assert lib.xdfm_cin_bwd_pack_elems(300, 4, 4) == 0          # > 256 rows per call is rejected
        # one dWt copy per n-split: (L2 of config 2: 7 workgroups per split -> 71 splits of 928 columns)
        assert lib.xdfm_cin_bwd_w_ws_elems(128, 64, 26, 65536) == 26 * 128 * 64 * 71
        assert lib.xdfm_cin_bwd_w_ws_elems(6, 5, 3, 40) == 3 * 32 * 32 * 2
        _lib.set_option("cin_math", 1)            # f16x3 kernels: 128-float header + 2 KB (hi + lo) per (step, row tile)
        assert lib.xdfm_cin_fwd_pack_elems(128, 128, 26) == 128 + 1 * (16 * 13 + 2) * 4 * 512    # 104 stages of 2 steps + a spare stage
        # ragged last block: 4 steps; whole ring stages (1 step at 8 row tiles) + a spare one.  Hp == m: the level can be
        # level 0 (x_prev is x0), whose folded pack (182 pairs i <= j per lane half = 23 steps, + a spare stage) follows
        assert lib.xdfm_cin_fwd_pack_elems(256, 26, 26) == 128 + 1 * (3 * 13 + 4 + 1) * 8 * 512 + 128 + (23 + 1) * 8 * 512
        assert lib.xdfm_cin_fwd_pack_elems(512, 26, 22) == 128 + 2 * (3 * 11 + 3 + 1) * 8 * 512
        assert lib.xdfm_cin_fwd_pack_elems(128, 128, 7) == 1 * (128 * 4 + 4) * 64 * 4             # odd m: fp32 kernel
        assert lib.xdfm_cin_bwd_pack_elems(128, 128, 26) == 128 + (4 * 26 * 8 + 2 * 8) * 512
        assert lib.xdfm_cin_bwd_pack_elems(300, 4, 4) == 0
        # row scales (one 256-float header per n-split: xdfm_cin_bwd_prep finds them per split) + partial row maxima of the
        # stand-alone passes (4 blocks per row) + fp16 hi/lo planes of dOut + the slabs
        # (8-wave workgroups: 4 workgroups per n-split -> 64 splits; the fp32 kernels' 71 slabs would need less)
        assert lib.xdfm_cin_bwd_w_ws_elems(128, 64, 26, 65536) == max(64 * 256 + 896 + 128 * 65536 + 26 * 128 * 64 * 64, 26 * 128 * 64 * 71)
        # dbias partials of the fused dOut pass: one per row and block of 4 splits (4096 columns) -- no more than cin_dout's own
        assert lib.xdfm_cin_bwd_prep_ws_elems(128, 64, 26, 4096, 16) == lib.xdfm_cin_dout_ws_elems(128, 4096, 16) == 128 * 64
    finally:
        _lib.set_option("cin_math", old)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from xdfm_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.XdfmError):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    import torch
    from xdfm_amd import ops
    from deepctr.layers import CIN
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        CIN(3, (4,))(torch.zeros(2, 3, 4))
    plan = ops.EmbedPlan([0], [4], [], 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.EmbedGather.apply(torch.zeros(2, 1), None, plan, False, torch.zeros(4, 4))


def test_product_never_imports_oracle():
    """The shipped package must not reference oracle/ (it is test infrastructure)."""
    for base, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(base, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(base, f)
