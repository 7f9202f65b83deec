"""xdfm_amd: the xDeepFM hot path (embedding gather -> CIN [-> attention pooling], forward and
backward) on hand-written gfx950 kernels, behind the API of Syclus123/xDeepFM-pytorch.

    xdfm_amd._lib      ctypes binding of libxdfm_hip.so (C ABI: include/xdfm.h)
    xdfm_amd.ops       autograd wrappers (EmbedGather, CINStack)
    xdfm_amd.layers    CIN / CINAttention / CINAttentionV2 / DNN / PredictionLayer
    xdfm_amd.models    xDeepFM / xDeepFMAttention / xDeepFMAttentionV2 (compile / fit / predict)
    xdfm_amd.dist      row-parallel data parallelism over RCCL

The sibling package `deepctr` re-exports these under the reference's module paths.
"""
__version__ = "0.1.0"
