from xdfm_amd.layers import (AttentionPooling, CINAttention, CINAttentionV2,  # noqa: F401
                             MultiHeadSelfAttention)
