from xdfm_amd.layers import (CIN, DNN, AttentionPooling, CINAttention, CINAttentionV2,  # noqa: F401
                             MultiHeadSelfAttention, PredictionLayer)
from . import cin_attention, core, interaction  # noqa: F401
