from xdfm_amd.models import BaseModel, Linear, xDeepFM, xDeepFMAttention, xDeepFMAttentionV2  # noqa: F401
from . import basemodel, xdeepfm, xdeepfm_attn  # noqa: F401
