from xdfm_amd.models import xDeepFMAttention, xDeepFMAttentionV2  # noqa: F401
