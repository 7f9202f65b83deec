"""deepctr.xdeepfm_pro of the reference (deepctr/xdeepfm_pro/__init__.py:34-42), served by xdfm_amd.pro."""
from xdfm_amd.pro import (AutoDisLayer, BaseModelSFG, DenseFeatureEncoder, LabelAwareAttention, SFGDecoder, SFGLoss,  # noqa: F401
                          xDeepFMPro, xDeepFMProLight)

__all__ = ['xDeepFMPro', 'xDeepFMProLight', 'SFGDecoder', 'SFGLoss', 'LabelAwareAttention', 'AutoDisLayer',
           'DenseFeatureEncoder', 'BaseModelSFG']
