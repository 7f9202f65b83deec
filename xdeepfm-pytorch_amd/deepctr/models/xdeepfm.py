from xdfm_amd.models import xDeepFM  # noqa: F401
