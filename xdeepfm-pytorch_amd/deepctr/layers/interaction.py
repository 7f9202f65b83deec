from xdfm_amd.layers import CIN  # noqa: F401
