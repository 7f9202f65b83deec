from xdfm_amd.pro import *  # noqa: F401,F403  (deepctr/xdeepfm_pro/sfg_decoder.py of the reference)
