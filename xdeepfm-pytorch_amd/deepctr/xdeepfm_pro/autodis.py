from xdfm_amd.pro import *  # noqa: F401,F403  (deepctr/xdeepfm_pro/autodis.py of the reference)
