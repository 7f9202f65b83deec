from xdfm_amd.pro import *  # noqa: F401,F403  (deepctr/xdeepfm_pro/xdeepfm_pro.py of the reference)
