from xdfm_amd.pro import *  # noqa: F401,F403  (deepctr/xdeepfm_pro/basemodel_sfg.py of the reference)
