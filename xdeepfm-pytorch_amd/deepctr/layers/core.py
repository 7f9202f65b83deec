from xdfm_amd.layers import DNN, PredictionLayer  # noqa: F401
