from xdfm_amd.callbacks import Callback, CallbackList, EarlyStopping, History, ModelCheckpoint  # noqa: F401
