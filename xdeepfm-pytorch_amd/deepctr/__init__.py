"""Drop-in module paths of the reference (`deepctr.inputs`, `deepctr.models`, `deepctr.layers`,
`deepctr.callbacks`) re-exporting the MI355X implementation in `xdfm_amd`.  Unlike the
reference's package __init__ (deepctr/__init__.py:3-6) nothing is fetched or spawned on import."""
from . import inputs, layers, models, callbacks  # noqa: F401

__version__ = "0.2.9+xdfm_amd"
