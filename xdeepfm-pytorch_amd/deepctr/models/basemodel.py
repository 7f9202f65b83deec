from xdfm_amd.models import BaseModel, Linear  # noqa: F401
