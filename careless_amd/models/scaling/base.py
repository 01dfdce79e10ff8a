from careless_amd.models.base import BaseModel


class Scaler(BaseModel):
    """Base class for scaling models (reference `careless/models/scaling/base.py`)."""
    trainable = True
