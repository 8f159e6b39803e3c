"""Drop-in for the reference's ``model.py``: same class names, constructor and call signatures,
state-dict keys; the arithmetic runs on hand-written gfx950 kernels (see adt_str_amd/network.py)."""
from adt_str_amd.frontend import ComputeMelSpectrogram  # noqa: F401
from adt_str_amd.network import (ADTModel, ADTModelConfig, Decoder, Encoder, PositionalEncoding,  # noqa: F401
                                 TokenEmbedding_plain)
