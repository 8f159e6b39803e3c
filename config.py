"""Drop-in for the reference's ``config.py``: the names its entry points import."""
from dataclasses import dataclass

from adt_str_amd.network import ADTModelConfig  # noqa: F401
from adt_str_amd.synth import SynthDrumConfig  # noqa: F401


@dataclass
class SharedConfig:
    input_sec: float
    time_res: float
    win_length: int
    sample_rate: int


@dataclass
class ClapConfig(SharedConfig):
    """``clap_config`` section + the shared keys (reference config.py:17-21)."""
    model_name: str
    batch_size: int
    sample_pack_root: str
    reference_root: str
