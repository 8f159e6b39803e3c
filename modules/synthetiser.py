"""Drop-in import path of the reference's ``modules/synthetiser.py``."""
from adt_str_amd.synth import SynthDrum, SynthDrumConfig  # noqa: F401
