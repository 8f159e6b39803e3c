"""Drop-in import path of the reference's ``modules/clap_encoder.py`` (audio tower only)."""
from adt_str_amd.clap_encoder import ClapWrapper  # noqa: F401
