"""Drop-in import path of the reference's ``utils/config_utils.py``."""
from adt_str_amd.config_utils import deep_merge_dicts, load_config_from_yaml  # noqa: F401
