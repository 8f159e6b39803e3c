"""Drop-in import path of the reference's ``utils/mapping_utils.py`` (hot-path tables only)."""
from adt_str_amd import mapping as _m


class MappingUtils:
    def __init__(self):
        self.GM_standard_midi_to_Gm_custom_Mapping = dict(_m.GM_TO_CUSTOM)
        self.ADTOF_mapping = dict(_m.ADTOF_MAPPING)
        self.ADTOF_inverse_mapping = {k: list(v) for k, v in _m.ADTOF_INVERSE_MAPPING.items()}
        self.ADTOF_label_mapping = dict(_m.ADTOF_LABEL)
        self.ADTOF_label_to_midi_mapping = dict(_m.ADTOF_LABEL_TO_PITCH)
