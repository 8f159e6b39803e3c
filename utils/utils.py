"""Drop-in import path of the hot-path helpers of the reference's ``utils/utils.py``."""
from adt_str_amd.masks import (_causal_mask, _key_padding_mask_from_lengths, create_mask_plain,  # noqa: F401
                               select_inference_device)
