"""Drop-in for the reference's ``data_modules/copy_originals_to_augmented.py``:
``python data_modules/copy_originals_to_augmented.py <config.yaml> [--overwrite]``.

Second step of the curation pipeline (DATASET_AUGMENTATION_PIPELINE.md): every ``<clap_config.reference_root>/<label>/`` tree
becomes ``<reference_root>_clap_augmented/<label>/gold`` -- the hand-labelled one-shots next to the bins that
``augment_data_with_CLAP.py`` filled.  An existing ``gold`` directory is skipped unless ``--overwrite`` (reference :62-80)."""
import argparse
import os
import sys

sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from adt_str_amd.config_utils import load_merged  # noqa: E402
from adt_str_amd.curation import copy_originals_to_gold  # noqa: E402
from config import ClapConfig  # noqa: E402


def main(argv=None) -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("config_path", type=str, help="Path to the config file")
    parser.add_argument("--overwrite", action="store_true", help="If set, overwrite existing 'gold' directories under each instrument label")
    args = parser.parse_args(argv)
    cfg = load_merged(args.config_path)
    section = dict(cfg["clap_config"])
    section.update(cfg["shared"])
    c = ClapConfig(**section)
    if not os.path.isdir(c.reference_root):
        raise FileNotFoundError(f"reference_root does not exist: {c.reference_root}")
    copied, skipped = copy_originals_to_gold(c.reference_root, f"{c.reference_root}_clap_augmented", overwrite=args.overwrite)
    print(f"Finished. Copied: {copied}, Skipped: {skipped}")


if __name__ == "__main__":
    main()
