"""Oracle (test infrastructure): synthetic one-shot bank in the reference's HDF5
tree shape ``/<gm_custom_pitch>/<similarity_group>/<name>`` -> 1-D float32
(``data_modules/convert_augmented_to_hdf5.py:69-141``; consumed by
``modules/synthetiser.py:171-202,273-284``).

The bank is generated procedurally from a seed with numpy's PCG64 (bit-stable
across machines) so fixtures only have to store the seed.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# similarity groups in the order tolerance_thr_to_h5_group walks them (synthetiser.py:172-184)
GROUPS = ["gold", "100-90", "90-80", "80-70", "70-60", "60-50", "50-40", "40-30", "30-20", "20-10", "10-0"]


@dataclass
class TreeBank:
    tree: dict          # {str(pitch): {group: {name: float32[len]}}}
    sample_rate: int

    def as_tree(self) -> dict:
        return self.tree


def synthetic_bank(seed: int, sample_rate: int, pitches=range(35, 62), groups=("gold", "100-90", "90-80"),
                   shots_per_group: int = 2, min_len: int = 600, max_len: int = 2400,
                   sparse: bool = True) -> TreeBank:
    """Decaying noise + sine one-shots, peak-normalised to 1.0 like the
    reference's converter (convert_augmented_to_hdf5.py:101-103).  With
    ``sparse`` some (pitch, group) cells are left empty so the
    "valid_groups" filter (synthetiser.py:196) is exercised."""
    rng = np.random.default_rng(seed)
    tree: dict = {}
    for p in pitches:
        tree[str(p)] = {}
        for gi, g in enumerate(groups):
            if sparse and gi > 0 and rng.random() < 0.25:
                continue
            cell = {}
            for s in range(shots_per_group):
                n = int(rng.integers(min_len, max_len))
                t = np.arange(n) / sample_rate
                decay = rng.uniform(15.0, 80.0)
                f = rng.uniform(50.0, 5000.0)
                x = np.exp(-decay * t) * (0.6 * rng.standard_normal(n) + np.sin(2 * np.pi * f * t))
                x = (x / np.abs(x).max()).astype(np.float32)
                cell[f"shot_{p}_{gi}_{s}.wav"] = x
            tree[str(p)][g] = cell
    return TreeBank(tree=tree, sample_rate=sample_rate)
