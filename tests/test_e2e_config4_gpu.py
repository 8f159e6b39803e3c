"""BASELINE config[4] scaled down (tools/e2e.py): 2 000 synthetic one-shots -> CLAP curation on the GPU -> gold + bank ->
one epoch of the native loop over 512 note chunks with checkpoints, then resume from a mid-epoch checkpoint and end on
bitwise identical parameters.  Chain in the reference: augment_data_with_CLAP.py:71-193 -> copy_originals_to_augmented.py ->
convert_augmented_to_hdf5.py:69-141 -> train.py:253-328."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config4_end_to_end_scaled_down(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e
    out = e2e.main(["--workdir", str(tmp_path / "w"), "--shots", "2000", "--chunks", "512", "--batch-size", "16", "--tiny",
                    "--input-sec", "2.56", "--check-resume", "--keep"])
    assert out["steps"] == 512 // 16 and out["bank_shots"] == 2000 + 26 * 5                    # every pack file and every reference is in the bank
    assert len(out["checkpoints"]) == 3 and out["checkpoints"][-1] == "checkpoint-30"            # save_total_limit 3, every 10 steps
    assert np.isfinite(out["final_loss"]) and out["embeds_per_s"] > 0 and out["train_clips_per_s"] > 0
    assert out["resume"]["bitwise_identical"] and out["resume"]["steps"] == out["steps"]
    w = tmp_path / "w"
    assert (w / "outputs" / "e2e" / "model.safetensors").exists() and (w / "oneshot@16000.npz").exists()
    aug = w / "refs_clap_augmented"
    assert all((aug / str(p) / "gold").is_dir() for p in range(35, 61))
    n_copied = sum(len(files) for d, _, files in os.walk(aug) if os.path.basename(d) != "gold")
    assert n_copied == 2000                                                                      # each pack file copied exactly once
