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


def test_config4_end_to_end_two_ranks(tmp_path):
    """The multi-GPU half of config[4] at two ranks (one-GPU box: both on GPU 0 over gloo, ADT_SHARE_GPU=1; on a node: one GPU each
    and RCCL): every rank embeds its stride of the files and the embeddings are all-gathered, rank 0 assigns, copies and
    builds the bank, both ranks train the epoch with the per-segment gradient all-reduce, checkpoint, and the resumed run ends on
    bitwise the same parameters."""
    import json
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), ADT_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "e2e.py"), "--workdir", str(tmp_path / "w"), "--shots", "601",
                        "--chunks", "256", "--batch-size", "16", "--tiny", "--input-sec", "2.56", "--check-resume", "--keep"],
                       capture_output=True, text=True, env=env, timeout=900)
    if r.returncode != 0:
        print(r.stdout[-3000:])
        print(r.stderr[-12000:])
    assert r.returncode == 0, "torchrun tools/e2e.py failed (output above)"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                                                     # rank 0 reports
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 256 // (16 * 2) and out["bank_shots"] == 601 + 26 * 5
    assert out["resume"]["bitwise_identical"] and np.isfinite(out["final_loss"])
    aug = tmp_path / "w" / "refs_clap_augmented"
    n_copied = sum(len(files) for d, _, files in os.walk(aug) if os.path.basename(d) != "gold")
    assert n_copied == 601                                                                       # odd count: the ranks' shards differ by one
