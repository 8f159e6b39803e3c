"""Oracle (test infrastructure) for the CLAP audio path (reference ``modules/clap_encoder.py:8-54``).

The arithmetic of this row lives in a third-party package, ``transformers`` (unpinned in the reference's
``requirements.txt:12``; 5.15.0 in this image): ``ClapFeatureExtractor`` (numpy float64 STFT / mel / dB)
and ``ClapAudioModel`` (HTSAT) + ``ClapModel.audio_projection``.  ``transformers`` is installed both here and
on the GPU box, so the checker is those very classes run on the CPU -- the code the reference itself calls
(``clap_encoder.py:11,18,22-23,45-54``).  Pretrained ``laion/clap-htsat-fused`` weights are not available
offline: parity is pinned for randomly initialised weights with the class-default extractor settings
("parity unpinned" for the real checkpoint).  ``is_longer`` is an explicit input (the extractor flips a random
entry when no clip exceeds 10 s, feature_extraction_clap.py:347-350).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch


def feature_extractor():
    from transformers import ClapFeatureExtractor
    return ClapFeatureExtractor()


def logmel_db(clips: Sequence[np.ndarray]) -> np.ndarray:
    """[B, 1001, 64] float32: the (identical) mel of the 4 fusion channels for clips no longer than 10 s."""
    fe = feature_extractor()
    out = []
    for c in clips:
        mel, longer = fe._get_input_mel(np.asarray(c, dtype=np.float64), fe.nb_max_samples, "fusion", "repeatpad")
        assert not longer
        out.append(mel[0])
    return np.stack(out).astype(np.float32)


def features(clips: Sequence[np.ndarray]):
    """The whole ``ClapFeatureExtractor.__call__`` (fusion truncation, repeatpad): (input_features [B, 4, 1001, 64] float32,
    is_longer [B, 1] bool).  Consumes numpy's global RNG like the reference (crops of long clips, the forced flag)."""
    out = feature_extractor()([np.asarray(c, dtype=np.float64) for c in clips], sampling_rate=48000, return_tensors="np")
    return np.asarray(out["input_features"], dtype=np.float32), np.asarray(out["is_longer"], dtype=bool)


def random_clap_model(seed: int = 0):
    """ClapModel with the fused-HTSAT audio tower of ``laion/clap-htsat-fused`` (enable_fusion, aff_2d), random weights."""
    from transformers import ClapConfig, ClapModel
    torch.manual_seed(seed)
    cfg = ClapConfig(audio_config={"enable_fusion": True, "fusion_type": "aff_2d"})
    model = ClapModel(cfg).eval()
    with torch.no_grad():          # non-trivial BatchNorm statistics so the eval-mode affine is exercised
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0.0, 0.5)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.normal_(1.0, 0.1)
                m.bias.normal_(0.0, 0.1)
    return model


@torch.no_grad()
def audio_embeddings(model, input_features: torch.Tensor, is_longer: torch.Tensor) -> dict:
    """``ClapWrapper._get_audio_features`` (clap_encoder.py:30-54) on the CPU."""
    out = model.audio_model(input_features=input_features, is_longer=is_longer, return_dict=True)
    proj = model.audio_projection(out.pooler_output)
    return {"pooled": out.pooler_output, "embedding": proj / proj.norm(p=2, dim=-1, keepdim=True)}
