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


# ---------------------------------------------------------------------------------------------------------------- golden set G7
# Inputs of tests/golden/clap.npz (tools/make_golden.py:g7_clap runs the reference's own ClapWrapper on them).  The model's 28 M weights
# cannot travel in a < 1 MB fixture: they are regenerated from GOLDEN_MODEL_SEED by random_clap_model and checked by weights_checksum.
GOLDEN_MODEL_SEED = 3


def golden_short_clips() -> List[np.ndarray]:
    """Three one-shot-like clips @ 48 kHz (decaying noise + tone, peak-normalised like augment_data_with_CLAP.py:51-63)."""
    rng = np.random.default_rng(77)
    out = []
    for n, f0 in ((4800, 180.0), (19000, 95.0), (30000, 3100.0)):
        t = np.arange(n, dtype=np.float64) / 48000.0
        x = np.exp(-t * 14.0) * (0.7 * np.sin(2 * np.pi * f0 * t) + 0.4 * rng.standard_normal(n))
        out.append((x / np.abs(x).max()).astype(np.float32))
    return out


def golden_long_clip() -> np.ndarray:
    """One clip longer than 10 s (521 337 samples): a periodic burst pattern, so that the three crops and the shrunk mel differ."""
    n = 521337
    rng = np.random.default_rng(78)
    t = np.arange(n, dtype=np.float64) / 48000.0
    x = np.exp(-(t % 0.73) * 7.0) * (0.6 * np.sin(2 * np.pi * (140.0 + 35.0 * t) * t) + 0.3 * rng.standard_normal(n))
    return (x / np.abs(x).max()).astype(np.float32)


def weights_checksum(model) -> np.ndarray:
    """float64 [sum, sum of squares] over every audio-tower and projection parameter / buffer, in state-dict order."""
    s = s2 = 0.0
    for k, v in model.state_dict().items():
        if k.startswith(("audio_model.", "audio_projection.")) and v.dtype.is_floating_point:
            d = v.double()
            s += float(d.sum())
            s2 += float((d * d).sum())
    return np.array([s, s2], dtype=np.float64)
