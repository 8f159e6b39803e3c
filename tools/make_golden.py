#!/usr/bin/env python3
"""Capture golden vectors from the reference's own Python (build container only).

Runs ONLY where ``/root/reference`` is mounted.  The reference source is
imported, never copied; nothing here travels to the GPU box except the small
``tests/golden/*.npz`` data files this script writes (inputs + expected
outputs).

Third-party packages the reference imports but that are not installed here get
container-only stand-ins registered in ``sys.modules``:

  * ``torchaudio`` / ``torchaudio.transforms.MelSpectrogram`` -- a module built on
    ``torch.stft`` + the htk/no-norm filterbank (torchaudio's published
    definition; this is the "parity unpinned" part of the log-mel row: the
    golden pins the reference-owned post-processing ``model.py:91-97`` and the
    module tree / buffer names only).
  * ``h5py`` -- a dict-backed ``File`` (a data container: no arithmetic).
  * ``pedalboard`` -- recording stand-ins (an effect remembers its constructor arguments; no audio processing);
    ``wandb``, ``pretty_midi`` -- inert names.
  * ``omegaconf`` -- PyYAML + recursive merge.

Usage: ``python tools/make_golden.py`` (writes tests/golden/).
"""
from __future__ import annotations

import os
import random
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import logmel as o_logmel  # noqa: E402  (stand-in torchaudio reuses the restated STFT/fb)
from oracle.bank import synthetic_bank  # noqa: E402
from oracle import clap as o_clap  # noqa: E402  (seeded random-init ClapModel + the G7 input recipe shared with the tests)


# --------------------------------------------------------------------------- stand-ins
def _install_standins(h5_store: dict):
    from transformers import PreTrainedModel, PretrainedConfig  # noqa: F401  (resolve lazily-imported parts before faking anything they probe)
    from importlib.machinery import ModuleSpec

    def _mod(name):
        m = types.ModuleType(name)
        m.__spec__ = ModuleSpec(name, None)
        return m

    ta = _mod("torchaudio")
    tat = _mod("torchaudio.transforms")

    class _Spectrogram(torch.nn.Module):
        def __init__(self, n_fft):
            super().__init__()
            self.register_buffer("window", o_logmel.hann_window(n_fft))

    class _MelScale(torch.nn.Module):
        def __init__(self, sr, n_fft, n_mels, f_min):
            super().__init__()
            self.register_buffer("fb", o_logmel.mel_filterbank(sr, n_fft, n_mels, f_min))

    class MelSpectrogram(torch.nn.Module):
        def __init__(self, sample_rate, n_fft, hop_length, n_mels, f_min=0.0, power=2):
            super().__init__()
            assert power == 2
            self.n_fft, self.hop_length = n_fft, hop_length
            self.spectrogram = _Spectrogram(n_fft)
            self.mel_scale = _MelScale(sample_rate, n_fft, n_mels, f_min)

        def forward(self, wave):
            return o_logmel.mel_power(wave, self.n_fft, self.hop_length,
                                      self.spectrogram.window, self.mel_scale.fb)

    tat.MelSpectrogram = MelSpectrogram
    tat.Resample = lambda *a, **k: (lambda x: x)
    ta.transforms = tat
    ta.load = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stand-in"))
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.transforms"] = tat

    h5 = _mod("h5py")

    class _Group:
        def __init__(self, d):
            self._d = d

        def __contains__(self, key):
            node = self._d
            for part in key.split("/"):
                if not isinstance(node, dict) or part not in node:
                    return False
                node = node[part]
            return True

        def __getitem__(self, key):
            node = self._d
            for part in key.split("/"):
                node = node[part]
            return _Group(node) if isinstance(node, dict) else node

        def keys(self):
            return self._d.keys()

    class File(_Group):
        def __init__(self, path, mode="r"):
            super().__init__(h5_store[path])

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    h5.File = File
    sys.modules["h5py"] = h5

    pb = _mod("pedalboard")
    # recording stand-ins: a board is a list, an effect remembers its constructor arguments (G9 pins the reference's parameter
    # sampling, synthetiser.py:44-87; the audio processing itself is C++/JUCE inside pedalboard and cannot be captured here)
    class _Effect:
        def __init__(self, *a, **k):
            assert not a
            self.kwargs = dict(k)
    for name in ("Reverb", "Compressor", "Limiter"):
        setattr(pb, name, type(name, (_Effect,), {}))

    class Pedalboard(list):
        pass
    pb.Pedalboard = Pedalboard
    sys.modules["pedalboard"] = pb
    sys.modules["wandb"] = _mod("wandb")
    sys.modules["pretty_midi"] = _mod("pretty_midi")

    oc = _mod("omegaconf")
    import yaml

    def _merge(a, b):
        out = dict(a)
        for k, v in b.items():
            out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
        return out

    class OmegaConf:
        load = staticmethod(lambda p: yaml.safe_load(open(p)))
        create = staticmethod(lambda d: d)
        merge = staticmethod(_merge)
        to_container = staticmethod(lambda c, resolve=True: c)

    oc.OmegaConf = OmegaConf
    sys.modules["omegaconf"] = oc


class _RecordingRandom:
    """Proxy for the ``random`` module inside modules/synthetiser.py that logs
    every draw the renderer makes (synthetiser.py:194,197,199,217)."""

    def __init__(self, seed):
        self._r = random.Random(seed)
        self.choices: list = []
        self.uniforms: list = []

    def choice(self, seq):
        v = self._r.choice(seq)
        self.choices.append(v)
        return v

    def uniform(self, a, b):
        v = self._r.uniform(a, b)
        self.uniforms.append(v)
        return v

    def random(self):
        return self._r.random()


# --------------------------------------------------------------------------- G1 log-mel
def g1_logmel(model_mod):
    cases = {}
    rng = np.random.default_rng(1234)
    specs = [("16k", 16000, 8000, 2), ("24k", 24000, 61440, 1), ("16k_edge", 16000, 5000, 3)]
    for name, sr, L, B in specs:
        wave = (rng.standard_normal((B, L)) * 0.05).astype(np.float32)
        t = np.arange(L) / sr
        for b in range(B):
            for _ in range(6):
                t0 = rng.uniform(0, L / sr * 0.9)
                f = rng.uniform(60, 6000)
                env = np.where(t >= t0, np.exp(-(t - t0) * rng.uniform(8, 60)), 0.0)
                wave[b] += (0.5 * env * np.sin(2 * np.pi * f * (t - t0))).astype(np.float32)
        wave = np.clip(wave, -1, 1)
        if name == "16k_edge":
            wave[1] = 0.0                                         # all-zero clip -> clamp floor
            wave[2] = np.where((np.arange(L) // 40) % 2 == 0, 1.0, -1.0)  # full-scale square
        m = model_mod.ComputeMelSpectrogram(sample_rate=sr, win_length=2048, time_res=0.01, n_mels=128)
        out = m(torch.from_numpy(wave)).contiguous().numpy()
        cases[f"{name}_wave"] = wave
        cases[f"{name}_out"] = out
        cases[f"{name}_sr"] = np.int64(sr)
        keys = sorted(k for k in m.state_dict().keys())
        cases[f"{name}_state_keys"] = np.array(keys)
    np.savez_compressed(os.path.join(OUT, "logmel.npz"), **cases)
    print("G1 logmel:", {k: v.shape for k, v in cases.items() if k.endswith("_out")})


# --------------------------------------------------------------------------- G2 tokenizer
def g2_tokenizer():
    from modules.midi_tokenizer import MidiTokenizer, MidiTokenizerConfig
    out = {}
    rng = np.random.default_rng(7)
    idx = 0
    for adtof in (False, True):
        for add_vel in (False, True):
            tk = MidiTokenizer(MidiTokenizerConfig(ADTOF_mapping=adtof, BOS_token=2, EOS_token=3,
                                                   pad_token=1, silence_token=0, add_velocity=add_vel))
            for n in (0, 1, 7, 40):
                onset = np.sort(rng.uniform(0, 2.95, n)).astype(np.float32)
                pitch = rng.integers(35, 82, n).astype(np.float32)
                vel = rng.integers(1, 128, n).astype(np.float32)
                notes = np.stack([onset, onset + 0.1, pitch, vel], 1).astype(np.float32) if n else np.zeros((0, 4), np.float32)
                mapped = tk.map_notes_to_Gm_custom(torch.from_numpy(notes.copy()), random_velocity=False).numpy() if n else notes
                toks = tk.notes_to_adt_tokens(torch.from_numpy(mapped.copy())).numpy()
                dec = tk.decode(toks.tolist()).numpy()
                out[f"c{idx}_cfg"] = np.array([int(adtof), int(add_vel)])
                out[f"c{idx}_notes"] = notes
                out[f"c{idx}_mapped"] = mapped
                out[f"c{idx}_tokens"] = toks
                out[f"c{idx}_decoded"] = dec.reshape(-1, 4) if dec.size else np.zeros((0, 4), np.float32)
                idx += 1
    tk = MidiTokenizer(MidiTokenizerConfig(False, 2, 3, 1, 0, True))
    out["empty_tokens"] = tk.empty_adt_tokens().numpy()
    # malformed sequences through decode (midi_tokenizer.py:69-100)
    bad = [[2, 340, 10, 335, 450, 3], [2, 10, 20, 335, 3], [2, 450, 10, 335, 3, 3], [2, 0, 3], [2, 10, 335, 12, 338, 3]]
    for i, seq in enumerate(bad):
        for add_vel in (False, True):
            tk = MidiTokenizer(MidiTokenizerConfig(False, 2, 3, 1, 0, add_vel))
            dec = tk.decode(seq).numpy()
            out[f"bad{i}_{int(add_vel)}_tokens"] = np.array(seq)
            out[f"bad{i}_{int(add_vel)}_decoded"] = dec.reshape(-1, 4) if dec.size else np.zeros((0, 4), np.float32)
    out["n_cases"] = np.int64(idx)
    out["n_bad"] = np.int64(len(bad))
    np.savez_compressed(os.path.join(OUT, "tokenizer.npz"), **out)
    print("G2 tokenizer:", idx, "cases")


# --------------------------------------------------------------------------- G3 masks / collate
def g3_masks_collate():
    from utils.utils import create_mask_plain
    sys.argv = ["x", "dummy.yaml"]          # train_dataset.py parses argv at import (train_dataset.py:232-234)
    from data_modules.train_dataset import collate_fn
    out = {}
    for i, (T, lens) in enumerate([(5, [5, 3, 1]), (1, [1]), (9, [0, 9, 4, 8])]):
        cm, pm = create_mask_plain(T, torch.tensor(lens), None)
        out[f"m{i}_T"] = np.int64(T)
        out[f"m{i}_lens"] = np.array(lens)
        out[f"m{i}_causal"] = cm.numpy()
        out[f"m{i}_pad"] = pm.numpy()
    rng = np.random.default_rng(3)
    batches = [[(100, 5), (80, 9), (100, 9), (60, 2)], [(10, 3)], [(7, 4), (9, 4)]]
    for i, spec in enumerate(batches):
        batch = [(torch.from_numpy(rng.standard_normal(w).astype(np.float32)),
                  torch.from_numpy(rng.integers(0, 1400, t))) for w, t in spec]
        res = collate_fn([(w, t.tolist()) for w, t in batch])
        out[f"c{i}_n"] = np.int64(len(spec))
        for j, (w, t) in enumerate(batch):
            out[f"c{i}_wav{j}"] = w.numpy()
            out[f"c{i}_tok{j}"] = t.numpy()
        out[f"c{i}_wavs"] = res["wavs"].numpy()
        out[f"c{i}_tokens"] = res["tokens"].numpy()
        out[f"c{i}_token_lengths"] = res["token_lengths"].numpy()
    out["n_masks"] = np.int64(3)
    out["n_collate"] = np.int64(len(batches))
    np.savez_compressed(os.path.join(OUT, "masks_collate.npz"), **out)
    print("G3 masks/collate ok")


# --------------------------------------------------------------------------- G4 mixer
def g4_mixer(h5_store):
    import modules.synthetiser as synth_mod
    out = {}
    sr = 16000
    bank = synthetic_bank(seed=11, sample_rate=sr)
    path = f"/fake/oneshot@{sr}.hdf5"
    h5_store[path] = bank.as_tree()
    case = 0
    for adtof, thr, mixup_range, input_sec, n_notes, seed in [
        (False, 0.8, 0.8, 1.0, 12, 0), (False, 1.0, 0.0, 1.0, 5, 1), (True, 0.8, 0.5, 1.0, 10, 2),
        (False, 0.8, 0.8, 0.5, 6, 3), (False, 0.9, 0.3, 1.0, 0, 4),
    ]:
        cfg = synth_mod.SynthDrumConfig(
            input_sec=input_sec, time_res=0.01, win_length=2048, sample_rate=sr,
            oneshot_path="/fake/oneshot", similarity_threshold=thr, max_hat_std_velocity=0.15,
            max_hat_mean_velocity=0.1, max_cymbals_std_velocity=0.15, max_cymbals_mean_velocity=0.65,
            ADTOF_mapping=adtof, mixup_range=mixup_range, use_fx_prob=0.0, use_reverb_prob=0.5,
            use_limiter_prob=0.5, use_compression_prob=0.5)
        sd = synth_mod.SynthDrum(cfg)
        rng = np.random.default_rng(100 + seed)
        onset = np.sort(rng.uniform(0, input_sec * 0.95 if case != 3 else 0.9, n_notes)).astype(np.float32)
        if adtof:
            pitch = rng.choice([35, 38, 41, 42, 48, 52, 58, 61], n_notes).astype(np.float32)
        else:
            pitch = rng.integers(35, 61, n_notes).astype(np.float32)
        vel = rng.integers(0, 128, n_notes).astype(np.float32)
        if n_notes:
            vel[0] = 0.0 if case == 1 else vel[0]          # velocity 0 -> volume 0 (synthetiser.py:205-206)
        notes = np.stack([onset, onset + 0.1, pitch, vel], 1).astype(np.float32) if n_notes else np.zeros((0, 4), np.float32)
        rec = _RecordingRandom(seed)
        synth_mod.random = rec
        wav = sd(notes.tolist())
        synth_mod.random = random
        out[f"s{case}_cfg"] = np.array([int(adtof), thr, mixup_range, input_sec, sr], np.float64)
        out[f"s{case}_notes"] = notes
        out[f"s{case}_wav"] = wav.numpy().astype(np.float32)
        out[f"s{case}_uniforms"] = np.array(rec.uniforms, np.float64)
        out[f"s{case}_choices"] = np.array([str(c) for c in rec.choices])
        out[f"s{case}_seed"] = np.int64(seed)
        case += 1
    out["n_cases"] = np.int64(case)
    out["bank_seed"] = np.int64(11)
    np.savez_compressed(os.path.join(OUT, "mixer.npz"), **out)
    print("G4 mixer:", case, "cases")


def g9_fx_params():
    """BoardChain.get_board (synthetiser.py:30-87) for seeded RNG states: which effects, in which order, with which parameters.
    A fresh BoardChain per draw (the reference keeps ONE board per VolumeMixer and appends to it on every call, so its chain
    grows over a run -- not reproduced)."""
    import json
    import modules.synthetiser as synth_mod
    draws = []
    for seed in range(24):
        random.seed(seed)
        torch.manual_seed(seed)
        probs = [(0.5, 0.5, 0.5), (1.0, 1.0, 1.0), (0.0, 1.0, 0.3)][seed % 3]
        board = synth_mod.BoardChain(16000, *probs).get_board()
        draws.append({"seed": seed, "probs": list(probs),
                      "effects": [[type(e).__name__, {k: float(v) for k, v in e.kwargs.items()}] for e in board]})
    np.savez_compressed(os.path.join(OUT, "fx_params.npz"), draws=np.array(json.dumps(draws)))
    print("G9 fx params:", len(draws), "boards,", sum(len(d["effects"]) for d in draws), "effects")


# --------------------------------------------------------------------------- G5/G6 ADT network
def _adt_config(tiny: bool):
    from config import ADTModelConfig
    if tiny:
        return ADTModelConfig(input_sec=0.5, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=1,
                              dec_layers=1, nhead=2, d_query=16, dropout=0.0, tgt_vocab_size=1400, enc_lr=1e-4,
                              dec_lr=1e-4, plain=True, n_mels=128)
    return ADTModelConfig(input_sec=2.56, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=4,
                          dec_layers=4, nhead=6, d_query=128, dropout=0.0, tgt_vocab_size=1400, enc_lr=1e-4,
                          dec_lr=1e-4, plain=True, n_mels=128)


def _make_batch(rng, B, L, T):
    wave = np.clip(rng.standard_normal((B, L)) * 0.1, -1, 1).astype(np.float32)
    lens = rng.integers(max(T // 3, 2), T + 2, B)
    lens[0] = T + 1
    tokens = np.full((B, T + 1), 1, np.int64)
    for b in range(B):
        n = int(lens[b])
        body = rng.integers(4, 530, n - 2)
        tokens[b, :n] = np.concatenate([[2], body, [3]])
    token_lengths = np.where(lens == lens.max(), lens - 1, lens)   # collate rule (train_dataset.py:46-51)
    return wave, tokens, token_lengths.astype(np.int64)


def g5_adt_tiny(model_mod):
    from utils.utils import create_mask_plain
    torch.manual_seed(0)
    cfg = _adt_config(tiny=True)
    model = model_mod.ADTModel(cfg)
    with torch.no_grad():      # non-trivial biases / LN affine so every term is exercised
        for n, p in model.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.05)
    rng = np.random.default_rng(5)
    B, L, T = 3, 8000, 12
    wave, tokens, token_lengths = _make_batch(rng, B, L, T)
    model.train()
    src = torch.from_numpy(wave)
    tok = torch.from_numpy(tokens)
    tgt_in, labels = tok[:, :-1], tok[:, 1:]
    _, pad_mask = create_mask_plain(tgt_in.size(1), torch.from_numpy(token_lengths), None)
    cap = {}
    h1 = model.encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("memory", o.detach()))
    h2 = model.decoder.register_forward_hook(lambda m, i, o: cap.__setitem__("logits", o.detach()))
    h3 = model.compute_spectrogram.register_forward_hook(lambda m, i, o: cap.__setitem__("logmel", o.detach()))
    loss = model(src=src, tgt=tgt_in, tgt_mask=None, tgt_padding_mask=pad_mask, labels=labels)
    loss.backward()
    for h in (h1, h2, h3):
        h.remove()
    out = {"wave": wave, "tokens": tokens, "token_lengths": token_lengths,
           "logmel": cap["logmel"].contiguous().numpy(), "memory": cap["memory"].numpy(),
           "logits": cap["logits"].numpy(), "loss": loss.detach().numpy()}
    sd = model.state_dict()
    out["state_keys"] = np.array(list(sd.keys()))
    for k, v in sd.items():
        if "pos_embedding" in k or k.startswith("compute_spectrogram."):
            continue                              # constants: regenerated, not stored
        out["w::" + k] = v.numpy()
    for name in ("project_to_mel.weight", "encoder.encoder.layers.0.self_attn.in_proj_weight",
                 "encoder.encoder.layers.0.linear1.bias", "encoder.layer_norm.weight",
                 "decoder.decoder.layers.0.multihead_attn.in_proj_weight",
                 "decoder.decoder.layers.0.norm3.bias", "decoder.tgt_tok_emb.embedding.weight",
                 "decoder.generator.bias", "encoder.dense_layer.weight"):
        out["g::" + name] = dict(model.named_parameters())[name].grad.numpy()
    # greedy sample (model.py:260-324) -- eval mode, grad enabled as in the reference
    gen = model.sample(src=src, src_mask=None, tgt_mask=None, max_length=10, start_token=2, end_token=3)
    out["sample_ids"] = gen.numpy()
    np.savez_compressed(os.path.join(OUT, "adt_tiny.npz"), **out)
    print("G5 adt tiny: loss", float(loss), "logits", cap["logits"].shape, "sample", gen.shape)


def g6_adt_full(model_mod):
    """Full-size (setting-1 architecture) statistics only; weights are seeded
    by a portable numpy recipe that tests re-create (oracle/adt.py:seeded_state)."""
    from utils.utils import create_mask_plain
    from oracle.adt import seeded_state
    cfg = _adt_config(tiny=False)
    model = model_mod.ADTModel(cfg)
    state = seeded_state(model.state_dict(), seed=0)
    model.load_state_dict(state)
    rng = np.random.default_rng(6)
    B, L, T = 2, 40960, 24
    wave, tokens, token_lengths = _make_batch(rng, B, L, T)
    model.train()
    tok = torch.from_numpy(tokens)
    tgt_in, labels = tok[:, :-1], tok[:, 1:]
    _, pad_mask = create_mask_plain(tgt_in.size(1), torch.from_numpy(token_lengths), None)
    cap = {}
    h2 = model.decoder.register_forward_hook(lambda m, i, o: cap.__setitem__("logits", o.detach()))
    with torch.no_grad():
        loss = model(src=torch.from_numpy(wave), tgt=tgt_in, tgt_mask=None, tgt_padding_mask=pad_mask, labels=labels)
    h2.remove()
    lg = cap["logits"].double()
    out = {"seed": np.int64(0), "batch_seed": np.int64(6), "B": np.int64(B), "L": np.int64(L), "T": np.int64(T),
           "loss": loss.numpy(), "logits_mean": lg.mean().numpy(), "logits_std": lg.std().numpy(),
           "logits_absmax": lg.abs().max().numpy(), "logits_sample": cap["logits"][:, ::5, ::97].numpy(),
           "state_keys": np.array(list(model.state_dict().keys())),
           "state_shapes": np.array([str(tuple(v.shape)) for v in model.state_dict().values()]),
           "n_params": np.int64(sum(p.numel() for p in model.parameters()))}
    np.savez_compressed(os.path.join(OUT, "adt_full_stats.npz"), **out)
    print("G6 adt full: loss", float(loss), "params", int(out["n_params"]))


# --------------------------------------------------------------------------- G8 curation
def g8_curation():
    """Runs the reference's own curation lines (augment_data_with_CLAP.py:139-151 similarity + sort,
    :160-193 binning + greedy copy) on synthetic embeddings.  The script keeps that logic inline under
    ``__main__``, so the lines are executed from the mounted file with file copying / progress bars faked."""
    src = open(os.path.join(REF, "data_modules", "augment_data_with_CLAP.py")).read().split("\n")
    sim_block = "\n".join(l[4:] for l in src[138:151])            # "# Compute cosine similarity" .. scores.sort
    bin_block = "\n".join(l[4:] for l in src[159:193])            # bin_size .. pbar.close()
    assert "cosine_similarity" in sim_block and "scores.sort" in sim_block and "score_to_bin_label" in bin_block
    rng = np.random.default_rng(8)
    out = {}
    for case, (n_u, labels, num_bins, dim) in enumerate([(300, list(range(35, 47)), 10, 64), (64, [35, 38, 42, 421], 5, 512),
                                                          (500, list(range(35, 82)) + [421], 10, 32)]):
        C = len(labels)
        protos = rng.standard_normal((C, dim)).astype(np.float32)
        refs = {k: [] for k in labels}
        for ci, k in enumerate(labels):
            for _ in range(3):
                v = protos[ci] + 0.5 * rng.standard_normal(dim).astype(np.float32)
                refs[k].append(torch.from_numpy(v / np.linalg.norm(v)))
        means = torch.stack([torch.mean(torch.stack(refs[k]), dim=0) for k in labels])
        samp = protos[rng.integers(0, C, n_u)] * rng.uniform(0.2, 1.5, (n_u, 1)).astype(np.float32) + \
            rng.standard_normal((n_u, dim)).astype(np.float32) * rng.uniform(0.3, 3.0, (n_u, 1)).astype(np.float32)
        samp = samp / np.linalg.norm(samp, axis=1, keepdims=True)
        samp[5] = samp[4]                                               # exact duplicate -> tie handling
        samp_t = torch.from_numpy(samp.astype(np.float32))
        copies = []

        class _Bar:
            def __init__(self, *a, **k): pass
            def update(self, n=1): pass
            def close(self): pass

        class _Shutil:
            @staticmethod
            def copy2(srcp, dst): copies.append((srcp, str(dst)))

        class _P:                                                       # pathlib.Path stand-in without filesystem access
            def __init__(self, p): self.p = str(p)
            def __truediv__(self, o): return _P(self.p + "/" + str(o))
            def mkdir(self, **k): pass
            @property
            def name(self): return self.p.split("/")[-1]
            def __str__(self): return self.p

        ns = dict(torch=torch, tqdm=_Bar, reference_embeddings=means, non_empty_keys=labels, sample_pack_embeddings=samp_t,
                  wav_files=[f"/packs/s{i:05d}.wav" for i in range(n_u)], num_bins=num_bins, shutil=_Shutil, Path=_P,
                  augmented_root=_P("/aug"), print=lambda *a, **k: None)
        exec(compile(sim_block, "ref_sim", "exec"), ns)
        exec(compile(bin_block, "ref_bin", "exec"), ns)
        assert len(copies) == n_u
        out[f"c{case}_labels"] = np.array(labels)
        out[f"c{case}_num_bins"] = np.int64(num_bins)
        out[f"c{case}_means"] = means.numpy()
        out[f"c{case}_samples"] = samp_t.numpy()
        out[f"c{case}_order"] = np.array([int(c[0][-9:-4]) for c in copies])
        out[f"c{case}_class"] = np.array([int(c[1].split("/")[2]) for c in copies])
        out[f"c{case}_bin"] = np.array([c[1].split("/")[3] for c in copies])
    out["n_cases"] = np.int64(3)
    np.savez_compressed(os.path.join(OUT, "curation.npz"), **out)
    print("G8 curation: 3 cases")


def g7_clap():
    """G7: the reference's own ``ClapWrapper.get_audio_features`` / ``_get_audio_features`` (modules/clap_encoder.py:21-54)
    on a random-init fused-HTSAT ``ClapModel``.  ``ClapWrapper.__init__`` needs the network (``from_pretrained``), so the instance is made
    with ``__new__`` + ``nn.Module.__init__`` and given the attributes ``__init__`` would have set; the processor forwards to
    ``ClapFeatureExtractor`` (what ``ClapProcessor`` does for ``audio=``).  Everything after that is the reference's code.

    Case A: three short clips (the extractor then flags ONE random clip ``is_longer``: numpy's global RNG, seeded here).
    Case B: a short clip and one longer than 10 s (three random crops + the shrunk mel; no forced flag).
    Stored: the inputs (the long clip as its recipe, ``oracle.clap.golden_long_clip``), ``is_longer``, every 8th frame of channel 0 and 3
    of ``input_features`` + per-clip sums, pooled [B, 768], embedding [B, 512], the numpy seeds and a checksum of the weights."""
    from transformers import ClapFeatureExtractor
    from modules.clap_encoder import ClapWrapper          # /root/reference/modules/clap_encoder.py
    assert ClapWrapper.__module__ == "modules.clap_encoder" and sys.modules["modules.clap_encoder"].__file__.startswith(REF)
    model = o_clap.random_clap_model(o_clap.GOLDEN_MODEL_SEED)

    class _Processor:                                      # ClapProcessor(audio=...) == its feature extractor's __call__
        def __init__(self):
            self.fe = ClapFeatureExtractor()

        def __call__(self, audio=None, return_tensors=None, sampling_rate=None, **kw):
            return self.fe(audio, return_tensors=return_tensors, sampling_rate=sampling_rate, **kw)

    w = ClapWrapper.__new__(ClapWrapper)
    torch.nn.Module.__init__(w)
    w.clap_model, w.device, w.sample_rate = model, torch.device("cpu"), 48000
    w.config, w.text_model, w.audio_model = model.config, model.text_model, model.audio_model
    w.processor = _Processor()
    captured = {}
    inner = w._get_audio_features

    def spy(**inputs):
        captured["input_features"] = inputs["input_features"].clone()
        captured["is_longer"] = inputs["is_longer"].clone()
        pooled = {}
        h = model.audio_model.register_forward_hook(lambda m, a, o: pooled.__setitem__("p", o.pooler_output.clone()))
        try:
            out = inner(**inputs)
        finally:
            h.remove()
        captured["pooled"] = pooled["p"]
        return out

    w._get_audio_features = spy
    out = {"model_seed": np.int64(o_clap.GOLDEN_MODEL_SEED), "weights_checksum": o_clap.weights_checksum(model)}
    for case, (clips, np_seed) in {"a": (o_clap.golden_short_clips(), 11), "b": ([o_clap.golden_short_clips()[1], o_clap.golden_long_clip()], 12)}.items():
        np.random.seed(np_seed)
        with torch.no_grad():
            emb = w.get_audio_features([torch.from_numpy(c).unsqueeze(0) for c in clips])
        f = captured["input_features"].numpy()
        out[f"{case}_np_seed"] = np.int64(np_seed)
        out[f"{case}_n"] = np.int64(len(clips))
        out[f"{case}_lengths"] = np.array([len(c) for c in clips], dtype=np.int64)
        if case == "a":
            for i, c in enumerate(clips):
                out[f"a_clip{i}"] = c
        out[f"{case}_is_longer"] = captured["is_longer"].numpy()
        out[f"{case}_feat_ch0_every8"] = f[:, 0, ::8].copy()
        out[f"{case}_feat_ch3_every8"] = f[:, 3, ::8].copy()
        out[f"{case}_feat_sum"] = f.astype(np.float64).sum(axis=(2, 3))
        out[f"{case}_pooled"] = captured["pooled"].numpy()
        out[f"{case}_embedding"] = emb.numpy()
        assert emb.shape == (len(clips), 512) and np.allclose(np.linalg.norm(emb.numpy(), axis=-1), 1.0, atol=1e-5)
    np.savez_compressed(os.path.join(OUT, "clap.npz"), **out)
    print("G7 clap: is_longer a", out["a_is_longer"].ravel().tolist(), "b", out["b_is_longer"].ravel().tolist(),
          "bytes", os.path.getsize(os.path.join(OUT, "clap.npz")))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    h5_store: dict = {}
    _install_standins(h5_store)
    # The reference's packages (modules/, utils/, data_modules/) are namespace packages; this repository has regular packages of
    # the same names (its drop-in import paths), which would win.  Everything needed from this repository is imported above, so
    # take it (and the cwd) off the path before the reference goes on.
    sys.path[:] = [q for q in sys.path if os.path.abspath(q or ".") != REPO]
    for name in [m for m in sys.modules if m.split(".")[0] in ("modules", "utils", "data_modules", "config", "model", "train", "build_model")]:
        del sys.modules[name]
    sys.path.insert(0, REF)
    sys.argv = ["make_golden", "dummy.yaml"]
    import model as model_mod                      # /root/reference/model.py
    which = set(sys.argv[2:]) if len(sys.argv) > 2 else None
    g1_logmel(model_mod)
    g2_tokenizer()
    g3_masks_collate()
    g4_mixer(h5_store)
    g5_adt_tiny(model_mod)
    g6_adt_full(model_mod)
    g8_curation()
    g9_fx_params()
    g7_clap()


if __name__ == "__main__":
    main()
