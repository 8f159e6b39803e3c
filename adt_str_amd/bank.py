"""Flat, HBM-resident one-shot bank.

The reference keeps one-shots in an HDF5 file ``<prefix>@<sr>.hdf5`` laid out
``/<gm_custom_pitch>/<similarity_group>/<name>`` (1-D float32, gzip;
``data_modules/convert_augmented_to_hdf5.py:69-141``) and re-opens it for every
note (``modules/synthetiser.py:273``).  Here the same content is one contiguous
float32 array plus offsets, uploaded to the GPU once:

  data      float32 [total_samples]
  offsets   int64   [n_shots + 1]
  pitch/group/name per shot; ``cells[(pitch, group)]`` = shot ids in name order
  (h5py lists group members in name order, which is the order
  ``random.choice(list(...keys()))`` indexes, synthetiser.py:199).

On disk it is an ``.npz`` with those arrays (``save`` / ``load``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch

# similarity groups from best to worst (synthetiser.py:172-184)
GROUPS = ["gold", "100-90", "90-80", "80-70", "70-60", "60-50", "50-40", "40-30", "30-20", "20-10", "10-0"]


@dataclass
class OneShotBank:
    data: np.ndarray
    offsets: np.ndarray
    pitch: np.ndarray            # int32 [n_shots]
    group: np.ndarray            # int32 [n_shots] index into GROUPS
    names: List[str]
    sample_rate: int
    cells: Dict[Tuple[int, str], List[int]] = field(default_factory=dict)
    _dev: dict = field(default_factory=dict, repr=False)

    def __post_init__(self):
        if not self.cells:
            order = sorted(range(len(self.names)), key=lambda i: (int(self.pitch[i]), int(self.group[i]), self.names[i]))
            for i in order:
                self.cells.setdefault((int(self.pitch[i]), GROUPS[int(self.group[i])]), []).append(i)

    @property
    def n_shots(self) -> int:
        return len(self.names)

    def shot(self, i: int) -> np.ndarray:
        return self.data[self.offsets[i]:self.offsets[i + 1]]

    def has_cell(self, pitch: int, group: str) -> bool:
        return (int(pitch), group) in self.cells

    def cell_names(self, pitch: int, group: str) -> List[str]:
        return [self.names[i] for i in self.cells[(int(pitch), group)]]

    def shot_id(self, pitch: int, group: str, name: str) -> int:
        for i in self.cells[(int(pitch), group)]:
            if self.names[i] == name:
                return i
        raise KeyError(f"{pitch}/{group}/{name}")

    @staticmethod
    def from_tree(tree: dict, sample_rate: int) -> "OneShotBank":
        """Build from the HDF5-shaped nested dict ``{pitch: {group: {name: array}}}``."""
        chunks, offsets, pitch, group, names = [], [0], [], [], []
        for p in sorted(tree, key=lambda s: int(s)):
            for g in sorted(tree[p], key=GROUPS.index):
                for name in sorted(tree[p][g]):
                    x = np.asarray(tree[p][g][name], dtype=np.float32).reshape(-1)
                    chunks.append(x)
                    offsets.append(offsets[-1] + x.size)
                    pitch.append(int(p))
                    group.append(GROUPS.index(g))
                    names.append(name)
        data = np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
        return OneShotBank(data=data, offsets=np.asarray(offsets, np.int64), pitch=np.asarray(pitch, np.int32),
                           group=np.asarray(group, np.int32), names=names, sample_rate=sample_rate)

    @staticmethod
    def from_directory(root: str, sample_rate: int, device=None) -> "OneShotBank":
        """Build from a curated tree ``<root>/<label>/<bin>/<file>.wav`` -- what the reference's
        ``augment_data_with_CLAP.py`` + ``copy_originals_to_augmented.py`` leave on disk and its
        ``convert_augmented_to_hdf5.py:69-141`` packs into HDF5: every file is loaded as mono, resampled to ``sample_rate``
        (K13 on the GPU when ``device`` is a GPU and the rate differs) and peak-normalised (``:101-103``); files are taken in
        sorted path order, paths with fewer than three components are skipped (``:93-96``), and a repeated stem inside one
        cell gets the ``_2``, ``_3`` ... suffix (``:113-118``).  Unreadable or silent files are skipped with a message, like
        the reference's ``except``."""
        import glob
        import os
        from .audio_io import read_wav, read_wav_batch
        files = sorted(glob.glob(os.path.join(root, "**", "*.[Ww][Aa][Vv]"), recursive=True))
        tree: dict = {}
        resamplers: dict = {}
        todo = []                                                     # (path, label, group, stem) in sorted path order
        for path in files:
            rel = os.path.relpath(path, root).split(os.sep)
            if len(rel) < 3 or rel[1] not in GROUPS:
                continue
            try:
                int(rel[0])
            except ValueError as e:
                print(f"Failed to load '{path}': {e}")
                continue
            todo.append((path, rel[0], rel[1], os.path.splitext(rel[-1])[0]))

        def per_file(path):                                           # another rate: mono -> K13 on the GPU -> peak-normalise
            audio, sr = read_wav(path)
            x = audio.mean(axis=0)
            if sr != sample_rate:
                from .resample import Resample
                if device is None or torch.device(device).type != "cuda":
                    raise RuntimeError(f"{path} is {sr} Hz: resampling to {sample_rate} Hz runs on the GPU (pass device=)")
                if sr not in resamplers:
                    resamplers[sr] = Resample(sr, sample_rate)
                x = resamplers[sr](torch.from_numpy(x).to(device)[None])[0].cpu().numpy()
            peak = float(np.abs(x).max()) if x.size else 0.0
            if not peak > 0.0:
                raise ValueError("silent or empty file")
            return (x / peak).astype(np.float32)

        # decoded, down-mixed, (on a GPU) resampled and normalised a few thousand files per call (audio_io.read_wav_batch /
        # load_clips_batch: read_wav(path)[0].mean(axis=0) -> K13 -> x / peak, bitwise); the shots are views of the batch buffers
        from .audio_io import load_clips_batch
        gpu = device is not None and torch.device(device).type == "cuda"
        for lo in range(0, len(todo), 4096):
            part = todo[lo:lo + 4096]
            paths = [t[0] for t in part]
            if gpu:
                clips, peaks, status = load_clips_batch(paths, sample_rate, device, normalize=True)
                good = [c is not None for c in clips]
                flat = torch.cat([c for c in clips if c is not None]).cpu().numpy() if any(good) else np.zeros(0, np.float32)
                ends = np.cumsum([c.numel() if c is not None else 0 for c in clips])
                peaks = peaks.cpu().numpy()
            else:
                b = read_wav_batch(paths, normalize=True)
                good = ((b.status == 0) & (b.sample_rate == sample_rate)).tolist()
                flat, ends, peaks = b.data.numpy(), b.offsets[1:], b.peak
            for j, (path, label, group, stem) in enumerate(part):
                try:
                    if good[j]:
                        if not peaks[j] > 0.0:
                            raise ValueError("silent or empty file")
                        x = flat[ends[j - 1] if j else 0:ends[j]]
                    else:
                        x = per_file(path)                            # raises what read_wav raises for an unreadable file
                except Exception as e:
                    print(f"Failed to load '{path}': {e}")
                    continue
                cell = tree.setdefault(label, {}).setdefault(group, {})
                name, k = stem, 1
                while name in cell:
                    k += 1
                    name = f"{stem}_{k}"
                cell[name] = x
        return OneShotBank.from_tree(tree, sample_rate)

    def save(self, path: str) -> None:
        np.savez(path, data=self.data, offsets=self.offsets, pitch=self.pitch, group=self.group,
                 names=np.array(self.names), sample_rate=np.int64(self.sample_rate))

    @staticmethod
    def load(path: str) -> "OneShotBank":
        z = np.load(path, allow_pickle=False)
        return OneShotBank(data=z["data"], offsets=z["offsets"], pitch=z["pitch"], group=z["group"],
                           names=[str(n) for n in z["names"]], sample_rate=int(z["sample_rate"]))

    def device_arrays(self, device) -> Tuple[torch.Tensor, torch.Tensor]:
        """(data, offsets) on ``device``; uploaded once per device."""
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.from_numpy(self.data).to(device), torch.from_numpy(self.offsets).to(device))
        return self._dev[key]


def synthetic_tree(seed: int, sample_rate: int, pitches=range(35, 62), groups=("gold", "100-90", "90-80"),
                   shots_per_group: int = 4, min_sec: float = 0.15, max_sec: float = 1.0) -> dict:
    """Procedural stand-in for a curated one-shot library (decaying noise + a sine partial,
    peak-normalised like convert_augmented_to_hdf5.py:101-103).  Used by bench.py / demos, where
    no real sample pack is available."""
    rng = np.random.default_rng(seed)
    tree: dict = {}
    for p in pitches:
        tree[str(p)] = {}
        for gi, g in enumerate(groups):
            cell = {}
            for s in range(shots_per_group):
                n = int(rng.uniform(min_sec, max_sec) * sample_rate)
                t = np.arange(n) / sample_rate
                x = np.exp(-rng.uniform(6.0, 60.0) * t) * (0.6 * rng.standard_normal(n) + np.sin(2 * np.pi * rng.uniform(50.0, 5000.0) * t))
                cell[f"synth_{p}_{gi}_{s}.wav"] = (x / np.abs(x).max()).astype(np.float32)
            tree[str(p)][g] = cell
    return tree
