"""On-GPU one-shot drum renderer: drop-in for the reference's ``SynthDrum``
(``modules/synthetiser.py:159-292``) on top of ``adt_mix_render_f32``.

``SynthDrum(config)(notes) -> wav[W]`` keeps the reference call; the batch entry
point ``render_batch(list_of_notes)`` renders every clip of a training batch in
one C-ABI call from the HBM-resident flat bank (``bank.OneShotBank``) instead of
re-opening an HDF5 file per note (``synthetiser.py:273``).

Random draws come from Python's ``random`` in exactly the order the reference
consumes them (timbre picks once per pitch and clip ``:192-202,274-281``, then
one ``uniform(0, mixup_range)`` per note ``:217``), so seeding ``random`` gives
the reference's clip.  The optional FX chain (``use_fx_prob``; pedalboard Reverb / Compressor / Limiter in the
reference, ``:30-87,121-137,154-155``) is drawn here in the reference's order too (``random`` for the coin flips and
the reverb, ``torch.randn`` for the compressor / limiter, ``utils/utils.py:266-269``) and processed on the GPU by K14
(``adt_mix_render_fx_f32``) between the mix and its peak normalisation.  Every clip gets a fresh chain: the reference
appends to one shared ``Pedalboard`` on every call, so its chain grows during a run -- not reproduced.
"""
from __future__ import annotations

import math
import os
import random
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _ffi
from .bank import OneShotBank
from .mapping import ADTOF_INVERSE_MAPPING, ADTOF_LABEL, ADTOF_MAPPING

# per-instrument mix volume (synthetiser.py:104-113)
VOLUME_PER_INSTRUMENT = {"BD": 1.0, "SD": 1.0, "TT": 1.0, "HH": 0.7, "CY + RD": 0.7, "Cowbell": 0.7, "Claves": 0.7,
                         "Other": 1.0}
_THR_GROUP = {1.0: "gold", 0.9: "100-90", 0.8: "90-80", 0.7: "80-70", 0.6: "70-60", 0.5: "60-50", 0.4: "50-40",
              0.3: "40-30", 0.2: "30-20", 0.1: "20-10", 0.0: "10-0"}

NOTE_DTYPE = np.dtype([("start", "<i4"), ("main_shot", "<i4"), ("sub_shot", "<i4"), ("track", "<i4"),
                       ("one_minus_mixup", "<f4"), ("mixup", "<f4"), ("vol", "<f4"), ("track_gain", "<f4")])
assert NOTE_DTYPE.itemsize == 32          # struct adt_note in include/adt_hip.h
FX_DTYPE = np.dtype([("flags", "<i4"), ("room_size", "<f4"), ("damping", "<f4"), ("wet_level", "<f4"), ("dry_level", "<f4"),
                     ("width", "<f4"), ("c_threshold_db", "<f4"), ("c_ratio", "<f4"), ("c_attack_ms", "<f4"), ("c_release_ms", "<f4"),
                     ("l_threshold_db", "<f4"), ("l_release_ms", "<f4")])
assert FX_DTYPE.itemsize == 48            # struct adt_fx_params
LIMITER_RELEASE_MS = 100.0                # pedalboard.Limiter default (the reference passes threshold_db only, synthetiser.py:78)


def draw_from_normal_distribution(std: float, mean: float, high_bound: float, low_bound: float) -> float:
    """utils/utils.py:266-269: one ``torch.randn(1)`` from the global CPU generator."""
    return torch.clamp(torch.clamp(torch.randn(1) * std + mean, -1.0, 1.0).abs() * high_bound, low_bound, high_bound).item()


def draw_board(use_reverb_prob: float, use_compression_prob: float, use_limiter_prob: float):
    """``BoardChain.get_board`` on a fresh board (synthetiser.py:44-87) -> list of (effect name, kwargs), same draws in the same order."""
    board = []
    if random.random() < use_reverb_prob:
        room_size = random.uniform(0.2, 0.8)
        damping = random.uniform(0.2, 0.8)
        wet_level = random.uniform(0.1, 0.4)
        width = random.uniform(0.6, 1.0)
        board.append(("Reverb", dict(room_size=room_size, damping=damping, wet_level=wet_level, dry_level=1 - wet_level, width=width,
                                     freeze_mode=0.0)))
    if random.random() < use_compression_prob:
        threshold = -draw_from_normal_distribution(std=0.15, mean=0.5, high_bound=10, low_bound=0)
        ratio = draw_from_normal_distribution(std=0.15, mean=0.5, high_bound=10, low_bound=1.0)
        attack = draw_from_normal_distribution(std=0.05, mean=0.1, high_bound=1000, low_bound=0)
        release = draw_from_normal_distribution(std=0.15, mean=0.2, high_bound=1000, low_bound=0)
        board.append(("Compressor", dict(threshold_db=threshold, ratio=ratio, attack_ms=attack, release_ms=release)))
    if random.random() < use_limiter_prob:
        threshold = -draw_from_normal_distribution(std=0.2, mean=0.4, high_bound=3, low_bound=0)
        board.append(("Limiter", dict(threshold_db=threshold)))
    return board


def board_to_record(board) -> np.ndarray:
    """[(name, kwargs)] -> one FX_DTYPE record (struct adt_fx_params)."""
    rec = np.zeros((), FX_DTYPE)
    rec["c_ratio"], rec["l_release_ms"] = 1.0, LIMITER_RELEASE_MS
    for name, kw in board:
        if name == "Reverb":
            rec["flags"] |= 1
            for k in ("room_size", "damping", "wet_level", "dry_level", "width"):
                rec[k] = kw[k]
        elif name == "Compressor":
            rec["flags"] |= 2
            rec["c_threshold_db"], rec["c_ratio"], rec["c_attack_ms"], rec["c_release_ms"] = (kw["threshold_db"], kw["ratio"], kw["attack_ms"],
                                                                                              kw["release_ms"])
        elif name == "Limiter":
            rec["flags"] |= 4
            rec["l_threshold_db"] = kw["threshold_db"]
            rec["l_release_ms"] = kw.get("release_ms", LIMITER_RELEASE_MS)
        else:
            raise ValueError(f"unknown effect {name}")
    return rec


@dataclass
class SynthDrumConfig:
    """Same fields as the reference's ``SynthDrumConfig(SharedConfig)`` (synthetiser.py:14-27, config.py:8-13)."""
    input_sec: float
    time_res: float
    win_length: int
    sample_rate: int
    oneshot_path: str
    similarity_threshold: float
    max_hat_std_velocity: float
    max_hat_mean_velocity: float
    max_cymbals_std_velocity: float
    max_cymbals_mean_velocity: float
    ADTOF_mapping: bool
    mixup_range: float
    use_fx_prob: float
    use_reverb_prob: float
    use_limiter_prob: float
    use_compression_prob: float


def _velocity_table() -> np.ndarray:
    """``_vel_to_vol`` (synthetiser.py:204-212) for velocities 0..127, evaluated
    with the reference's own scalar fp32 tensor arithmetic."""
    tab = np.zeros(128, np.float32)
    for v in range(1, 128):
        nv = torch.clamp(torch.tensor(float(v)), 0, 127) / 127.0
        tab[v] = float(0.1 + (1.0 - 0.1) * (6 ** nv - 1) / (6 - 1))
    return tab


_VEL_TABLE = _velocity_table()


def vel_to_vol(velocity: float) -> np.float32:
    if velocity == 0:
        return np.float32(0.0)
    if float(velocity).is_integer() and 0 < velocity <= 127:
        return _VEL_TABLE[int(velocity)]
    nv = torch.clamp(torch.tensor(float(velocity)), 0, 127) / 127.0
    return np.float32(float(0.1 + (1.0 - 0.1) * (6 ** nv - 1) / (6 - 1)))


@dataclass
class MixPlan:
    """Host-side description of one batch for ``adt_mix_render_f32``."""
    notes: np.ndarray           # NOTE_DTYPE [n_notes], clip by clip, grouped by track
    clip_note_off: np.ndarray   # int32 [B + 1]
    clip_len: np.ndarray        # int32 [B]
    clip_gain: np.ndarray       # float32 [B]
    picks: list                 # per clip: {pitch: ((pitch, group, name), (pitch, group, name))}
    mixups: list                # per clip: list of per-note mixup draws (input order)
    fx: Optional[np.ndarray] = None   # FX_DTYPE [B] (flags == 0: clip without FX); None when use_fx_prob == 0
    boards: Optional[list] = None     # per clip: the drawn [(effect, kwargs)] (for inspection / tests)

    @property
    def width(self) -> int:
        return int(self.clip_len.max()) if self.clip_len.size else 0


class SynthDrum:
    def __init__(self, config: SynthDrumConfig, bank: Optional[OneShotBank] = None, device: Optional[str] = None):
        self.config = config
        self.sample_rate = config.sample_rate
        self.oneshot_path = f"{config.oneshot_path}@{self.sample_rate}.npz"
        self.similarity_threshold = config.similarity_threshold
        self.ADTOF_mapping = config.ADTOF_mapping
        if config.use_fx_prob and not 12544 <= config.sample_rate <= 96000:
            raise ValueError("the FX chain needs a sample rate in [12544, 96000] Hz")
        if bank is None:
            if not os.path.exists(self.oneshot_path):
                raise FileNotFoundError(f"one-shot bank {self.oneshot_path} not found (flat .npz; see adt_str_amd/bank.py)")
            bank = OneShotBank.load(self.oneshot_path)
        if bank.sample_rate != self.sample_rate:
            raise ValueError(f"bank is {bank.sample_rate} Hz, config asks for {self.sample_rate} Hz")
        self.bank = bank
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()) \
            if torch.cuda.is_available() else None
        self._thr_groups = self.tolerance_thr_to_h5_group()
        self._stream = None               # side stream for batches with FX (render_plan)

    # ---- reference-named helpers -------------------------------------------------
    def floor_to_tenth(self, x: float) -> float:
        return math.floor(x * 10) / 10

    def tolerance_thr_to_h5_group(self) -> List[str]:
        """Groups from "gold" down to the one containing the threshold (synthetiser.py:171-190)."""
        groups, level = [], 1.0
        floor = self.floor_to_tenth(self.similarity_threshold)
        while level >= floor:
            groups.append(_THR_GROUP[round(level, 1)])
            level -= 0.1
        return groups

    def random_choice_timbre(self, group: int):
        """One timbre pick (synthetiser.py:192-202): optional ADTOF member pitch,
        then a similarity group that exists for it, then a one-shot name."""
        pitch = int(group)
        if self.ADTOF_mapping:
            pitch = int(random.choice(ADTOF_INVERSE_MAPPING[pitch]))
        cache = self.__dict__.setdefault("_timbre_cache", {})
        if pitch not in cache:                                    # the lists the reference rebuilds from the HDF5 file at every pick
            valid = [g for g in self._thr_groups if self.bank.has_cell(pitch, g)]
            cache[pitch] = (valid, {g: self.bank.cell_names(pitch, g) for g in valid})
        valid, names = cache[pitch]
        g = random.choice(valid)
        name = random.choice(names[g])
        return pitch, g, name

    def _vel_to_vol(self, velocity):
        return vel_to_vol(float(velocity))

    # ---- planning (host) -----------------------------------------------------------
    def _clip_length(self, notes32: np.ndarray) -> int:
        """``int(max(max_offset + 0.1, input_sec) * sr)`` in the reference's mixed
        fp32-tensor / Python-float arithmetic (synthetiser.py:262-263,243)."""
        end32 = np.float32(notes32[:, 1].max()) + np.float32(0.1)
        if np.float32(self.config.input_sec) > end32:
            return int(self.config.input_sec * self.config.sample_rate)
        return int(np.float32(end32 * np.float32(self.config.sample_rate)))

    def _pitch_tables(self):
        """Per-pitch lookups that do not depend on the draws: the instrument's mix volume (synthetiser.py:104-113,151-153)."""
        if getattr(self, "_gain_of", None) is None:
            self._gain_of = {p: VOLUME_PER_INSTRUMENT[ADTOF_LABEL[ADTOF_MAPPING[p] if not self.ADTOF_mapping else p]]
                             for p in range(35, 62) if (p in ADTOF_MAPPING if not self.ADTOF_mapping else p in ADTOF_LABEL)}
            self._shot_of: dict = {}
        return self._gain_of, self._shot_of

    def plan(self, batch: Sequence[Sequence[Sequence[float]]]) -> MixPlan:
        """Host half of a batch render.  The only per-note Python work left is what has to stay sequential: the reference's
        draws from ``random`` in its own order (two timbre picks at a pitch's first note ``:274-281``, then one
        ``uniform(0, mixup_range)`` per note ``:217`` -- ``uniform(a, b)`` is ``a + (b - a) * random()``, evaluated below
        for the whole batch at once, and the FX coin after a clip's last note ``:154-155``); every other field of the note
        records is computed with numpy over all notes of the batch."""
        sr = self.config.sample_rate
        sr32, rng_mix = np.float32(sr), float(self.config.mixup_range)
        rnd, pick = random.random, self.random_choice_timbre
        gain_of, shot_of = self._pitch_tables()
        use_fx = float(self.config.use_fx_prob) > 0.0
        B = len(batch)
        arrs = [np.asarray(notes, dtype=np.float32).reshape(-1, 4) for notes in batch]
        counts = np.fromiter((x.shape[0] for x in arrs), np.int64, B)
        offs = np.zeros(B + 1, np.int64)
        np.cumsum(counts, out=offs[1:])
        a = np.concatenate(arrs) if B else np.zeros((0, 4), np.float32)
        pit, vel = a[:, 2], a[:, 3]
        bad = ~((pit >= 35) & (pit <= 61) & (a[:, 1] >= a[:, 0]))
        if bad.any():
            raise ValueError(f"Invalid note: {a[int(np.argmax(bad))]}")
        pl_all = pit.astype(np.int64).tolist()
        u, main_id, sub_id, track, gain = [], [], [], [], []
        picks_all, fx_recs, boards = [], [], []
        for c in range(B):                                         # the sequential part: RNG draws in the reference's order
            lo, hi = int(offs[c]), int(offs[c + 1])
            picks: dict = {}
            if hi == lo:                                           # synthetiser.py:257-258 (returns before any draw)
                picks_all.append(picks); fx_recs.append(np.zeros((), FX_DTYPE)); boards.append([])
                continue
            ids: dict = {}
            for p in pl_all[lo:hi]:
                e = ids.get(p)
                if e is None:
                    m_, s_ = pick(p), pick(p)
                    picks[p] = (m_, s_)
                    for t_ in (m_, s_):
                        if t_ not in shot_of:
                            shot_of[t_] = self.bank.shot_id(*t_)
                    e = ids[p] = (shot_of[m_], shot_of[s_], len(ids), gain_of[p])
                u.append(rnd())
                main_id.append(e[0]); sub_id.append(e[1]); track.append(e[2]); gain.append(e[3])
            picks_all.append(picks)
            board = []
            if rnd() < self.config.use_fx_prob:
                board = draw_board(self.config.use_reverb_prob, self.config.use_compression_prob, self.config.use_limiter_prob)
            boards.append(board)
            fx_recs.append(board_to_record(board))
        n = a.shape[0]
        mixups = (0.0 + (rng_mix - 0.0) * np.asarray(u, np.float64))             # random.uniform(0, mixup_range)
        rec = np.zeros(n, NOTE_DTYPE)
        rec["start"] = (a[:, 0] * sr32).astype(np.int32)                          # int(onset * sr) in fp32 (synthetiser.py:229)
        rec["main_shot"], rec["sub_shot"], rec["track"], rec["track_gain"] = main_id, sub_id, track, gain
        rec["one_minus_mixup"] = (1 - mixups).astype(np.float32)
        rec["mixup"] = mixups.astype(np.float32)
        rec["vol"] = self._vols(vel)
        clip_of = np.repeat(np.arange(B, dtype=np.int64), counts)
        rec = rec[np.argsort(clip_of * 64 + rec["track"], kind="stable")]         # grouped by track inside each clip (< 64 tracks: 27 pitches)
        lens = np.full(B, int(self.config.input_sec * sr), np.int64)
        gains = np.zeros(B, np.float32)
        live = np.nonzero(counts)[0]
        if live.size:
            # int(max(max_offset + 0.1, input_sec) * sr) in the reference's mixed fp32-tensor / Python-float arithmetic (:262-263,243)
            end32 = np.maximum.reduceat(a[:, 1], offs[live]).astype(np.float32) + np.float32(0.1)
            longer = ~(np.float32(self.config.input_sec) > end32)
            lens[live[longer]] = (end32[longer] * sr32).astype(np.float32).astype(np.int64)
            gains[live] = self._vols(np.maximum(np.float32(0), np.maximum.reduceat(vel, offs[live])))
        mix_all = [mixups[int(offs[c]):int(offs[c + 1])].tolist() for c in range(B)]
        fx = np.stack(fx_recs).astype(FX_DTYPE) if (use_fx and fx_recs) else None
        return MixPlan(notes=rec, clip_note_off=offs.astype(np.int32), clip_len=lens.astype(np.int32),
                       clip_gain=gains, picks=picks_all, mixups=mix_all, fx=fx, boards=boards)

    @staticmethod
    def _vols(vel: np.ndarray) -> np.ndarray:
        """``_vel_to_vol`` (synthetiser.py:204-212) over an array: the table for whole velocities, the scalar law otherwise."""
        vel = np.asarray(vel, np.float32)
        whole = (vel == np.floor(vel)) & (vel > 0) & (vel <= 127)
        vol = np.zeros(vel.shape, np.float32)
        vol[whole] = _VEL_TABLE[vel[whole].astype(np.int64)]
        for i in np.nonzero(~whole & (vel != 0))[0]:
            vol[i] = vel_to_vol(float(vel[i]))
        return vol

    # ---- rendering (GPU) -----------------------------------------------------------
    def render_plan(self, plan: MixPlan, width: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Render a planned batch.  A batch with FX clips and no caller-provided buffer is rendered on this object's own HIP stream
        and the caller's stream only waits for the result: the FX kernels are latency-bound chains on a handful of waves
        (6.5 ms for a 64 x 10 s batch), and issued this way they run underneath whatever the caller's stream still has queued --
        in a training loop, the previous step."""
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("SynthDrum renders on the GPU only (there is no CPU path)")
        if out is None and plan.fx is not None and bool(plan.fx["flags"].any()) and len(plan.clip_len) and (width or plan.width):
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=self.device)
            main = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self._stream):
                res = self._render_plan(plan, width, None)
            main.wait_stream(self._stream)
            res.record_stream(main)
            return res
        return self._render_plan(plan, width, out)

    def _render_plan(self, plan: MixPlan, width: Optional[int], out: Optional[torch.Tensor]) -> torch.Tensor:
        dev = self.device
        B = len(plan.clip_len)
        width = plan.width if width is None else width
        if out is None:
            out = torch.empty((B, width), dtype=torch.float32, device=dev)
        if B == 0 or width == 0:
            return out
        data, offsets = self.bank.device_arrays(dev)
        n_notes = int(plan.notes.shape[0])
        notes_d = torch.from_numpy(plan.notes.view(np.uint8).reshape(-1)).to(dev, non_blocking=True) if n_notes else \
            torch.zeros(32, dtype=torch.uint8, device=dev)
        off_d = torch.from_numpy(plan.clip_note_off).to(dev, non_blocking=True)
        len_d = torch.from_numpy(plan.clip_len).to(dev, non_blocking=True)
        gain_d = torch.from_numpy(plan.clip_gain).to(dev, non_blocking=True)
        ws_bytes = _ffi.load().adt_mix_workspace_bytes(n_notes, B)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        if plan.fx is not None and bool(plan.fx["flags"].any()):
            fx_d = torch.from_numpy(plan.fx.view(np.uint8).reshape(-1).copy()).to(dev, non_blocking=True)
            _ffi.call("adt_mix_render_fx_f32", _ffi.dptr(data), _ffi.dptr(offsets), self.bank.n_shots, _ffi.dptr(notes_d),
                      n_notes, _ffi.dptr(off_d), _ffi.dptr(len_d), _ffi.dptr(gain_d), B, width, _ffi.dptr(fx_d), self.sample_rate,
                      _ffi.dptr(out), out.stride(0), _ffi.dptr(ws), ws_bytes, _ffi.current_stream())
        else:
            _ffi.call("adt_mix_render_f32", _ffi.dptr(data), _ffi.dptr(offsets), self.bank.n_shots, _ffi.dptr(notes_d),
                      n_notes, _ffi.dptr(off_d), _ffi.dptr(len_d), _ffi.dptr(gain_d), B, width, _ffi.dptr(out),
                      out.stride(0), _ffi.dptr(ws), ws_bytes, _ffi.current_stream())
        return out

    def render_batch(self, batch: Sequence[Sequence[Sequence[float]]], width: Optional[int] = None):
        """-> (wavs[B, width or max W] on the GPU, zero-padded like collate_fn; lengths[B])."""
        plan = self.plan(batch)
        return self.render_plan(plan, width), torch.from_numpy(plan.clip_len.astype(np.int64))

    def __call__(self, notes, eval_rendering: bool = False) -> torch.Tensor:
        if eval_rendering:
            raise NotImplementedError("eval_rendering relies on default_timbre_path, which the reference never defines")
        wavs, lens = self.render_batch([list(notes)])
        return wavs[0, : int(lens[0])]
