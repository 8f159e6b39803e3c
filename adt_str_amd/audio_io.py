"""Minimal audio / MIDI file I/O for the inference CLI (the reference leans on torchaudio.load and
pretty_midi, ``inference.py:14-32,82-93``; neither is available on the GPU box).

  read_wav   RIFF/WAVE PCM 8/16/24/32-bit and IEEE float32 -> float32 [channels, samples] in [-1, 1]
  read_wav_batch / copy_files   the curation path's batched host I/O (libadt_hip.so: adt_wav_*_batch, adt_copy_files)
  write_wav  float32 -> 16-bit PCM
  write_drum_midi  notes [[onset s, offset s, pitch, velocity]...] -> Standard MIDI File, format 0,
             channel 10 (percussion), 480 ticks per quarter at 120 bpm (pretty_midi's defaults)
"""
from __future__ import annotations

import ctypes
import os
import struct
from typing import NamedTuple, Optional, Sequence, Tuple

import numpy as np


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:          # WAVE_FORMAT_EXTENSIBLE: real tag in the sub-format GUID
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif tag == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, channels, rate, _, _, bits = fmt
    if tag == 3 and bits == 32:
        x = np.frombuffer(pcm, "<f4").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(pcm, "<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(pcm, "<i4").astype(np.float32) / 2147483648.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm[: len(pcm) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
    elif tag == 1 and bits == 8:
        x = (np.frombuffer(pcm, np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported WAV encoding (format tag {tag}, {bits} bits)")
    x = x[: len(x) // channels * channels].reshape(-1, channels).T
    return np.ascontiguousarray(x), int(rate)


def write_wav(path: str, wav: np.ndarray, sample_rate: int) -> None:
    x = np.atleast_2d(np.asarray(wav, np.float32))
    pcm = (np.clip(x, -1.0, 1.0).T * 32767.0).round().astype("<i2").tobytes()
    ch = x.shape[0]
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, sample_rate,
                                                                               sample_rate * ch * 2, ch * 2, 16)
    with open(path, "wb") as fh:
        fh.write(hdr + b"data" + struct.pack("<I", len(pcm)) + pcm)


def _vlq(n: int) -> bytes:
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def write_drum_midi(path: str, notes: Sequence[Sequence[float]], ticks_per_quarter: int = 480, tempo_us: int = 500000) -> None:
    tick = lambda sec: int(round(sec * 1e6 / tempo_us * ticks_per_quarter))
    events = []
    for onset, offset, pitch, vel in notes:
        p, v = int(pitch) & 0x7F, max(1, min(127, int(vel)))
        events.append((tick(float(onset)), 1, bytes([0x99, p, v])))
        events.append((max(tick(float(offset)), tick(float(onset)) + 1), 0, bytes([0x89, p, 0])))
    events.sort(key=lambda e: (e[0], e[1]))
    track = b"\x00\xff\x51\x03" + tempo_us.to_bytes(3, "big")
    now = 0
    for t, _, msg in events:
        track += _vlq(t - now) + msg
        now = t
    track += b"\x00\xff\x2f\x00"
    with open(path, "wb") as fh:
        fh.write(b"MThd" + struct.pack(">IHHH", 6, 0, 1, ticks_per_quarter) + b"MTrk" + struct.pack(">I", len(track)) + track)


# ---------------------------------------------------------------------------------------------- batched host I/O (H1)
class _WavInfo(ctypes.Structure):          # include/adt_hip.h: adt_wav_info
    _fields_ = [("frames", ctypes.c_int64), ("data_offset", ctypes.c_int64), ("data_bytes", ctypes.c_int64),
                ("sample_rate", ctypes.c_int32), ("channels", ctypes.c_int32), ("format", ctypes.c_int32), ("bits", ctypes.c_int32),
                ("status", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class WavBatch(NamedTuple):
    data: "object"            # torch float32 [total samples] on the host (pinned when asked): the clips back to back, mono
    offsets: np.ndarray       # int64 [n + 1]: clip i is data[offsets[i]:offsets[i + 1]] (empty for a file that failed)
    sample_rate: np.ndarray   # int32 [n]
    peak: np.ndarray          # float32 [n]: max |x| of the mono clip BEFORE normalisation
    status: np.ndarray        # int32 [n]: 0, or the ADT_E* code of a file that could not be decoded


def io_threads() -> int:
    """Pool size of the batched calls: ADT_CURATION_IO_THREADS, default the CPUs this process may run on (at most 32)."""
    env = os.environ.get("ADT_CURATION_IO_THREADS")
    if env:
        return max(1, int(env))
    try:
        return max(1, min(32, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(32, os.cpu_count() or 1))


def _c_paths(paths: Sequence[str]):
    arr = (ctypes.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
    return arr, ctypes.cast(arr, ctypes.c_void_p)


def read_wav_batch(paths: Sequence[str], normalize: bool = False, pin: bool = False, threads: Optional[int] = None) -> WavBatch:
    """Decode ``paths`` on the library's thread pool: every file as ``read_wav(path)[0].mean(axis=0)`` (bitwise), optionally
    divided by its peak -- the reference's per-file ``torchaudio.load`` -> ``mean(dim=0)`` -> ``x / x.abs().max()``
    (data_modules/augment_data_with_CLAP.py:51-68) for a whole batch in two calls (headers, then samples)."""
    import torch
    from . import _ffi
    n = len(paths)
    threads = io_threads() if threads is None else threads
    info = (_WavInfo * max(n, 1))()
    keep, pp = _c_paths(paths)
    _ffi.call("adt_wav_probe_batch", pp, n, threads, ctypes.cast(info, ctypes.c_void_p))
    view = np.frombuffer(info, dtype=np.dtype({"names": [f[0] for f in _WavInfo._fields_],
                                               "formats": [np.int64] * 3 + [np.int32] * 6}), count=max(n, 1))[:n]
    frames = np.where(view["status"] == 0, view["frames"], 0).astype(np.int64)
    offsets = np.zeros(n + 1, np.int64)
    np.cumsum(frames, out=offsets[1:])
    total = int(offsets[-1])
    data = torch.empty(total, dtype=torch.float32, pin_memory=bool(pin and total and torch.cuda.is_available()))
    peak = np.zeros(n, np.float32)
    if n:
        _ffi.call("adt_wav_decode_batch", pp, n, threads, ctypes.cast(info, ctypes.c_void_p), offsets.ctypes.data, 1 if normalize else 0,
                  data.data_ptr(), peak.ctypes.data)
    del keep
    status = view["status"].copy()
    if (status != 0).any() and (frames[status != 0] != 0).any():      # a file that failed between the probe and the read: its slot is undefined
        for i in np.nonzero((status != 0) & (frames != 0))[0]:
            data[offsets[i]:offsets[i + 1]] = float("nan")
    return WavBatch(data, offsets, view["sample_rate"].copy(), peak, status)


def copy_files(srcs: Sequence[str], dsts: Sequence[str], threads: Optional[int] = None) -> np.ndarray:
    """``shutil.copy2(src, dst)`` for every pair on the library's thread pool (contents, mode bits, times; no extended
    attributes).  Returns int32 status per pair (0 = copied)."""
    from . import _ffi
    assert len(srcs) == len(dsts)
    n = len(srcs)
    status = np.zeros(n, np.int32)
    if n:
        k1, p1 = _c_paths(srcs)
        k2, p2 = _c_paths(dsts)
        _ffi.call("adt_copy_files", p1, p2, n, io_threads() if threads is None else threads, status.ctypes.data)
        del k1, k2
    return status


_RESAMPLERS: dict = {}
_PAD_BUDGET = 1 << 26            # elements of one padded [clips, longest] matrix handed to the resampler (256 MB of fp32)


def load_clips_batch(paths: Sequence[str], sample_rate: int, device, normalize: bool = True, threads: Optional[int] = None,
                     decoded: Optional[WavBatch] = None):
    """The reference's per-file ``load -> mono -> Resample(sr, sample_rate) -> x / max|x|`` (data_modules/augment_data_with_CLAP.py:
    51-68; convert_augmented_to_hdf5.py:97-103) for a batch: one batched decode into pinned memory, ONE host-to-device copy, the
    resampler (K13) over zero-padded groups of equal source rate, peaks and the division on the GPU.  Bitwise the per-file result:
    a clip's zero padding is what the resampler assumes past its end anyway, max|x| is exact and fp32 division is IEEE on both sides.

    Returns ``(clips, peaks, status)``: ``clips[i]`` a 1-D fp32 tensor on ``device`` (None when ``status[i] != 0`` or the file is
    empty), ``peaks[i]`` the clip's max|x| before the division (after resampling), ``status`` as in `read_wav_batch`.
    ``decoded``: the files' ``read_wav_batch(paths, normalize=False, pin=True)`` when a worker thread has already produced it."""
    import torch
    from .resample import Resample
    device = torch.device(device)

    def up(a):
        """Small host array -> device without stalling: a copy from pageable memory is stream-ordered AND blocks the host, i.e. it waits
        for everything the GPU still has queued (the previous batch's tower); through pinned memory it is just enqueued."""
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.pin_memory().to(device, non_blocking=True) if device.type == "cuda" else t.to(device)

    b = decoded if decoded is not None else read_wav_batch(paths, normalize=False, pin=device.type == "cuda", threads=threads)
    n = len(paths)
    lens = np.diff(b.offsets)
    ok = (b.status == 0) & (lens > 0)
    dev = b.data.to(device, non_blocking=True)
    off = b.offsets.tolist()
    clips: list = [None] * n
    peaks = torch.zeros(n, dtype=torch.float32, device=device)
    same = ok & (b.sample_rate == sample_rate)
    if same.any():
        # peaks of the clips already at the target rate came with the decode; divide the whole buffer once (other clips: by 1)
        pk = up(np.where(same, b.peak, np.float32(1.0)).astype(np.float32))
        peaks = torch.where(up(same), pk, peaks)
        if normalize:
            dev = dev / torch.repeat_interleave(pk, up(lens), output_size=int(b.offsets[-1]))
        for i in np.nonzero(same)[0]:
            clips[i] = dev[off[i]:off[i + 1]]
    for sr in sorted(set(b.sample_rate[ok & ~same].tolist())):
        key = (int(sr), int(sample_rate))
        if key not in _RESAMPLERS:
            _RESAMPLERS[key] = Resample(*key)
        rs = _RESAMPLERS[key]
        idx = np.nonzero(ok & (b.sample_rate == sr))[0]
        idx = idx[np.argsort(lens[idx], kind="stable")]              # neighbours in length share a padded matrix
        lo = 0
        while lo < len(idx):
            hi = lo + 1
            while hi < len(idx) and (hi + 1 - lo) * int(lens[idx[hi]]) <= _PAD_BUDGET:
                hi += 1
            part = idx[lo:hi]
            ln = up(lens[part])
            width = int(lens[part[-1]])
            # padded[r, t] = clip r's sample t, zero past its end: a gather by computed index (a boolean-mask assignment would run
            # nonzero() and wait for the GPU)
            col = torch.arange(width, device=device)[None, :]
            src_idx = (up(np.asarray(b.offsets)[part].astype(np.int64))[:, None] + col).clamp_(max=max(int(b.offsets[-1]) - 1, 0))
            padded = torch.where(col < ln[:, None], dev[src_idx], torch.zeros((), dtype=torch.float32, device=device))
            del src_idx
            out = rs(padded)
            out_len_h = -((-lens[part] * rs.new) // rs.orig)                                   # ceil(new * L / orig) per clip
            out_len = up(out_len_h)
            valid = torch.arange(out.shape[1], device=device)[None, :] < out_len[:, None]
            pk = torch.where(valid, out.abs(), torch.zeros((), device=device)).amax(dim=1)
            pk = torch.where((out != out).logical_and(valid).any(dim=1), torch.full_like(pk, float("nan")), pk)
            if normalize:
                out = out / pk[:, None]
            peaks[up(part)] = pk
            for r, i in enumerate(part):
                clips[i] = out[r, :int(out_len_h[r])]
            lo = hi
    return clips, peaks, b.status
