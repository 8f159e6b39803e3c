"""Minimal audio / MIDI file I/O for the inference CLI (the reference leans on torchaudio.load and
pretty_midi, ``inference.py:14-32,82-93``; neither is available on the GPU box).

  read_wav   RIFF/WAVE PCM 8/16/24/32-bit and IEEE float32 -> float32 [channels, samples] in [-1, 1]
  write_wav  float32 -> 16-bit PCM
  write_drum_midi  notes [[onset s, offset s, pitch, velocity]...] -> Standard MIDI File, format 0,
             channel 10 (percussion), 480 ticks per quarter at 120 bpm (pretty_midi's defaults)
"""
from __future__ import annotations

import struct
from typing import Sequence, Tuple

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
