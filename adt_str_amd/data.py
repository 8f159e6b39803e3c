"""Batch assembly: drop-in for the reference's ``collate_fn`` (``data_modules/train_dataset.py:41-56``)
and the per-item logic of ``LakhDataset.__getitem__`` (``:213-229``), re-cut for a GPU-side renderer:
items carry *notes*, and the whole batch is rendered by one mixer call instead of one Python render per
item inside DataLoader workers.

The reference hides its per-item host work behind up to 16 DataLoader worker processes
(``train.py:235-238``).  Here the host work per batch is a few milliseconds of numpy (``GpuBatcher.item``,
``host_batch``, ``SynthDrum.plan``) and runs ONE batch ahead of the GPU on a background thread
(``Prefetcher``): the draws from ``random`` / ``torch`` keep the order they have in a plain loop, because
one thread makes all of them, batch after batch.
"""
from __future__ import annotations

import queue
import random
import threading
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

PAD_TOKEN = 1
_NO_NOTES = np.zeros((0, 4), np.float32)


def collate_fn(batch):
    """list of (wav[W_i], tokens[T_i]) -> dict(wavs [B, Wmax] zero padded, tokens [B, Tmax] PAD padded,
    token_lengths [B]) with the reference's rule that lengths equal to the batch maximum are
    decremented by one (train_dataset.py:46-51; the lengths index tokens[:, :-1])."""
    wavs = [item[0] for item in batch]
    lengths = [len(item[1]) for item in batch]
    tokens = [torch.as_tensor(item[1]) for item in batch]
    top = max(lengths) if lengths else 0
    if top > 0:
        lengths = [n - 1 if n == top else n for n in lengths]
    return {"wavs": pad_sequence(wavs, batch_first=True, padding_value=0.0),
            "tokens": pad_sequence(tokens, batch_first=True, padding_value=PAD_TOKEN).long(),
            "token_lengths": torch.tensor(lengths).long()}


def notes_from_bytes(blob: bytes) -> torch.Tensor:
    """Row schema of the Lakh parquet shards: ``notes`` = float32 [N, 4] bytes (data_modules/midi_parser.py:57-63)."""
    return torch.from_numpy(np.frombuffer(blob, dtype=np.float32).copy()).reshape(-1, 4)


@dataclass
class HostBatch:
    """Everything about a batch the host decides, before any GPU work: the mixer plan and the padded token matrix."""
    plan: object                    # synth.MixPlan
    tokens: np.ndarray              # int64 [B, Tmax], PAD padded
    token_lengths: np.ndarray       # int64 [B], collate's max -> max - 1 rule applied
    rng_state: Optional[tuple] = None   # (random.getstate(), torch.get_rng_state()) right after this batch's draws


class GpuBatcher:
    """Turns note chunks into training batches on the GPU.

    ``item(notes)`` reproduces ``LakhDataset.__getitem__`` (train_dataset.py:213-229) up to the render:
    with probability ``empty_tokens_percentage`` an empty item ([BOS, SIL, EOS], silent clip), else GM->custom
    pitch mapping, optional random velocities, tokenisation.  ``host_batch(items)`` plans the mixer call and pads the
    tokens like ``collate_fn`` (numpy only); ``upload`` renders all clips in one mixer call.  ``batch`` = both, with
    the dict ``collate_fn`` returns."""

    def __init__(self, tokenizer, synthetiser, empty_tokens_percentage: float = 0.05, random_velocity_prob: float = 0.5):
        self.tokenizer, self.synth = tokenizer, synthetiser
        self.empty_p, self.rand_vel_p = empty_tokens_percentage, random_velocity_prob

    def item(self, notes: torch.Tensor):
        """-> (notes float32 [n, 4] ndarray (custom pitches; empty for a silent item), tokens tensor)."""
        if random.random() < self.empty_p:
            return _NO_NOTES, self.tokenizer.empty_adt_tokens()
        notes = self.tokenizer.map_notes_to_Gm_custom(notes.clone(), random_velocity=random.random() < self.rand_vel_p)
        return notes.numpy(), self.tokenizer.notes_to_adt_tokens(notes)

    def host_batch(self, items: Sequence, snapshot_rng: bool = False) -> HostBatch:
        plan = self.synth.plan([it[0] for it in items])
        toks = [np.asarray(it[1]).astype(np.int64, copy=False) for it in items]      # .long(): truncation, like collate_fn
        lens = np.fromiter((t.shape[0] for t in toks), np.int64, len(toks))
        top = int(lens.max()) if len(toks) else 0
        mat = np.full((len(toks), top), PAD_TOKEN, np.int64)
        for i, t in enumerate(toks):
            mat[i, :t.shape[0]] = t
        if top > 0:
            lens = np.where(lens == top, lens - 1, lens)
        state = (random.getstate(), torch.get_rng_state()) if snapshot_rng else None
        return HostBatch(plan, mat, lens, state)

    def upload(self, hb: HostBatch, width: Optional[int] = None, device_tokens: bool = False):
        """Render the planned clips (one mixer call on the current stream) -> the ``collate_fn`` dict with ``wavs`` on the GPU;
        ``device_tokens``: tokens / lengths as GPU tensors too (what the native loop feeds the step)."""
        wavs = self.synth.render_plan(hb.plan, width)
        tokens, lengths = torch.from_numpy(hb.tokens), torch.from_numpy(hb.token_lengths)
        if device_tokens:
            tokens, lengths = tokens.to(wavs.device, non_blocking=True), lengths.to(wavs.device, non_blocking=True)
        return {"wavs": wavs, "tokens": tokens, "token_lengths": lengths}

    def batch(self, items: Sequence):
        return self.upload(self.host_batch(items))


class NoteChunkDataset(torch.utils.data.Dataset):
    """Note chunks (``float32 [N, 4]`` byte blobs, the parquet row schema) + a ``GpuBatcher``: ``__getitem__`` returns
    ``(notes, tokens)`` -- the clip itself is rendered per *batch* on the GPU by ``collate`` (the reference renders per item on
    the CPU, train_dataset.py:213-229)."""

    def __init__(self, rows: List[bytes], batcher: GpuBatcher):
        self.rows, self.batcher = rows, batcher

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, index):
        return self.batcher.item(notes_from_bytes(self.rows[index]))

    def collate(self, items):
        return self.batcher.batch(items)

    def host_batch(self, indices: Sequence[int], snapshot_rng: bool = False) -> HostBatch:
        return self.batcher.host_batch([self[i] for i in indices], snapshot_rng)


class Prefetcher:
    """Runs ``make(i)`` for i = start .. n - 1 on a background thread, at most ``depth`` results ahead of the consumer.

    Iterating yields the results in order; an exception in ``make`` is re-raised at the consumer.  With ``depth = 0`` nothing is
    threaded (``make`` runs inline): the two modes produce the same sequence, since all calls happen in index order on one thread."""

    _END = object()

    def __init__(self, make: Callable[[int], object], n: int, start: int = 0, depth: int = 2):
        self.make, self.n, self.start, self.depth = make, n, start, depth
        self._q: "queue.Queue" = queue.Queue(maxsize=max(depth, 1))
        self._stop = threading.Event()
        self._thread = None
        if depth > 0:
            self._thread = threading.Thread(target=self._run, name="adt-prefetch", daemon=True)
            self._thread.start()

    def _run(self):
        try:
            for i in range(self.start, self.n):
                item = self.make(i)
                while not self._stop.is_set():
                    try:
                        self._q.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        continue
                if self._stop.is_set():
                    return
            self._q.put(self._END)
        except BaseException as e:                       # hand the failure to the consumer
            self._q.put(e)

    def __iter__(self):
        if self._thread is None:
            for i in range(self.start, self.n):
                yield self.make(i)
            return
        while True:
            item = self._q.get()
            if item is self._END:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def close(self):
        self._stop.set()
        if self._thread is not None:
            while self._thread.is_alive():               # drain so a producer blocked in put() can leave
                try:
                    self._q.get_nowait()
                except queue.Empty:
                    pass
                self._thread.join(timeout=0.05)
