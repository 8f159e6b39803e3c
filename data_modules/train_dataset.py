"""Drop-in import path of the reference's ``data_modules/train_dataset.py`` (Lakh path only)."""
import glob
import os
from dataclasses import dataclass
from typing import List, Optional

from adt_str_amd.data import GpuBatcher, NoteChunkDataset, collate_fn, notes_from_bytes  # noqa: F401


@dataclass
class LakhDatasetConfig:
    input_sec: float
    time_res: float
    win_length: int
    sample_rate: int
    dataset_path: str
    empty_tokens_percentage: float
    random_velocity_prob: float
    dataset_name: str
    partitions: Optional[List[str]] = None


class LakhDataset(NoteChunkDataset):
    """Note chunks from the Lakh parquet shards ``<dataset_path>/<A..Z>.parquet`` (column ``notes`` =
    float32 [N, 4] bytes).  ``__getitem__`` returns ``(notes, tokens)`` -- the clip itself is rendered
    per *batch* on the GPU by ``GpuBatcher.batch`` (the reference renders per item on the CPU,
    train_dataset.py:213-229)."""

    def __init__(self, config: LakhDatasetConfig, tokenizer, synthetiser):
        import pyarrow.parquet as pq
        parts = config.partitions or [chr(c) for c in range(ord("A"), ord("Z") + 1)]
        files = [f for f in (os.path.join(config.dataset_path, f"{p}.parquet") for p in parts) if os.path.exists(f)]
        if not files:
            raise FileNotFoundError(f"no parquet shards under {config.dataset_path}")
        rows = []
        for f in files:
            rows.extend(pq.read_table(f, columns=["notes"]).column("notes").to_pylist())
        super().__init__(rows, GpuBatcher(tokenizer, synthetiser, config.empty_tokens_percentage, config.random_velocity_prob))
