"""Drop-in for the reference's ``data_modules/augment_data_with_CLAP.py``:
``python data_modules/augment_data_with_CLAP.py <config.yaml> [--num_bins 10]``.

Same inputs and outputs: every ``*.wav`` under ``clap_config.reference_root/<GM pitch>/`` defines the class means, every
``*.wav`` under ``clap_config.sample_pack_root`` is embedded, scored against the means and copied once into
``<reference_root>_clap_augmented/<class>/<upper>-<lower>/`` in descending score order (reference lines 84-199).
What moved to the GPU: the feature extractor (K9), the HTSAT tower (K10/K11 + GEMMs), the cosine arg-max (K12); the
O(N x C) Python tuple list + sort of the reference becomes a sort of N items with the same order (adt_str_amd/curation.py).
Under ``torchrun`` every rank embeds a strided slice of the files and the embeddings are all-gathered; rank 0 copies.

Files at other rates are resampled on the GPU (K13, the reference's ``torchaudio.transforms.Resample`` at :55-58)."""
import argparse
import os
import shutil
import sys
from glob import glob
from pathlib import Path

sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from adt_str_amd.audio_io import read_wav  # noqa: E402
from adt_str_amd.config_utils import load_merged  # noqa: E402
from adt_str_amd.curation import assign, class_mean_embeddings  # noqa: E402
from config import ClapConfig  # noqa: E402
from modules.clap_encoder import ClapWrapper  # noqa: E402


def sort_paths_by_parent_folder(file_paths):
    """Numeric parent folders first (by value), then the others by name; file name breaks ties (reference :38-48)."""
    def sort_key(path):
        parent = Path(path).parent.name
        try:
            return (0, int(parent), Path(path).name.lower())
        except ValueError:
            return (1, parent, Path(path).name.lower())
    return sorted(file_paths, key=sort_key)


_RESAMPLERS = {}


def load_audio(path, target_sample_rate: int) -> torch.Tensor:
    """[1, L] mono fp32 at the target rate (reference :51-59)."""
    audio, sr = read_wav(path)
    waveform = torch.from_numpy(audio.mean(axis=0, keepdims=True))
    if sr != target_sample_rate:
        from adt_str_amd.resample import Resample
        if (sr, target_sample_rate) not in _RESAMPLERS:
            _RESAMPLERS[(sr, target_sample_rate)] = Resample(sr, target_sample_rate)
        waveform = _RESAMPLERS[(sr, target_sample_rate)](waveform.cuda()).cpu()
    return waveform


def normalize(waveform: torch.Tensor) -> torch.Tensor:
    return waveform / torch.max(torch.abs(waveform))


def _load_normalized(path, sample_rate):
    return normalize(load_audio(path, sample_rate))


def _load_batch(paths, sample_rate, device, decoded):
    """The batch's clips as the reference prepares them (load -> mono -> resample -> x / max|x|, :51-68), [1, L] each on
    ``device``: one batched decode (``decoded``, made by the I/O thread), one copy across PCIe, resampling (K13) and the
    normalisation batched on the GPU (adt_str_amd.audio_io.load_clips_batch: the per-file arithmetic, bitwise).  A file that
    cannot be decoded, or is empty, takes the per-file route, which raises what the reference would."""
    from adt_str_amd.audio_io import load_clips_batch
    clips, _, _ = load_clips_batch(paths, sample_rate, device, normalize=True, decoded=decoded)
    return [c[None] if c is not None else _load_normalized(paths[j], sample_rate).to(device) for j, c in enumerate(clips)]


def _embed(wrapper, files, batch_size, sample_rate):
    """Embeddings [len(files), 512] on the wrapper's device, this rank's strided slice computed here, the rest gathered.

    The reference reads, resamples and normalises one file at a time on the thread that also drives the GPU
    (augment_data_with_CLAP.py:66-68,124-137).  Here batch i + 1 is decoded on the library's thread pool (one call per batch,
    off the GIL) while the GPU embeds batch i; the order of the results is the order of ``files``."""
    from concurrent.futures import ThreadPoolExecutor
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = list(range(rank, len(files), world))
    chunks = []
    starts = list(range(0, len(mine), batch_size))
    device = torch.device(wrapper.device)
    from adt_str_amd.audio_io import read_wav_batch
    with ThreadPoolExecutor(max_workers=1, thread_name_prefix="adt-curation-io") as pool:
        names = lambda i: [files[j] for j in mine[i:i + batch_size]]
        submit = lambda i: pool.submit(read_wav_batch, names(i), False, device.type == "cuda")
        pending = submit(starts[0]) if starts else None
        for k, i in enumerate(starts):
            decoded = pending.result()
            pending = submit(starts[k + 1]) if k + 1 < len(starts) else None      # decode the next batch while this one is embedded
            batch = _load_batch(names(i), sample_rate, device, decoded)
            chunks.append(wrapper.get_audio_features(batch).float())
    local = torch.cat(chunks) if chunks else torch.zeros((0, 512), device=wrapper.device)
    if world == 1:
        return local
    per = -(-len(files) // world)
    padded = torch.zeros((per, local.shape[1]), device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    out = torch.empty((len(files), local.shape[1]), device=local.device)
    for r in range(world):
        idx = list(range(r, len(files), world))
        out[idx] = parts[r][: len(idx)]
    return out


PHASE_SECONDS = {}     # wall time of the last run()'s phases: embed_references, embed_packs, assign, copy (read by tools/e2e.py)


def run(cfg: dict, num_bins: int = 10, clap_model=None, copy: bool = True):
    """The reference's ``__main__`` body.  ``clap_model``: an already built ``transformers.ClapModel`` (offline use).
    Returns ``(assignment, wav_files, augmented_root)``."""
    import time
    PHASE_SECONDS.clear()
    t_phase = time.perf_counter()

    def phase(name):
        nonlocal t_phase
        torch.cuda.synchronize()
        now = time.perf_counter()
        PHASE_SECONDS[name] = now - t_phase
        t_phase = now
    if num_bins <= 0 or 100 % num_bins != 0:
        raise ValueError("--num_bins must be a positive integer that divides 100 evenly")
    section = dict(cfg["clap_config"])
    section.update(cfg["shared"])
    c = ClapConfig(**section)
    device = torch.device("cuda", torch.cuda.current_device())
    wrapper = ClapWrapper(device=device, model_name=c.model_name, sample_rate=c.sample_rate, clap_model=clap_model)
    wav_files = glob(f"{c.sample_pack_root}/**/*.[Ww][Aa][Vv]", recursive=True)
    print(f"Total: {len(wav_files)}")
    reference_files = sort_paths_by_parent_folder(glob(f"{c.reference_root}/**/*.[Ww][Aa][Vv]", recursive=True))
    print(f"Total: {len(reference_files)}")

    reference_dict = {k: [] for k in range(35, 82)}
    reference_dict.update({421: []})                      # electric hi-hat
    for file, emb in zip(reference_files, _embed(wrapper, reference_files, c.batch_size, c.sample_rate)):
        reference_dict[int(Path(file).parent.name)].append(emb)
    labels, reference_embeddings = class_mean_embeddings(reference_dict)
    print(f"Reference embeddings shape: {tuple(reference_embeddings.shape)}")
    phase("setup_and_embed_references")
    sample_pack_embeddings = _embed(wrapper, wav_files, c.batch_size, c.sample_rate)
    print(f"Sample pack embeddings shape: {tuple(sample_pack_embeddings.shape)}")
    phase("embed_packs")
    res = assign(sample_pack_embeddings, reference_embeddings, labels, num_bins)
    phase("assign")

    augmented_root = Path(f"{c.reference_root}_clap_augmented")
    if copy and (not dist.is_initialized() or dist.get_rank() == 0):
        if augmented_root.exists():
            shutil.rmtree(augmented_root)
        augmented_root.mkdir(parents=True, exist_ok=True)
        # destination -> source in the reference's copy order (descending score; a later copy onto the same destination
        # overwrites an earlier one, :184-196), then the copies themselves in parallel: every destination is written once
        plan = {}
        root_s = str(augmented_root)
        for i, label, bin_label in zip(res.order.tolist(), res.label.tolist(), res.bin):
            plan[os.path.join(root_s, str(label), bin_label, os.path.basename(wav_files[i]))] = wav_files[i]
        for d in {os.path.dirname(dst) for dst in plan}:
            os.makedirs(d, exist_ok=True)
        from adt_str_amd.audio_io import copy_files
        dsts, srcs = list(plan), list(plan.values())
        status = copy_files(srcs, dsts)                    # shutil.copy2 per pair (contents, mode, times) on the library's thread pool
        for j in np.nonzero(status)[0]:                    # the reference logs and carries on (:188-189)
            print(f"Failed to copy {srcs[j]} -> {dsts[j]}: status {int(status[j])} (adt_copy_files)")
        # the reference counts one copy per source path (:185-187), including a copy that overwrites an earlier file of the same name in
        # the same <class>/<bin>/ (the plan above keeps the last writer of a destination, as the sequential loop would leave it)
        copied = len(res.order) - int((status != 0).sum())
        print(f"Copied: {copied}, Skipped (duplicates): {len(wav_files) * (len(labels) - 1)}")
    phase("copy")
    return res, wav_files, augmented_root


parser = argparse.ArgumentParser()
parser.add_argument("config_path", type=str, help="Path to the config file")
parser.add_argument("--num_bins", type=int, default=10, help="Number of bins for discretization (must evenly divide 100)")
if __name__ == "__main__":
    args = parser.parse_args()
    if args.num_bins <= 0 or 100 % args.num_bins != 0:
        parser.error("--num_bins must be a positive integer that divides 100 evenly")
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    run(load_merged(args.config_path), args.num_bins)
    if dist.is_initialized():
        dist.destroy_process_group()
