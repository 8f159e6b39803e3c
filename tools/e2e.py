#!/usr/bin/env python3
"""BASELINE config[4], end to end on synthetic data: one-shot library -> CLAP curation -> bank -> one training epoch.

The reference chain is ``data_modules/augment_data_with_CLAP.py:71-193`` (embed references + sample packs, cosine
similarity, bins, copy) -> ``copy_originals_to_augmented.py:33-83`` (references become the ``gold`` bin) ->
``convert_augmented_to_hdf5.py:69-141`` (the one-shot bank) -> ``train.py:253-328`` (one epoch on mixer-rendered
clips, checkpoints every ``save_every_n_steps``).  Here:

  1. a synthetic library is written to disk: ``refs/<pitch>/*.wav`` (gold one-shots per custom-GM pitch 35..60) and
     ``packs/pack_<k>/*.wav`` (N unlabelled one-shots), 16-bit WAV at 48 kHz;
  2. ``data_modules.augment_data_with_CLAP.run`` embeds and curates them on the GPU (K9-K12; random-init
     ``laion/clap-htsat-fused`` architecture -- the pretrained checkpoint is not available offline) and copies every
     pack file into ``refs_clap_augmented/<class>/<bin>/``;
  3. the references are copied to ``<class>/gold`` and the tree becomes a flat ``OneShotBank`` at the training rate
     (``OneShotBank.from_directory``: mono, resample on the GPU, peak-normalise);
  4. ``run_native_training`` runs one epoch over synthetic Lakh-style note chunks with the curated bank
     (``save_every_n_steps`` checkpoints, final ``model.safetensors``); with ``--check-resume`` the last checkpoint but one
     is resumed and must end on bitwise identical parameters.

Prints ONE JSON line with the stage times and rates.  Under ``torchrun`` (one rank per GPU) stages 2 and 4 shard
over the ranks (strided files + one all_gather; batch-sharded data parallel training over RCCL).

    python tools/e2e.py --shots 100000 --chunks 200000                      # config[4] sizes (8-GPU node: torchrun ... tools/e2e.py)
    python tools/e2e.py --shots 2000 --chunks 512 --tiny                    # what tests/test_e2e_config4_gpu.py runs
"""
from __future__ import annotations

import argparse
import json
import os
import random
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PITCHES = list(range(35, 61))                # the custom-GM pitches the model is trained on (mapping_utils.py:3-51)
CLAP_SR = 48000


def _shot(rng, proto, sr, n=None):
    """A decaying one-shot around a class prototype (centre frequency, decay, noise share), peak-normalised."""
    f0, decay, noise = proto
    n = n or int(rng.uniform(0.08, 0.5) * sr)
    t = np.arange(n, dtype=np.float32) / sr
    x = np.exp(-t * decay * rng.uniform(0.7, 1.4)) * (noise * rng.standard_normal(n).astype(np.float32)
                                                       + (1 - noise) * np.sin(2 * np.pi * f0 * rng.uniform(0.9, 1.1) * t))
    return (x / np.abs(x).max()).astype(np.float32)


def write_library(root, n_shots, refs_per_class, seed, rank=0, world=1):
    """refs/<pitch>/ref_<i>.wav and packs/pack_<k>/shot_<j>.wav (each rank writes its stride of the files)."""
    from adt_str_amd.audio_io import write_wav
    rng = np.random.default_rng(seed)
    protos = {p: (float(rng.uniform(40, 6000)), float(rng.uniform(6, 60)), float(rng.uniform(0.05, 0.9))) for p in PITCHES}
    jobs = [("refs", str(p), f"ref_{i}.wav", p) for p in PITCHES for i in range(refs_per_class)]
    cls = rng.integers(0, len(PITCHES), n_shots)
    jobs += [("packs", f"pack_{j // 1000:03d}", f"shot_{j:06d}.wav", PITCHES[int(cls[j])]) for j in range(n_shots)]
    for d in {os.path.join(root, top, sub) for top, sub, _, _ in jobs}:
        os.makedirs(d, exist_ok=True)

    def one(k):
        top, sub, name, pitch = jobs[k]
        write_wav(os.path.join(root, top, sub, name), _shot(np.random.default_rng(seed * 1000003 + k), protos[pitch], CLAP_SR), CLAP_SR)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=8) as pool:                 # numpy + file writes release the GIL
        list(pool.map(one, range(rank, len(jobs), world), chunksize=256))
    return os.path.join(root, "refs"), os.path.join(root, "packs")


def note_chunks(n_chunks, seed, max_notes=40):
    """Lakh-style rows: float32 [N, 4] (onset s, offset s, GM key, velocity) bytes (midi_parser.py:57-63), onsets < 2.95 s."""
    rng = np.random.default_rng(seed)
    gm = np.array([35, 36, 37, 38, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 53, 55, 57, 59], np.float32)
    rows = []
    for _ in range(n_chunks):
        n = int(rng.integers(4, max_notes + 1))
        on = np.sort(rng.uniform(0.0, 2.95, n)).astype(np.float32)
        rows.append(np.stack([on, on + np.float32(0.1), rng.choice(gm, n), rng.integers(1, 128, n).astype(np.float32)], 1).astype(np.float32).tobytes())
    return rows


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--workdir", default=None, help="scratch directory (default: a fresh one under $TMPDIR)")
    ap.add_argument("--shots", type=int, default=2000, help="unlabelled one-shots in the sample packs (config[4]: 100000)")
    ap.add_argument("--refs-per-class", type=int, default=5)
    ap.add_argument("--chunks", type=int, default=512, help="note chunks of the training epoch (config[4]: 200000)")
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--clap-batch", type=int, default=512)
    ap.add_argument("--input-sec", type=float, default=10.0)
    ap.add_argument("--sample-rate", type=int, default=16000)
    ap.add_argument("--save-every", type=int, default=0, help="checkpoint every n steps (0: a third of the epoch)")
    ap.add_argument("--tiny", action="store_true", help="1 + 1 layer network with 2 heads (tests); default is the setting-1 network")
    ap.add_argument("--check-resume", action="store_true")
    ap.add_argument("--curation-only", action="store_true", help="stop after the bank is built (config[4]'s curation half)")
    ap.add_argument("--keep", action="store_true", help="keep the scratch directory")
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args(argv)

    from adt_str_amd.trainer import init_distributed, latest_checkpoint, output_path, run_native_training, _checkpoint_dirs
    rank, local_rank, world = init_distributed()
    assert torch.cuda.is_available(), "tools/e2e.py needs a GPU (there is no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    import tempfile
    if a.workdir is None:
        box = [tempfile.mkdtemp(prefix="adt_e2e_") if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        a.workdir = box[0]
    os.makedirs(a.workdir, exist_ok=True)
    times = {}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # 1. library on disk
    t0 = time.perf_counter()
    ref_root, pack_root = write_library(a.workdir, a.shots, a.refs_per_class, a.seed, rank, world)
    barrier()
    times["write_library_s"] = time.perf_counter() - t0

    # 2. CLAP curation (K9-K12)
    from adt_str_amd.clap_encoder import random_init_clap_model
    from data_modules import augment_data_with_CLAP as aug
    cfg_clap = {"shared": {"sample_rate": CLAP_SR, "input_sec": a.input_sec, "time_res": 0.01, "win_length": 2048},
                "clap_config": {"model_name": "laion/clap-htsat-fused (random init)", "batch_size": a.clap_batch,
                                "sample_pack_root": pack_root, "reference_root": ref_root}}
    t0 = time.perf_counter()
    clap_model = random_init_clap_model(0)             # stands in for ClapModel.from_pretrained (no checkpoints on the box): timed apart
    times["clap_model_init_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    res, wav_files, aug_root = aug.run(cfg_clap, num_bins=10, clap_model=clap_model)
    barrier()
    times["curate_s"] = time.perf_counter() - t0
    times["curate_phases_s"] = dict(aug.PHASE_SECONDS)
    n_embedded = len(wav_files) + len(PITCHES) * a.refs_per_class

    # 3. gold + bank
    t0 = time.perf_counter()
    bank_path = os.path.join(a.workdir, f"oneshot@{a.sample_rate}.npz")
    if rank == 0:
        from adt_str_amd.bank import OneShotBank
        from adt_str_amd.curation import copy_originals_to_gold
        copy_originals_to_gold(ref_root, str(aug_root))
        bank = OneShotBank.from_directory(str(aug_root), a.sample_rate, device=dev)
        bank.save(bank_path)
        n_bank = bank.n_shots
    barrier()
    times["bank_s"] = time.perf_counter() - t0

    if a.curation_only:
        out = {"workload": "config[4] curation half (synthetic library)", "n_gpus": world, "shots": a.shots, "times": times,
               "embeds_per_s": n_embedded / times["curate_s"],
               "embeds_per_s_embedding_phase_incl_file_reads": len(wav_files) / max(times["curate_phases_s"].get("embed_packs", 0.0), 1e-9)}
        if rank == 0:
            out.update(bank_shots=n_bank, bins_used=sorted(set(res.bin)), n_assigned=len(res.order), aug_root=str(aug_root))
            print(json.dumps(out), flush=True)
        if not a.keep and rank == 0:
            shutil.rmtree(a.workdir, ignore_errors=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return out

    # 4. one epoch
    from adt_str_amd.data import GpuBatcher, NoteChunkDataset
    from adt_str_amd.network import ADTModel, ADTModelConfig
    from adt_str_amd.synth import SynthDrum, SynthDrumConfig
    from adt_str_amd.tokenizer import MidiTokenizer, MidiTokenizerConfig
    arch = dict(enc_layers=1, dec_layers=1, nhead=2) if a.tiny else dict(enc_layers=4, dec_layers=4, nhead=6)
    steps = a.chunks // (a.batch_size * world)
    save_every = a.save_every or max(1, steps // 3)
    out_dir = os.path.join(a.workdir, "outputs")
    cfg = {"training": dict(batch_size=a.batch_size, num_epochs=1, learning_rate=1e-4, weight_decay=1e-5, max_grad_norm=1.0, warmup_ratio=0.1,
                            gradient_accumulation_steps=1, min_learning_rate=None, lr_scheduler_type="cosine"),
           "logging": dict(output_dir=out_dir, logging_steps=max(1, steps // 10), save_every_n_steps=save_every),
           "checkpoint": dict(resume_from_checkpoint=None, auto_resume=False, max_checkpoints=3), "experiment": dict(seed=a.seed, run_name="e2e")}
    run_dir = output_path(cfg)                            # output_dir / run_name

    def build():
        random.seed(a.seed); torch.manual_seed(a.seed)
        model = ADTModel(ADTModelConfig(input_sec=a.input_sec, time_res=0.01, win_length=2048, sample_rate=a.sample_rate, d_query=128,
                                        dropout=0.1, tgt_vocab_size=1400, plain=True, n_mels=128, **arch)).to(dev)
        synth = SynthDrum(SynthDrumConfig(input_sec=a.input_sec, time_res=0.01, win_length=2048, sample_rate=a.sample_rate,
                                          oneshot_path=os.path.join(a.workdir, "oneshot"), similarity_threshold=0.0, max_hat_std_velocity=0.15,
                                          max_hat_mean_velocity=0.1, max_cymbals_std_velocity=0.15, max_cymbals_mean_velocity=0.65,
                                          ADTOF_mapping=False, mixup_range=0.8, use_fx_prob=0.3, use_reverb_prob=0.5, use_limiter_prob=0.5,
                                          use_compression_prob=0.5), device=dev)
        tk = MidiTokenizer(MidiTokenizerConfig(ADTOF_mapping=False, BOS_token=2, EOS_token=3, pad_token=1, silence_token=0, add_velocity=True))
        return model, NoteChunkDataset(note_chunks(a.chunks, a.seed + 1), GpuBatcher(tk, synth, 0.05, 0.5))

    model, ds = build()
    barrier()
    t0 = time.perf_counter()
    tr = run_native_training(model, ds, cfg)
    barrier()
    times["train_epoch_s"] = time.perf_counter() - t0
    out = {"workload": "config[4] end to end (synthetic library)", "n_gpus": world, "shots": a.shots, "chunks": a.chunks,
           "network": "tiny 1+1" if a.tiny else "setting-1 (69.0M)", "steps": tr.step_no, "times": times,
           "embeds_per_s": n_embedded / times["curate_s"],
           "embeds_per_s_embedding_phase_incl_file_reads": len(wav_files) / max(times["curate_phases_s"].get("embed_packs", 0.0), 1e-9), "train_clips_per_s": tr.step_no * a.batch_size * world / times["train_epoch_s"],
           "final_loss": tr.loss_history[-1][1] if getattr(tr, "loss_history", None) else None,
           "checkpoints": [os.path.basename(d) for d in _checkpoint_dirs(run_dir)]}
    if rank == 0:
        out["bank_shots"] = n_bank
        out["bins_used"] = sorted(set(res.bin))
    if a.check_resume:
        dirs = _checkpoint_dirs(run_dir)
        # a MID-epoch checkpoint: of the ones that are left, the one closest to half of the epoch (never the one written at the last step)
        cands = [d for d in dirs if int(d.rsplit("-", 1)[1]) < tr.step_no] or dirs
        src = min(cands, key=lambda d: abs(int(d.rsplit("-", 1)[1]) - tr.step_no / 2))
        final = tr.pflat.clone()
        cfg["checkpoint"]["resume_from_checkpoint"] = src
        model2, ds2 = build()
        tr2 = run_native_training(model2, ds2, cfg)
        barrier()
        out["resume"] = {"from": os.path.basename(src), "steps": tr2.step_no, "bitwise_identical": bool(torch.equal(tr2.pflat, final))}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if not a.keep and rank == 0:
        shutil.rmtree(a.workdir, ignore_errors=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
