"""Drop-in for the reference's ``inference.py``: ``python inference.py <in.wav> <config.yaml> [-o dir] [-s]``.

load -> mono -> fixed chunks of ``input_sec`` (last one zero padded, inference.py:35-48) -> greedy ``sample`` ->
``decode`` -> notes shifted by the chunk start -> unique rows -> Standard MIDI File (+ optional re-synthesis).
WAV reading and MIDI writing are in adt_str_amd/audio_io.py (no torchaudio / pretty_midi on the GPU box).
"""
import argparse
import os

import numpy as np
import torch

from adt_str_amd.audio_io import read_wav, write_drum_midi, write_wav
from adt_str_amd.tokenizer import MidiTokenizer, MidiTokenizerConfig
from build_model import build_model


def _chunk_audio(wav: torch.Tensor, chunk_samples: int):
    """[samples] -> [n_chunks, chunk_samples], the tail zero padded."""
    n = max(1, -(-wav.numel() // chunk_samples))
    out = torch.zeros(n * chunk_samples, dtype=wav.dtype, device=wav.device)
    out[: wav.numel()] = wav
    return out.view(n, chunk_samples)


def transcribe(model, cfg: dict, wav: torch.Tensor, batch_size: int = 8, token_sink=None):
    """``token_sink``: a list that receives the generated token ids of every chunk (one list of ints per chunk, in order)."""
    shared, inf = cfg["shared"], cfg["inference"]
    tok_cfg = cfg.get("tokenizer") or dict(ADTOF_mapping=False, BOS_token=2, EOS_token=3, pad_token=1, silence_token=0, add_velocity=True)
    tokenizer = MidiTokenizer(MidiTokenizerConfig(**tok_cfg))
    chunk = int(round(shared["input_sec"] * shared["sample_rate"]))
    chunks = _chunk_audio(wav, chunk)
    notes = []
    for i in range(0, chunks.shape[0], batch_size):
        ids = model.sample(src=chunks[i:i + batch_size], src_mask=None, tgt_mask=None, max_length=inf["max_length"],
                           start_token=tok_cfg["BOS_token"], end_token=tok_cfg["EOS_token"]).cpu()
        for j, row in enumerate(ids):
            if token_sink is not None:
                token_sink.append(row.tolist())
            dec = tokenizer.decode(row.tolist())
            if dec.numel():
                dec = dec.clone()
                dec[:, :2] += (i + j) * shared["input_sec"]
                notes.append(dec)
    if not notes:
        return np.zeros((0, 4), np.float32)
    return np.unique(torch.cat(notes).numpy(), axis=0)


def _parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser()
    ap.add_argument("input_path")
    ap.add_argument("config_path")
    # the reference's spellings (inference.py:55-68) first; --output_dir / --synthesize are this repository's earlier aliases
    ap.add_argument("-o", "--output_path", "--output_dir", dest="output_dir", default="outputs/", help="Directory to save output files")
    ap.add_argument("-s", "--synthetise_transcription", "--synthesize", dest="synthesize", action="store_true",
                    help="Resynthesize the drum transcription with the one-shot mixer")
    ap.add_argument("--save-tokens", action="store_true", help="also write <stem>.tokens.json: the generated token ids per chunk")
    return ap


def main():
    a = _parser().parse_args()
    model, cfg = build_model(a.config_path, device="cuda")
    audio, sr = read_wav(a.input_path)
    wav = torch.from_numpy(audio.mean(axis=0)).cuda()
    if sr != cfg["shared"]["sample_rate"]:                 # reference inference.py:88-90
        from adt_str_amd.resample import Resample
        wav = Resample(sr, cfg["shared"]["sample_rate"])(wav)
    tokens = [] if a.save_tokens else None
    notes = transcribe(model, cfg, wav, cfg["inference"]["batch_size"], tokens)
    os.makedirs(a.output_dir, exist_ok=True)
    stem = os.path.splitext(os.path.basename(a.input_path))[0]
    if tokens is not None:
        import json
        with open(os.path.join(a.output_dir, stem + ".tokens.json"), "w") as fh:
            json.dump({"precision": model.engine.precision, "max_length": cfg["inference"]["max_length"], "chunks": tokens}, fh)
    write_drum_midi(os.path.join(a.output_dir, stem + ".mid"), notes.tolist())
    print(f"{len(notes)} notes -> {os.path.join(a.output_dir, stem + '.mid')}")
    if a.synthesize and len(notes):
        from train import build_components
        _, _, synth = build_components(cfg)
        ok = [n for n in notes.tolist() if 35 <= n[2] <= 61]
        write_wav(os.path.join(a.output_dir, stem + "_resynth.wav"), synth(ok).cpu().numpy(), cfg["shared"]["sample_rate"])


if __name__ == "__main__":
    main()
