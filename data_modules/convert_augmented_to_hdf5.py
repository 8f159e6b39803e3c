"""Drop-in for the reference's ``data_modules/convert_augmented_to_hdf5.py``:
``python data_modules/convert_augmented_to_hdf5.py <input_root> <output> [--sample_rate 44100] [--overwrite]``.

Third step of the curation pipeline: every ``<input_root>/<label>/<bin>/<file>.wav`` is loaded as mono, resampled to
``--sample_rate`` and peak-normalised (reference :97-103), duplicates inside a cell get ``_2``, ``_3`` ... (:113-118).  The reference
packs them into ``<output>@<sr>.hdf5`` (one gzip dataset per file, re-opened per note by its synthesiser); here they go into the
flat bank ``<output>@<sr>.npz`` that ``SynthDrum`` uploads to HBM once (adt_str_amd/bank.py; INTEGRATION.md, "What changes for
a user").  Decode, resampling (K13) and normalisation run batched on the GPU (adt_str_amd.audio_io.load_clips_batch)."""
import argparse
import os
import sys

sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from adt_str_amd.bank import OneShotBank  # noqa: E402


def main(argv=None) -> str:
    parser = argparse.ArgumentParser()
    parser.add_argument("input_root", type=str, help="Path to augmented dataset root (e.g., /path/to/GM_Mapped_Reduced_clap_augmented)")
    parser.add_argument("output_hdf5", type=str, help="Path stem of the output bank (written as <stem>@<sample_rate>.npz)")
    parser.add_argument("--sample_rate", type=int, default=44100, help="Target sample rate for audio resampling (default: 44100)")
    parser.add_argument("--overwrite", action="store_true", help="If set, overwrite an existing output file")
    args = parser.parse_args(argv)
    if not os.path.isdir(args.input_root):
        raise FileNotFoundError(f"Input root does not exist: {args.input_root}")
    out = f"{args.output_hdf5}@{args.sample_rate}.npz"
    if os.path.exists(out):
        if not args.overwrite:
            raise FileExistsError(f"Output file exists: {out}. Use --overwrite to replace.")
        os.unlink(out)
    device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    bank = OneShotBank.from_directory(args.input_root, args.sample_rate, device=device)
    bank.save(out)
    print(f"Done. Wrote {bank.n_shots} items to {out}")
    return out


if __name__ == "__main__":
    main()
