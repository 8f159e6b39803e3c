"""Where a curation batch's wall time goes on the host side (20 000 synthetic one-shots, batches of 512): waiting for the I/O thread's decode,
the main thread's _load_batch (PCIe copy, K13 launches, normalisation), get_audio_features (launch side only), and the final synchronise."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "data_modules"))
import torch
import e2e
import augment_data_with_CLAP as drv
from adt_str_amd.audio_io import read_wav_batch
from adt_str_amd.clap_encoder import ClapWrapper, random_init_clap_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
with tempfile.TemporaryDirectory() as tmp:
    e2e.write_library(tmp, n, 5, 0)
    files = sorted(os.path.join(d, f) for d, _, fs in os.walk(os.path.join(tmp, "packs")) for f in fs if f.endswith(".wav"))
    if not files:
        files = sorted(os.path.join(d, f) for d, _, fs in os.walk(tmp) for f in fs if f.endswith(".wav"))
    wrap = ClapWrapper("random-init", "cuda:0", 48000, clap_model=random_init_clap_model(0))
    dev = torch.device("cuda:0")
    bs = 512
    t = {"decode": 0.0, "load_batch": 0.0, "features": 0.0}
    drv._embed(wrap, files[:1024], bs, 48000)           # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(0, len(files), bs):
        names = files[i:i + bs]
        a = time.perf_counter()
        dec = read_wav_batch(names, False, True)
        b = time.perf_counter()
        batch = drv._load_batch(names, 48000, dev, dec)
        c = time.perf_counter()
        emb = wrap.get_audio_features(batch).float()
        d = time.perf_counter()
        t["decode"] += b - a; t["load_batch"] += c - b; t["features"] += d - c
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    nb = -(-len(files) // bs)
    print(f"{len(files)} files, {nb} batches, serial loop {tot:.2f} s = {len(files) / tot:.0f} files/s; per batch: decode {t['decode'] / nb * 1e3:.1f} ms, "
          f"_load_batch {t['load_batch'] / nb * 1e3:.1f} ms, get_audio_features (host side) {t['features'] / nb * 1e3:.1f} ms")
    t0 = time.perf_counter()
    drv._embed(wrap, files, bs, 48000)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"pipelined _embed: {tot:.2f} s = {len(files) / tot:.0f} files/s")
