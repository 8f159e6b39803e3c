"""H1 (include/adt_hip.h): the curation path's batched WAV decode and file copies against the per-file Python they replace
(adt_str_amd.audio_io.read_wav -> mean over channels -> x / max|x|, the reference's data_modules/augment_data_with_CLAP.py:51-68;
shutil.copy2, :184-196).  Host code only: runs without a GPU."""
import os
import shutil
import struct

import numpy as np
import pytest
import torch

from adt_str_amd import audio_io as A


def _riff(chunks):
    body = b"WAVE" + b"".join(tag + struct.pack("<I", size if size is not None else len(data)) + data + (b"\0" if len(data) & 1 else b"")
                              for tag, data, size in chunks)
    return b"RIFF" + struct.pack("<I", len(body)) + body


def _fmt(tag, channels, rate, bits, extensible_sub=None):
    block = channels * bits // 8
    base = struct.pack("<HHIIHH", 0xFFFE if extensible_sub is not None else tag, channels, rate, rate * block, block, bits)
    if extensible_sub is None:
        return base
    return base + struct.pack("<HHI", 22, bits, 3) + struct.pack("<H", extensible_sub) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"


def _pcm(rng, n, channels, bits, tag=1):
    if tag == 3:
        return (rng.standard_normal(n * channels) * 0.4).astype("<f4").tobytes()
    if bits == 8:
        return rng.integers(0, 256, n * channels, dtype=np.uint8).tobytes()
    if bits == 16:
        return rng.integers(-32768, 32768, n * channels).astype("<i2").tobytes()
    if bits == 32:
        return rng.integers(-2 ** 31, 2 ** 31, n * channels).astype("<i4").tobytes()
    v = rng.integers(-2 ** 23, 2 ** 23, n * channels).astype(np.int64) & 0xFFFFFF
    return b"".join(int(x).to_bytes(3, "little") for x in v)


def _library(tmp_path):
    rng = np.random.default_rng(3)
    files = {}

    def put(name, blob):
        p = str(tmp_path / name)
        with open(p, "wb") as fh:
            fh.write(blob)
        files[name] = p

    for bits in (8, 16, 24, 32):
        for ch in (1, 2, 3):
            put(f"pcm{bits}_{ch}.wav", _riff([(b"fmt ", _fmt(1, ch, 48000, bits), None), (b"data", _pcm(rng, 777 + bits + ch, ch, bits), None)]))
    put("float_2.wav", _riff([(b"fmt ", _fmt(3, 2, 44100, 32), None), (b"data", _pcm(rng, 500, 2, 32, tag=3), None)]))
    put("float_nan.wav", _riff([(b"fmt ", _fmt(3, 1, 48000, 32), None), (b"data", np.array([0.5, np.nan, -0.25], "<f4").tobytes(), None)]))
    put("extensible.wav", _riff([(b"fmt ", _fmt(1, 2, 48000, 16, extensible_sub=1), None), (b"data", _pcm(rng, 300, 2, 16), None)]))
    put("list_chunk_odd.wav", _riff([(b"LIST", b"abc", None), (b"fmt ", _fmt(1, 1, 22050, 16), None), (b"fact", b"\x01\x02\x03\x04", None),
                                     (b"data", _pcm(rng, 123, 1, 16), None)]))
    put("two_data_chunks.wav", _riff([(b"fmt ", _fmt(1, 1, 48000, 16), None), (b"data", _pcm(rng, 50, 1, 16), None),
                                      (b"data", _pcm(rng, 70, 1, 16), None)]))
    put("truncated_data.wav", _riff([(b"fmt ", _fmt(1, 1, 48000, 16), None), (b"data", _pcm(rng, 100, 1, 16), 100000)]))
    put("ragged_frames.wav", _riff([(b"fmt ", _fmt(1, 2, 48000, 16), None), (b"data", _pcm(rng, 101, 1, 16), None)]))      # 101 samples, 2 channels
    put("pcm24_partial.wav", _riff([(b"fmt ", _fmt(1, 1, 48000, 24), None), (b"data", _pcm(rng, 40, 1, 24) + b"\x01\x02", None)]))
    put("silent.wav", _riff([(b"fmt ", _fmt(1, 1, 48000, 16), None), (b"data", b"\0" * 400, None)]))
    put("empty.wav", _riff([(b"fmt ", _fmt(1, 1, 48000, 16), None), (b"data", b"", None)]))
    # the ones read_wav refuses
    put("odd_bytes_16.wav", _riff([(b"fmt ", _fmt(1, 1, 48000, 16), None), (b"data", b"\x01\x02\x03", None)]))
    put("float64.wav", _riff([(b"fmt ", _fmt(3, 1, 48000, 64), None), (b"data", b"\0" * 64, None)]))
    put("adpcm.wav", _riff([(b"fmt ", _fmt(2, 1, 48000, 4), None), (b"data", b"\0" * 64, None)]))
    put("no_data.wav", _riff([(b"fmt ", _fmt(1, 1, 48000, 16), None)]))
    put("no_fmt.wav", _riff([(b"data", b"\0" * 64, None)]))
    put("short_fmt.wav", _riff([(b"fmt ", b"\x01\x00\x01\x00", None), (b"data", b"\0" * 64, None)]))
    put("not_riff.wav", b"OggS" + b"\0" * 100)
    put("tiny.wav", b"RIFF")
    files["missing.wav"] = str(tmp_path / "missing.wav")
    return files


@pytest.mark.parametrize("threads", [1, 5])
def test_batched_decode_is_read_wav_mean_normalize_bitwise(tmp_path, threads):
    files = _library(tmp_path)
    paths = list(files.values())
    for norm in (False, True):
        b = A.read_wav_batch(paths, normalize=norm, threads=threads)
        assert b.offsets[0] == 0 and b.offsets[-1] == b.data.numel() and len(b.offsets) == len(paths) + 1
        n_ok = 0
        for i, (name, p) in enumerate(files.items()):
            got = b.data[b.offsets[i]:b.offsets[i + 1]]
            try:
                x, sr = A.read_wav(p)
            except Exception:
                assert b.status[i] != 0 and got.numel() == 0, name
                continue
            n_ok += 1
            assert b.status[i] == 0 and b.sample_rate[i] == sr, name
            mono = torch.from_numpy(x.mean(axis=0, keepdims=True))[0]                        # load_audio's mean over channels
            assert got.shape == mono.shape, name
            if mono.numel():
                peak = torch.max(torch.abs(mono))
                assert np.array_equal(np.float32(b.peak[i]), peak.numpy(), equal_nan=True), name
                want = mono / peak if norm else mono                                         # normalize(): x / max|x|
            else:
                want = mono
            assert np.array_equal(got.numpy(), want.numpy(), equal_nan=True), name
        assert n_ok == 22
    assert A.read_wav_batch([]).data.numel() == 0


def test_batched_copy_is_copy2(tmp_path):
    files = _library(tmp_path)
    srcs = [p for n, p in files.items() if os.path.exists(p)]
    os.utime(srcs[0], ns=(1_600_000_000_123_456_789, 1_500_000_000_987_654_321))
    os.chmod(srcs[1], 0o640)
    (tmp_path / "out").mkdir()
    dsts = [str(tmp_path / "out" / os.path.basename(p)) for p in srcs]
    with open(dsts[2], "wb") as fh:                                   # an existing, longer destination is replaced
        fh.write(b"x" * 1_000_000)
    status = A.copy_files(srcs + [files["missing.wav"], srcs[0]], dsts + [str(tmp_path / "out" / "m.wav"), str(tmp_path / "nodir" / "a.wav")], threads=4)
    assert status[:-2].tolist() == [0] * len(srcs) and status[-2] != 0 and status[-1] != 0
    before = open(srcs[3], "rb").read()
    assert A.copy_files([srcs[3]], [srcs[3]])[0] != 0 and open(srcs[3], "rb").read() == before     # onto itself: refused, not truncated
    for s, d in zip(srcs, dsts):
        ref = str(tmp_path / "ref.bin")
        shutil.copy2(s, ref)
        with open(d, "rb") as a, open(ref, "rb") as b:
            assert a.read() == b.read()
        sa, sb = os.stat(d), os.stat(ref)
        assert sa.st_mode == sb.st_mode and sa.st_mtime_ns == sb.st_mtime_ns          # (atime moves with every read of the source)


def test_bank_from_directory_matches_the_per_file_route(tmp_path):
    """OneShotBank.from_directory decodes through the batched call; shot for shot it is read_wav -> mean -> x / peak, silent and
    unreadable files skipped, duplicate stems suffixed, non-integer labels and shallow paths ignored."""
    from adt_str_amd.bank import GROUPS, OneShotBank
    rng = np.random.default_rng(5)
    root = tmp_path / "aug"
    expect = {}
    for label in ("36", "38", "hat"):
        for g in GROUPS[:2]:
            (root / label / g).mkdir(parents=True)
            for k in range(3):
                ch = 1 + (k % 2)
                x = (rng.standard_normal((ch, 200 + 10 * k)) * 0.2).astype(np.float32)
                A.write_wav(str(root / label / g / f"s{k}.wav"), x, 16000)
                if label != "hat":
                    m = A.read_wav(str(root / label / g / f"s{k}.wav"))[0].mean(axis=0)
                    expect[(int(label), g, f"s{k}")] = (m / float(np.abs(m).max())).astype(np.float32)
    A.write_wav(str(root / "36" / GROUPS[0] / "silent.wav"), np.zeros(100, np.float32), 16000)
    (root / "36" / GROUPS[0] / "junk.wav").write_bytes(b"not a wav file")
    A.write_wav(str(root / "36" / "shallow.wav"), np.ones(10, np.float32) * 0.1, 16000)
    bank = OneShotBank.from_directory(str(root), 16000)
    assert bank.n_shots == len(expect) == 12
    for (pitch, g, name), want in expect.items():
        assert np.array_equal(bank.shot(bank.shot_id(pitch, g, name)), want), (pitch, g, name)
    assert OneShotBank.from_directory(str(root), 8000).n_shots == 0      # another rate needs the GPU resampler: every file is skipped with a message


def test_pipeline_scripts_gold_copy_and_bank_conversion(tmp_path, capsys):
    """The reference's curation pipeline after augment_data_with_CLAP.py (DATASET_AUGMENTATION_PIPELINE.md): copy_originals_to_augmented.py
    then convert_augmented_to_hdf5.py, with the reference's command lines (the bank is this repo's flat .npz instead of HDF5)."""
    import yaml
    from adt_str_amd.bank import GROUPS, OneShotBank
    from data_modules import convert_augmented_to_hdf5 as conv
    from data_modules import copy_originals_to_augmented as gold
    rng = np.random.default_rng(1)
    ref = tmp_path / "GM"
    for label in ("36", "38"):
        (ref / label / "sub").mkdir(parents=True)
        for k in range(2):
            A.write_wav(str(ref / label / f"r{k}.wav"), (rng.standard_normal(300) * 0.2).astype(np.float32), 44100)
        A.write_wav(str(ref / label / "sub" / "deep.wav"), (rng.standard_normal(200) * 0.2).astype(np.float32), 44100)
    (ref / "notes.txt").write_text("not a label directory")
    aug = tmp_path / "GM_clap_augmented"
    (aug / "36" / GROUPS[1]).mkdir(parents=True)
    A.write_wav(str(aug / "36" / GROUPS[1] / "picked.wav"), (rng.standard_normal(250) * 0.2).astype(np.float32), 44100)
    cfg = tmp_path / "clap.yaml"
    cfg.write_text(yaml.safe_dump({"clap_config": {"reference_root": str(ref), "sample_pack_root": str(tmp_path / "packs")}}))
    gold.main([str(cfg)])
    assert "Copied: 2, Skipped: 0" in capsys.readouterr().out
    assert sorted(os.listdir(aug / "36" / "gold")) == ["r0.wav", "r1.wav", "sub"] and (aug / "38" / "gold" / "sub" / "deep.wav").exists()
    assert os.stat(aug / "38" / "gold" / "r0.wav").st_mtime_ns == os.stat(ref / "38" / "r0.wav").st_mtime_ns
    gold.main([str(cfg)])
    assert "Copied: 0, Skipped: 2" in capsys.readouterr().out
    (aug / "36" / "gold" / "stale.wav").write_bytes(b"x")
    gold.main([str(cfg), "--overwrite"])
    assert "Copied: 2, Skipped: 0" in capsys.readouterr().out and not (aug / "36" / "gold" / "stale.wav").exists()

    out = conv.main([str(aug), str(tmp_path / "oneshot"), "--sample_rate", "44100"])
    assert out.endswith("oneshot@44100.npz") and "Wrote 7 items" in capsys.readouterr().out      # gold/sub/deep.wav sits deeper but still counts
    bank = OneShotBank.load(out)
    assert bank.sample_rate == 44100 and bank.has_cell(36, "gold") and bank.has_cell(36, GROUPS[1]) and bank.has_cell(38, "gold")
    m = A.read_wav(str(ref / "38" / "r1.wav"))[0].mean(axis=0)
    assert np.array_equal(bank.shot(bank.shot_id(38, "gold", "r1")), (m / float(np.abs(m).max())).astype(np.float32))
    with pytest.raises(FileExistsError):
        conv.main([str(aug), str(tmp_path / "oneshot"), "--sample_rate", "44100"])
    conv.main([str(aug), str(tmp_path / "oneshot"), "--sample_rate", "44100", "--overwrite"])


def test_randomised_wav_layouts_decode_like_read_wav(tmp_path):
    """200 random RIFF layouts (chunk order, odd-sized extra chunks, channels 1-4, every supported encoding, cut-off files, random
    garbage after the last chunk): the batched decoder accepts exactly what read_wav accepts and returns its mono samples bitwise."""
    rng = np.random.default_rng(2026)
    paths = []
    for i in range(200):
        tag, bits = [(1, 8), (1, 16), (1, 24), (1, 32), (3, 32)][int(rng.integers(0, 5))]
        ch, n = int(rng.integers(1, 5)), int(rng.integers(0, 400))
        chunks = []
        for _ in range(int(rng.integers(0, 3))):
            chunks.append((bytes(rng.choice(list(b"LISTfactcue bext"), 4).astype(np.uint8)), rng.integers(0, 256, int(rng.integers(0, 23)), dtype=np.uint8).tobytes(), None))
        fmt = (b"fmt ", _fmt(tag, ch, int(rng.choice([8000, 16000, 44100, 48000])), bits, extensible_sub=tag if rng.random() < 0.3 else None), None)
        data = (b"data", _pcm(rng, n, ch, bits, tag=tag), None)
        body = chunks + ([fmt, data] if rng.random() < 0.8 else [data, fmt])
        blob = _riff(body)
        r = rng.random()
        if r < 0.1:
            blob = blob[: int(rng.integers(8, len(blob) + 1))]                      # cut off anywhere
        elif r < 0.2:
            blob += rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8).tobytes()   # trailing garbage
        p = str(tmp_path / f"r{i}.wav")
        with open(p, "wb") as fh:
            fh.write(blob)
        paths.append(p)
    b = A.read_wav_batch(paths, normalize=False, threads=4)
    n_ok = 0
    for i, p in enumerate(paths):
        got = b.data[b.offsets[i]:b.offsets[i + 1]].numpy()
        try:
            x, sr = A.read_wav(p)
        except Exception:
            assert b.status[i] != 0 and got.size == 0, p
            continue
        n_ok += 1
        assert b.status[i] == 0 and b.sample_rate[i] == sr, p
        assert np.array_equal(got, x.mean(axis=0), equal_nan=True), p
    assert 120 < n_ok < 200                                                       # most parse, the cut-off ones mostly do not


def test_load_clips_batch_on_the_cpu_for_files_at_the_target_rate(tmp_path):
    """Without a GPU the batched loader still serves files that need no resampling (decode + peak-normalise); a file at another
    rate needs K13 and raises."""
    rng = np.random.default_rng(4)
    paths = []
    for i in range(5):
        p = str(tmp_path / f"c{i}.wav")
        A.write_wav(p, (rng.standard_normal((1 + i % 2, 100 + 17 * i)) * 0.3).astype(np.float32), 48000)
        paths.append(p)
    (tmp_path / "bad.wav").write_bytes(b"nope")
    clips, peaks, status = A.load_clips_batch(paths + [str(tmp_path / "bad.wav")], 48000, "cpu")
    assert status[-1] != 0 and clips[-1] is None
    for p, c, pk in zip(paths, clips, peaks):
        m = A.read_wav(p)[0].mean(axis=0)
        assert np.array_equal(c.numpy(), (torch.from_numpy(m) / torch.max(torch.abs(torch.from_numpy(m)))).numpy())
        assert float(pk) == float(np.abs(m).max())
    other = str(tmp_path / "other.wav")
    A.write_wav(other, (rng.standard_normal(200) * 0.3).astype(np.float32), 44100)
    with pytest.raises(RuntimeError):
        A.load_clips_batch([other], 48000, "cpu")


def test_write_then_batched_read_round_trip(tmp_path):
    """write_wav (16-bit PCM) -> read_wav_batch: the samples come back as the quantised input, for mono and stereo, any length."""
    rng = np.random.default_rng(8)
    paths, want = [], []
    for i in range(20):
        ch, n = 1 + i % 2, int(rng.integers(1, 3000))
        x = np.clip(rng.standard_normal((ch, n)) * 0.4, -1, 1).astype(np.float32)
        p = str(tmp_path / f"w{i}.wav")
        A.write_wav(p, x, 48000)
        q = (np.clip(x, -1.0, 1.0) * 32767.0).round().astype(np.int16).astype(np.float32) / 32768.0
        paths.append(p)
        want.append(q.mean(axis=0))
    b = A.read_wav_batch(paths, threads=3)
    assert (b.status == 0).all() and (b.sample_rate == 48000).all()
    for i, w in enumerate(want):
        assert np.array_equal(b.data[b.offsets[i]:b.offsets[i + 1]].numpy(), w)
