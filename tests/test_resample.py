"""K13 polyphase resampler vs the oracle restatement of torchaudio.transforms.Resample (conv1d on the CPU).
fp32 with <= 37 taps: 2e-6 absolute on inputs in [-1, 1].  CPU part: the host kernel bank equals the oracle's."""
import math

import numpy as np
import pytest
import torch

from oracle import resample as o_res

PAIRS = [(44100, 48000), (22050, 16000), (48000, 16000), (16000, 24000), (96000, 48000), (8000, 48000), (32000, 48000)]


@pytest.mark.parametrize("orig,new", PAIRS)
def test_host_kernel_bank_equals_oracle(orig, new):
    from adt_str_amd.resample import sinc_kernel_bank
    k, width, o, n = o_res.sinc_resample_kernel(orig, new)
    bank, rng, w2, o2, n2 = sinc_kernel_bank(orig, new)
    assert (width, o, n) == (w2, o2, n2) and np.array_equal(k[:, 0].numpy(), bank)
    for p in range(n):                                  # everything outside the advertised tap range is an exact zero
        assert not bank[p, :rng[p, 0]].any() and not bank[p, rng[p, 1]:].any()


def test_oracle_preserves_a_sine_and_length_rule():
    sr, new = 44100, 48000
    t = torch.arange(sr) / sr
    y = o_res.resample(torch.sin(2 * math.pi * 1000.0 * t), sr, new)
    assert y.shape[-1] == math.ceil(new * sr / sr)
    tt = torch.arange(y.shape[-1]) / new
    assert (y[200:-200] - torch.sin(2 * math.pi * 1000.0 * tt)[200:-200]).abs().max() < 2e-3
    assert o_res.resample(torch.zeros(3, 1001), 22050, 16000).shape == (3, math.ceil(16000 * 1001 / 22050))


@pytest.mark.gpu
@pytest.mark.parametrize("orig,new", PAIRS)
def test_gpu_matches_oracle(orig, new):
    from adt_str_amd.resample import Resample
    g = torch.Generator().manual_seed(orig + new)
    x = torch.rand((3, 20011), generator=g) * 2 - 1
    x[1, :5000] = 0.0
    ref = o_res.resample(x, orig, new)
    got = Resample(orig, new)(x.cuda()).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max() < 2e-6
    one = Resample(orig, new)(x[0].cuda()).cpu()         # 1-D input keeps its rank
    assert one.shape == ref[0].shape and (one - ref[0]).abs().max() < 2e-6


@pytest.mark.gpu
def test_gpu_edge_cases():
    from adt_str_amd.resample import Resample
    x = torch.randn(2, 1, 7).cuda()
    assert Resample(16000, 16000)(x) is x                                    # identity returns the input (torchaudio does too)
    y = Resample(48000, 16000)(x)
    assert y.shape == (2, 1, 3) and (y.cpu() - o_res.resample(x.cpu(), 48000, 16000)).abs().max() < 2e-6
    assert Resample(16000, 48000)(torch.zeros(0, 100).cuda()).shape == (0, 300)
    with pytest.raises(ValueError):
        Resample(0, 16000)


@pytest.mark.gpu
def test_batched_load_resample_normalize_is_the_per_file_route(tmp_path, monkeypatch):
    """audio_io.load_clips_batch (one decode call, one PCIe copy, K13 over zero-padded groups, peaks and the division on the GPU)
    against the reference's per-file load -> mean -> Resample -> x / max|x| (data_modules/augment_data_with_CLAP.py:51-68), bitwise,
    on a ragged library of mixed rates / channel counts; the padded groups are also forced to be tiny so several are formed."""
    from adt_str_amd import audio_io as A
    from adt_str_amd.resample import Resample
    rng = np.random.default_rng(11)
    paths, rates = [], []
    for i in range(40):
        sr = (44100, 48000, 22050, 48000, 96000)[i % 5]
        ch = 1 + (i % 3 == 0)
        n = int(rng.integers(50, 30000))
        p = str(tmp_path / f"f{i}.wav")
        A.write_wav(p, (rng.standard_normal((ch, n)) * 0.3).astype(np.float32), sr)
        paths.append(p); rates.append(sr)
    A.write_wav(str(tmp_path / "silent.wav"), np.zeros(100, np.float32), 44100)
    (tmp_path / "junk.wav").write_bytes(b"junk")
    A.write_wav(str(tmp_path / "empty.wav"), np.zeros(0, np.float32), 48000)
    paths += [str(tmp_path / "silent.wav"), str(tmp_path / "junk.wav"), str(tmp_path / "empty.wav")]

    def per_file(p):
        x, sr = A.read_wav(p)
        w = torch.from_numpy(x.mean(axis=0, keepdims=True))
        if sr != 48000:
            w = Resample(sr, 48000)(w.cuda()).cpu()
        return w[0], torch.max(torch.abs(w))

    for budget in (1 << 26, 40000):
        monkeypatch.setattr(A, "_PAD_BUDGET", budget)
        for norm in (True, False):
            clips, peaks, status = A.load_clips_batch(paths, 48000, "cuda:0", normalize=norm, threads=3)
            assert status[-2] != 0 and clips[-2] is None and clips[-1] is None and (status[:-2] == 0).all()
            for i, p in enumerate(paths[:-2]):
                want, peak = per_file(p)
                assert clips[i].shape == want.shape, p
                assert np.array_equal(peaks[i].cpu().numpy(), peak.numpy(), equal_nan=True), p
                ref = want / peak if norm else want
                assert np.array_equal(clips[i].cpu().numpy(), ref.numpy(), equal_nan=True), (p, budget, norm)


@pytest.mark.gpu
def test_bank_from_directory_on_the_gpu_matches_per_file(tmp_path):
    from adt_str_amd import audio_io as A
    from adt_str_amd.bank import GROUPS, OneShotBank
    from adt_str_amd.resample import Resample
    rng = np.random.default_rng(2)
    root = tmp_path / "aug"
    expect = {}
    for label in ("36", "42"):
        for g in GROUPS[:2]:
            (root / label / g).mkdir(parents=True)
            for k in range(4):
                sr = (16000, 44100, 48000, 16000)[k]
                x = (rng.standard_normal((1 + k % 2, 300 + 37 * k)) * 0.2).astype(np.float32)
                path = str(root / label / g / f"s{k}.wav")
                A.write_wav(path, x, sr)
                m = torch.from_numpy(A.read_wav(path)[0].mean(axis=0))
                if sr != 16000:
                    m = Resample(sr, 16000)(m.cuda()[None])[0].cpu()
                m = m.numpy()
                expect[(int(label), g, f"s{k}")] = (m / float(np.abs(m).max())).astype(np.float32)
    A.write_wav(str(root / "36" / GROUPS[0] / "silent.wav"), np.zeros(100, np.float32), 44100)
    bank = OneShotBank.from_directory(str(root), 16000, device="cuda:0")
    assert bank.n_shots == len(expect) == 16
    for (pitch, g, name), want in expect.items():
        assert np.array_equal(bank.shot(bank.shot_id(pitch, g, name)), want), (pitch, g, name)
