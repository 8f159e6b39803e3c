"""K2 one-shot mixer: host planning (CPU) and GPU rendering vs the oracle and
the reference's golden clips (tests/golden/mixer.npz: notes + recorded RNG
draws + wav rendered by the reference's own SynthDrum)."""
import os
import random

import numpy as np
import pytest
import torch

from adt_str_amd.bank import OneShotBank
from adt_str_amd.synth import NOTE_DTYPE, SynthDrum, SynthDrumConfig, vel_to_vol
from oracle import mixer as o_mixer
from oracle.bank import synthetic_bank


def make_cfg(sr, input_sec, thr, mixup_range, adtof):
    return SynthDrumConfig(input_sec=input_sec, time_res=0.01, win_length=2048, sample_rate=sr, oneshot_path="/none",
                           similarity_threshold=thr, max_hat_std_velocity=0.15, max_hat_mean_velocity=0.1,
                           max_cymbals_std_velocity=0.15, max_cymbals_mean_velocity=0.65, ADTOF_mapping=adtof,
                           mixup_range=mixup_range, use_fx_prob=0.0, use_reverb_prob=0.5, use_limiter_prob=0.5,
                           use_compression_prob=0.5)


@pytest.fixture(scope="module")
def golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "mixer.npz"))
    tree = synthetic_bank(int(g["bank_seed"]), 16000).tree
    return g, tree, OneShotBank.from_tree(tree, 16000)


def golden_cases(g):
    for c in range(int(g["n_cases"])):
        adtof, thr, mr, isec, sr = g[f"s{c}_cfg"]
        yield c, make_cfg(int(sr), float(isec), float(thr), float(mr), bool(adtof)), g[f"s{c}_notes"], int(g[f"s{c}_seed"])


def oracle_from_plan(plan, i, notes, cfg, tree):
    picks = plan.picks[i]

    def timbres(p):
        (mp, mg, mn), (sp, sg, sn) = picks[p]
        return tree[str(mp)][mg][mn], tree[str(sp)][sg][sn]

    return o_mixer.render(notes, cfg.input_sec, cfg.sample_rate, cfg.ADTOF_mapping, timbres, plan.mixups[i])


def test_plan_reproduces_reference_draws(golden):
    """Seeding ``random`` like the reference run must give the same timbre picks,
    mixups and clip length, i.e. the host consumes the RNG in the reference's order."""
    g, tree, bank = golden
    for c, cfg, notes, seed in golden_cases(g):
        sd = SynthDrum(cfg, bank=bank, device="cpu")
        random.seed(seed)
        plan = sd.plan([notes.tolist()])
        rec = [str(x) for x in g[f"s{c}_choices"]]
        mine = []
        seen = []
        for n in notes:
            p = int(n[2])
            if p in seen:
                continue
            seen.append(p)
            for (pp, grp, name) in plan.picks[0][p]:
                mine += ([str(pp)] if cfg.ADTOF_mapping else []) + [grp, name]
        assert mine == rec
        assert np.allclose(plan.mixups[0], g[f"s{c}_uniforms"], rtol=0, atol=0)
        assert int(plan.clip_len[0]) == g[f"s{c}_wav"].shape[0]
        assert plan.notes.dtype == NOTE_DTYPE and plan.notes.shape[0] == len(notes)
        ref = oracle_from_plan(plan, 0, notes.tolist(), cfg, tree).numpy()
        assert np.array_equal(ref, g[f"s{c}_wav"])                  # oracle is bit-exact with the reference


def test_plan_layout_and_errors(golden):
    g, tree, bank = golden
    sd = SynthDrum(make_cfg(16000, 1.0, 0.8, 0.5, False), bank=bank, device="cpu")
    random.seed(0)
    a = [[0.5, 0.6, 40, 100], [0.1, 0.2, 36, 0], [0.2, 0.3, 40, 64.5], [0.3, 1.4, 36, 127]]
    plan = sd.plan([a, [], a[:1]])
    assert list(plan.clip_note_off) == [0, 4, 4, 5]
    assert list(plan.notes["track"][:4]) == [0, 0, 1, 1]            # grouped by track, first-appearance order
    assert list(plan.notes["start"][:4]) == [8000, 3200, 1600, 4800]
    assert plan.notes["vol"][2] == 0.0 and plan.clip_gain[1] == 0.0
    assert plan.clip_len[1] == 16000 and plan.clip_len[0] == int(np.float32(np.float32(1.4) + np.float32(0.1)) * np.float32(16000))
    assert plan.notes["vol"][1] == vel_to_vol(64.5)
    with pytest.raises(ValueError, match="Invalid note"):
        sd.plan([[[0.5, 0.4, 40, 100]]])
    with pytest.raises(ValueError, match="Invalid note"):
        sd.plan([[[0.5, 0.6, 62, 100]]])
    with pytest.raises(ValueError, match="sample rate"):           # the FX chain needs >= 12544 Hz (64-sample reverb chunks)
        SynthDrum(SynthDrumConfig(**{**make_cfg(16000, 1.0, 0.8, 0.5, False).__dict__, "use_fx_prob": 0.3, "sample_rate": 8000}), bank=bank)


def test_threshold_groups(golden):
    _, _, bank = golden
    for thr, n in [(1.0, 1), (0.95, 2), (0.8, 3), (0.25, 9), (0.0, 11)]:
        assert len(SynthDrum(make_cfg(16000, 1.0, thr, 0.5, False), bank=bank, device="cpu").tolerance_thr_to_h5_group()) == n


def test_bank_roundtrip(tmp_path, golden):
    _, tree, bank = golden
    p = str(tmp_path / "bank@16000.npz")
    bank.save(p)
    b2 = OneShotBank.load(p)
    assert np.array_equal(b2.data, bank.data) and b2.names == bank.names and b2.cells == bank.cells
    i = bank.shot_id(40, "gold", bank.cell_names(40, "gold")[1])
    assert np.array_equal(bank.shot(i), tree["40"]["gold"][bank.names[i]])


# ------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_matches_reference_golden(golden):
    g, tree, bank = golden
    for c, cfg, notes, seed in golden_cases(g):
        sd = SynthDrum(cfg, bank=bank, device="cuda:0")
        random.seed(seed)
        wav = sd(notes.tolist()).cpu().numpy()
        ref = g[f"s{c}_wav"]
        assert wav.shape == ref.shape
        assert np.abs(wav - ref).max() <= 1e-6, f"case {c}"


def random_batch(rng, B, input_sec, adtof):
    batch = []
    for b in range(B):
        n = int(rng.integers(0, 40)) if b % 7 else 0
        onset = np.sort(rng.uniform(0, input_sec * 0.98, n))
        pitch = rng.choice([35, 38, 41, 42, 48, 52, 58, 61], n) if adtof else rng.integers(35, 61, n)
        vel = rng.integers(0, 128, n)
        batch.append([[float(onset[i]), float(onset[i] + 0.1), float(pitch[i]), float(vel[i])] for i in range(n)])
    return batch


@pytest.mark.gpu
@pytest.mark.parametrize("adtof", [False, True])
def test_gpu_batch_vs_oracle(golden, adtof):
    _, tree, bank = golden
    cfg = make_cfg(16000, 2.0, 0.8, 0.8, adtof)
    sd = SynthDrum(cfg, bank=bank, device="cuda:0")
    rng = np.random.default_rng(5 + adtof)
    batch = random_batch(rng, 24, 2.0, adtof)
    random.seed(123)
    plan = sd.plan(batch)
    wavs = sd.render_plan(plan).cpu().numpy()
    assert wavs.shape == (24, plan.width)
    again = sd.render_plan(plan).cpu().numpy()
    assert np.array_equal(wavs, again, equal_nan=True)              # atomics only carry a max: bit-reproducible
    for i, notes in enumerate(batch):
        ref = oracle_from_plan(plan, i, notes, cfg, tree).numpy()
        W = int(plan.clip_len[i])
        assert ref.shape[0] == W
        if np.isnan(ref).any():
            assert np.isnan(wavs[i, :W]).all()
            continue
        assert np.abs(wavs[i, :W] - ref).max() <= 1e-6, f"clip {i}"
        assert np.all(wavs[i, W:] == 0)                             # collate-style zero padding


@pytest.mark.gpu
def test_gpu_full_size_batch(golden):
    """Training-size batch: 64 clips x 10 s @ 16 kHz, ~40 notes each.  Peak-normalised
    (max |wav| == clip gain), empty clips all zero, equal to each clip rendered alone."""
    _, tree, bank = golden
    cfg = make_cfg(16000, 10.0, 0.8, 0.8, False)
    sd = SynthDrum(cfg, bank=bank, device="cuda:0")
    rng = np.random.default_rng(9)
    batch = []
    for b in range(64):
        n = 0 if b % 20 == 19 else 40
        onset = np.sort(rng.uniform(0, 2.95, n))
        batch.append([[float(onset[i]), float(onset[i] + 0.1), float(rng.integers(35, 61)), float(rng.integers(1, 128))]
                      for i in range(n)])
    random.seed(1)
    plan = sd.plan(batch)
    wavs = sd.render_plan(plan, width=160000)
    assert wavs.shape == (64, 160000)
    peak = wavs.abs().amax(dim=1).cpu().numpy()
    for b in range(64):
        if not batch[b]:
            assert peak[b] == 0
        else:
            assert abs(peak[b] - plan.clip_gain[b]) <= 1e-6
    for b in (0, 33):
        ref = oracle_from_plan(plan, b, batch[b], cfg, tree).numpy()
        assert np.abs(wavs[b, : ref.shape[0]].cpu().numpy() - ref).max() <= 1e-6


@pytest.mark.gpu
def test_c_abi_rejects_small_workspace():
    from adt_str_amd import _ffi
    x = torch.zeros(64, device="cuda:0")
    with pytest.raises(_ffi.AdtError) as e:
        _ffi.call("adt_mix_render_f32", x.data_ptr(), x.data_ptr(), 1, x.data_ptr(), 4, x.data_ptr(), x.data_ptr(),
                  x.data_ptr(), 1, 16, x.data_ptr(), 16, x.data_ptr(), 8, 0)
    assert e.value.code == -1 and "workspace" in str(e.value)
