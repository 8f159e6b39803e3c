"""Drop-in surface that needs no GPU: import paths, config merge, state-dict contract,
WAV/MIDI I/O, training-argument factory."""
import os
import struct

import numpy as np
import pytest
import torch


def test_reference_import_paths_resolve():
    import build_model, config, inference, model, train  # noqa: F401
    from data_modules.train_dataset import LakhDataset, collate_fn  # noqa: F401
    from modules.midi_tokenizer import MidiTokenizer, MidiTokenizerConfig  # noqa: F401
    from modules.synthetiser import SynthDrum, SynthDrumConfig  # noqa: F401
    from modules.clap_encoder import ClapWrapper  # noqa: F401
    from utils.config_utils import deep_merge_dicts, load_config_from_yaml  # noqa: F401
    from utils.mapping_utils import MappingUtils
    from utils.utils import create_mask_plain  # noqa: F401
    from utils.audio_utils import load_and_resample, normalize, resample  # noqa: F401
    assert MappingUtils().ADTOF_label_mapping[42] == "HH"
    assert model.ADTModel.config_class is config.ADTModelConfig and config.ADTModelConfig.model_type == "adt_model"


def test_config_merge_semantics(tmp_path):
    from adt_str_amd.config_utils import deep_merge_dicts, load_merged
    base = {"a": {"x": 1, "y": [1, 2]}, "b": 2}
    over = {"a": {"y": [3]}, "c": {"z": 1}}
    assert deep_merge_dicts(base, over) == {"a": {"x": 1, "y": [3]}, "b": 2, "c": {"z": 1}}
    assert base == {"a": {"x": 1, "y": [1, 2]}, "b": 2}                      # inputs untouched
    cfg = load_merged(os.path.join(os.path.dirname(__file__), "..", "configs", "train", "setting-1.yaml"))
    assert cfg["model"]["d_query"] == 128 and cfg["model"]["tgt_vocab_size"] == 1400 and cfg["training"]["weight_decay"] == 1e-5


def test_state_dict_contract_matches_reference(golden_dir):
    """Keys, order and shapes recorded from the reference's ADTModel (setting-1)."""
    from model import ADTModel, ADTModelConfig
    g = np.load(os.path.join(golden_dir, "adt_full_stats.npz"))
    m = ADTModel(ADTModelConfig(input_sec=2.56, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=4, dec_layers=4, nhead=6,
                                d_query=128, dropout=0.0, tgt_vocab_size=1400, plain=True, n_mels=128))
    sd = m.state_dict()
    assert list(sd) == [str(k) for k in g["state_keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["state_shapes"]]
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"]) == 69000824


def test_checkpoint_roundtrip_and_nested_formats(tmp_path):
    import yaml
    from safetensors.torch import save_file
    from build_model import load_checkpoint_state
    from model import ADTModel, ADTModelConfig
    m = ADTModel(ADTModelConfig(input_sec=0.5, time_res=0.01, win_length=2048, sample_rate=16000, enc_layers=1, dec_layers=1, nhead=1,
                                d_query=128, dropout=0.0, tgt_vocab_size=1400, plain=True, n_mels=128))
    d1 = tmp_path / "st"; d1.mkdir()
    save_file({k: v.contiguous() for k, v in m.state_dict().items()}, str(d1 / "model.safetensors"))
    s1 = load_checkpoint_state(str(d1))
    assert all(torch.equal(s1[k], v) for k, v in m.state_dict().items())
    d2 = tmp_path / "pt"; d2.mkdir()
    torch.save({"model_state_dict": m.state_dict()}, str(d2 / "pytorch_model.bin"))
    assert set(load_checkpoint_state(str(d2))) == set(m.state_dict())
    with pytest.raises(FileNotFoundError):
        load_checkpoint_state(str(tmp_path))


def test_wav_and_midi_io(tmp_path):
    from adt_str_amd.audio_io import read_wav, write_drum_midi, write_wav
    x = (np.sin(np.arange(4000) * 0.05) * 0.5).astype(np.float32)
    p = str(tmp_path / "a.wav")
    write_wav(p, np.stack([x, -x]), 16000)
    y, sr = read_wav(p)
    assert sr == 16000 and y.shape == (2, 4000) and np.abs(y[0] - x).max() < 1e-4 and np.abs(y[1] + x).max() < 1e-4
    f32 = str(tmp_path / "f.wav")                                              # IEEE float WAV
    body = x.astype("<f4").tobytes()
    with open(f32, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 1, 24000, 96000, 4, 32) +
                 b"data" + struct.pack("<I", len(body)) + body)
    z, sr2 = read_wav(f32)
    assert sr2 == 24000 and np.array_equal(z[0], x)
    mid = str(tmp_path / "a.mid")
    write_drum_midi(mid, [[0.5, 0.6, 36, 100], [0.0, 0.1, 42, 64]])
    raw = open(mid, "rb").read()
    assert raw[:4] == b"MThd" and raw[14:18] == b"MTrk" and raw.endswith(b"\x00\xff\x2f\x00")
    assert bytes([0x99, 42, 64]) in raw and bytes([0x99, 36, 100]) in raw and raw.index(bytes([0x99, 42, 64])) < raw.index(bytes([0x99, 36, 100]))


def test_training_arguments_factory():
    import train
    from adt_str_amd.config_utils import load_merged
    cfg = load_merged(os.path.join(os.path.dirname(__file__), "..", "configs", "train", "setting-1.yaml"))
    a = train.create_training_arguments(cfg)
    assert a.per_device_train_batch_size == 64 and a.learning_rate == 1e-4 and a.max_grad_norm == 1.0 and a.weight_decay == 1e-5
    assert a.ddp_broadcast_buffers is False and a.remove_unused_columns is False


def test_chunking_matches_reference_rule():
    from inference import _chunk_audio
    c = _chunk_audio(torch.arange(10.0), 4)
    assert c.shape == (3, 4) and c[2].tolist() == [8.0, 9.0, 0.0, 0.0]
    assert _chunk_audio(torch.arange(8.0), 4).shape == (2, 4)


def test_inference_cli_accepts_the_references_flag_spellings():
    """reference inference.py:52-68: ``input_path config_path [-o | --output_path DIR] [-s | --synthetise_transcription]``, default
    output directory ``outputs/``."""
    from inference import _parser
    a = _parser().parse_args(["in.wav", "cfg.yaml", "--output_path", "o", "--synthetise_transcription"])
    assert (a.input_path, a.config_path, a.output_dir, a.synthesize) == ("in.wav", "cfg.yaml", "o", True)
    b = _parser().parse_args(["in.wav", "cfg.yaml", "-o", "p", "-s"])
    assert (b.output_dir, b.synthesize) == ("p", True)
    c = _parser().parse_args(["in.wav", "cfg.yaml"])
    assert (c.output_dir, c.synthesize) == ("outputs/", False)


def test_bench_clock_poll_is_silent_without_a_gpu():
    """bench.py samples the shader clock from sysfs during its timed steps; where there is no amdgpu node (this container: no GPU) the
    line simply goes out without the `clock` object -- no exception, no thread, no hang at exit."""
    import time
    import bench
    poll = bench.ClockPoll(0).start()
    time.sleep(0.2)
    assert poll._th is None and poll.stop(0.0) is None


def test_bench_clock_poll_reads_sysfs_in_process(tmp_path, monkeypatch):
    """The poll never starts a child process (a spawned rocm-smi under rocprofv3 is the exec-after-GPU-init hop the pool forbids): it reads the
    card's hwmon node -- or the pp_dpm_sclk level table -- itself.  A fake sysfs tree with two cards: index = PCI order."""
    import subprocess
    import time
    import bench
    for i, (bus, hz, uw) in enumerate([("0000:0a:00.0", 2110000000, 1203000000), ("0000:05:00.0", 1890000000, 990000000)]):
        pci = tmp_path / "pci" / bus
        hw = pci / "hwmon" / "hwmon3"
        hw.mkdir(parents=True)
        (hw / "freq1_input").write_text(f"{hz}\n")
        (hw / "power1_average").write_text(f"{uw}\n")
        (tmp_path / "drm" / f"card{i}").mkdir(parents=True)
        (tmp_path / "drm" / f"card{i}" / "device").symlink_to(pci)
    legacy = tmp_path / "pci" / "0000:0f:00.0"
    legacy.mkdir()
    (legacy / "pp_dpm_sclk").write_text("0: 500Mhz\n1: 2100Mhz *\n2: 2400Mhz\n")
    (tmp_path / "drm" / "card2").mkdir()
    (tmp_path / "drm" / "card2" / "device").symlink_to(legacy)
    monkeypatch.setattr(bench.ClockPoll, "SYSFS_DRM", str(tmp_path / "drm"))
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("the clock poll must not spawn a process")))
    monkeypatch.setattr(subprocess, "Popen", lambda *a, **k: (_ for _ in ()).throw(AssertionError("the clock poll must not spawn a process")))
    got = {}
    for index in (0, 1, 2, 3):
        poll = bench.ClockPoll(index).start()
        time.sleep(0.1)
        got[index] = poll.stop(0.0)
    assert got[0]["sclk_mhz"]["median"] == 1890 and abs(got[0]["power_w_median"] - 990.0) < 1e-6        # PCI order: 05 before 0a
    assert got[1]["sclk_mhz"]["median"] == 2110 and got[1]["samples"] >= 2 and abs(got[1]["held_over_nominal"] - 2110 / 2400) < 1e-9
    assert got[2]["sclk_mhz"]["median"] == 2100 and got[2]["power_w_median"] is None        # level table, no power node
    assert got[3] is None
    # a GPU box shows every GPU of its host in sysfs while the process sees one: the card is picked by the HIP device's PCI address
    assert os.path.realpath(bench.ClockPoll.find_nodes(0, "0000:0a:00")[0]).endswith("0000:0a:00.0/hwmon/hwmon3/freq1_input")
    assert os.path.realpath(bench.ClockPoll.find_nodes(0, "0000:ff:00")[0]).endswith("0000:05:00.0/hwmon/hwmon3/freq1_input")   # unknown address: PCI order
