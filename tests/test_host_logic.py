"""Host-side integer / layout code against the reference's golden vectors (no GPU):
tokenizer (a3), collate + masks (a4, a5), pitch maps."""
import os

import numpy as np
import pytest
import torch

from adt_str_amd.data import collate_fn, notes_from_bytes
from adt_str_amd.masks import create_mask_plain
from adt_str_amd.tokenizer import MidiTokenizer, MidiTokenizerConfig


@pytest.fixture(scope="module")
def tk_golden(golden_dir):
    return np.load(os.path.join(golden_dir, "tokenizer.npz"))


def test_tokenizer_matches_reference(tk_golden):
    g = tk_golden
    for c in range(int(g["n_cases"])):
        adtof, add_vel = (bool(x) for x in g[f"c{c}_cfg"])
        tk = MidiTokenizer(MidiTokenizerConfig(adtof, 2, 3, 1, 0, add_vel))
        notes = g[f"c{c}_notes"]
        mapped = tk.map_notes_to_Gm_custom(torch.from_numpy(notes.copy())).numpy() if len(notes) else notes
        assert np.array_equal(mapped, g[f"c{c}_mapped"])
        toks = tk.notes_to_adt_tokens(torch.from_numpy(mapped.copy())).numpy()
        assert np.array_equal(toks, g[f"c{c}_tokens"]), c
        dec = tk.decode(toks.tolist()).numpy().reshape(-1, 4)
        assert np.allclose(dec, g[f"c{c}_decoded"].reshape(-1, 4))
    assert np.array_equal(MidiTokenizer(MidiTokenizerConfig(False, 2, 3, 1, 0, True)).empty_adt_tokens().numpy(), g["empty_tokens"])


def test_decode_of_malformed_sequences(tk_golden):
    g = tk_golden
    for i in range(int(g["n_bad"])):
        for add_vel in (0, 1):
            tk = MidiTokenizer(MidiTokenizerConfig(False, 2, 3, 1, 0, bool(add_vel)))
            dec = tk.decode(g[f"bad{i}_{add_vel}_tokens"].tolist()).numpy().reshape(-1, 4)
            assert np.allclose(dec, g[f"bad{i}_{add_vel}_decoded"].reshape(-1, 4)), (i, add_vel)


def test_time_token_range_assertion():
    tk = MidiTokenizer(MidiTokenizerConfig(False, 2, 3, 1, 0, True))
    with pytest.raises(AssertionError, match="Time token is out of range"):
        tk.notes_to_adt_tokens(torch.tensor([[2.97, 3.07, 36.0, 100.0]]))
    assert tk.notes_to_adt_tokens(torch.tensor([[2.95, 3.05, 36.0, 100.0]])).tolist() == [2, 299, 336, 500, 3]


def test_masks_and_collate_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "masks_collate.npz"))
    for i in range(int(g["n_masks"])):
        cm, pm = create_mask_plain(int(g[f"m{i}_T"]), torch.from_numpy(g[f"m{i}_lens"]))
        assert np.array_equal(cm.numpy(), g[f"m{i}_causal"]) and np.array_equal(pm.numpy(), g[f"m{i}_pad"])
    assert create_mask_plain(4)[1] is None
    for i in range(int(g["n_collate"])):
        batch = [(torch.from_numpy(g[f"c{i}_wav{j}"]), g[f"c{i}_tok{j}"].tolist()) for j in range(int(g[f"c{i}_n"]))]
        out = collate_fn(batch)
        assert np.array_equal(out["wavs"].numpy(), g[f"c{i}_wavs"])
        assert np.array_equal(out["tokens"].numpy(), g[f"c{i}_tokens"]) and out["tokens"].dtype == torch.int64
        assert np.array_equal(out["token_lengths"].numpy(), g[f"c{i}_token_lengths"])


def test_notes_row_schema():
    a = np.array([[0.1, 0.2, 36, 100], [0.5, 0.6, 42, 64]], np.float32)
    assert torch.equal(notes_from_bytes(a.tobytes()), torch.from_numpy(a))
