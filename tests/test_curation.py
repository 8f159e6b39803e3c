"""Curation (a10): host ranking/binning on CPU and the cosine/argmax kernel on the GPU, against the golden
produced by executing the reference's own lines (tests/golden/curation.npz)."""
import os

import numpy as np
import pytest
import torch

from adt_str_amd.curation import class_mean_embeddings, rank_assignments, score_to_bin_label
from oracle import curation as o_cur


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "curation.npz"))


def test_oracle_reproduces_reference_lines(g):
    for c in range(int(g["n_cases"])):
        res = o_cur.assign(torch.from_numpy(g[f"c{c}_samples"]), torch.from_numpy(g[f"c{c}_means"]), g[f"c{c}_labels"].tolist(),
                           int(g[f"c{c}_num_bins"]))
        assert [r[1] for r in res] == g[f"c{c}_order"].tolist()
        assert [r[0] for r in res] == g[f"c{c}_class"].tolist()
        assert [r[2] for r in res] == [str(b) for b in g[f"c{c}_bin"]]


def test_host_ranking_given_scores(g):
    """CPU-only: with exact per-class scores the host ranking reproduces order, class and bin."""
    for c in range(int(g["n_cases"])):
        x, m = torch.from_numpy(g[f"c{c}_samples"]), torch.from_numpy(g[f"c{c}_means"])
        sc = torch.stack([torch.nn.functional.cosine_similarity(x, e, dim=1) for e in m], dim=1).numpy()
        res = rank_assignments(sc.argmax(1), sc.max(1), g[f"c{c}_labels"].tolist(), int(g[f"c{c}_num_bins"]))
        assert res.order.tolist() == g[f"c{c}_order"].tolist()
        assert res.label.tolist() == g[f"c{c}_class"].tolist()
        assert res.bin == [str(b) for b in g[f"c{c}_bin"]]


def test_bin_labels_and_means():
    assert score_to_bin_label(1.0) == "100-90" and score_to_bin_label(0.6) == "90-80" and score_to_bin_label(-1.0) == "10-0"
    assert score_to_bin_label(0.79, 5) == "100-80" and score_to_bin_label(2.0) == "100-90"
    assert [score_to_bin_label(s) for s in (0.61, 0.59, 0.01, -0.01)] == [o_cur.bin_label(s, 10) for s in (0.61, 0.59, 0.01, -0.01)]
    with pytest.raises(ValueError):
        score_to_bin_label(0.5, 7)
    labels, means = class_mean_embeddings({35: [torch.ones(4), 3 * torch.ones(4)], 36: [], 421: [torch.zeros(4)]})
    assert labels == [35, 421] and torch.equal(means[0], 2 * torch.ones(4))
    with pytest.raises(RuntimeError):
        class_mean_embeddings({35: []})


@pytest.mark.gpu
def test_gpu_matches_reference_golden(g):
    from adt_str_amd.curation import assign
    for c in range(int(g["n_cases"])):
        x, m = torch.from_numpy(g[f"c{c}_samples"]).cuda(), torch.from_numpy(g[f"c{c}_means"]).cuda()
        res, sc = assign(x, m, g[f"c{c}_labels"].tolist(), int(g[f"c{c}_num_bins"]), return_scores=True)
        ref = torch.stack([torch.nn.functional.cosine_similarity(x.cpu(), e, dim=1) for e in m.cpu()], dim=1)
        assert (sc.cpu() - ref).abs().max() < 2e-6
        # identical assignment wherever the reference's top-2 margin exceeds fp32 summation noise
        top2 = ref.topk(2, dim=1).values
        clear = ((top2[:, 0] - top2[:, 1]) > 1e-5).numpy()
        by_sample_class = dict(zip(g[f"c{c}_order"].tolist(), g[f"c{c}_class"].tolist()))
        mine = dict(zip(res.order.tolist(), res.label.tolist()))
        assert all(mine[i] == by_sample_class[i] for i in range(len(clear)) if clear[i])
        assert sorted(res.order.tolist()) == list(range(len(clear)))
        # same copy order up to swaps between samples whose best scores are within noise
        ref_rank = {s: k for k, s in enumerate(g[f"c{c}_order"].tolist())}
        disp = max(abs(ref_rank[s] - k) for k, s in enumerate(res.order.tolist()))
        assert disp <= 2
        agree = np.mean([a == str(b) for a, b in zip(res.bin, g[f"c{c}_bin"])])
        assert agree > 0.99


@pytest.mark.gpu
def test_gpu_large_and_limits():
    from adt_str_amd import _ffi
    from adt_str_amd.curation import assign
    gen = torch.Generator().manual_seed(0)
    x = torch.nn.functional.normalize(torch.randn(100_000, 512, generator=gen), dim=1).cuda()
    m = torch.nn.functional.normalize(torch.randn(48, 512, generator=gen), dim=1).cuda() * 0.8
    res = assign(x, m, list(range(35, 82)) + [421])
    ref = (x @ torch.nn.functional.normalize(m, dim=1).t())
    assert (torch.from_numpy(res.score).cuda() - ref.max(1).values[torch.from_numpy(res.order).cuda()]).abs().max() < 2e-6
    assert np.all(np.diff(res.score) <= 0)                                  # descending copy order
    with pytest.raises(_ffi.AdtError):
        assign(x[:4], torch.randn(80, 512).cuda(), list(range(80)))        # class table larger than the LDS budget
