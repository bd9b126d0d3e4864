"""CPU tests: host-side rows against golden vectors produced by the reference (tests/golden/make_golden_host.py)."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import GOLDEN


def data_args(**kw):
    a = dict(task="section", context="all", decoder_only=True, neighbor_mode="embedding", max_text_neighbors=5,
             max_image_neighbors=2, position_type="none", max_input_length=32, max_output_length=12, n_text_tokens=2,
             n_visual_tokens=2)
    a.update(kw)
    return SimpleNamespace(**a)


DATA_CASES = {
    "emb_all_dec": dict(),
    "emb_all_encdec": dict(decoder_only=False),
    "emb_all_dec_wide": dict(max_text_neighbors=11, max_image_neighbors=5, max_input_length=48),
    "raw_section_only": dict(neighbor_mode="raw", context="section_only"),
    "raw_text_only": dict(neighbor_mode="raw", context="text_only"),
    "raw_text_only_encdec": dict(neighbor_mode="raw", context="text_only", decoder_only=False),
}


@pytest.mark.parametrize("case", sorted(DATA_CASES))
def test_g6_dataset_items_bit_exact(case):
    from mmgl_amd.wikiweb2m import WikiWeb2M
    from mmgl_amd.wikiweb2m.synthetic import synthetic_id_list, synthetic_pages, synthetic_tokenizer
    z = np.load(os.path.join(GOLDEN, "g6_data.npz"))
    df = synthetic_pages(4, seed=3)
    ids = synthetic_id_list(df)
    ds = WikiWeb2M(data_args(**DATA_CASES[case]), df, ids, synthetic_tokenizer(), None)
    assert len(ds) == len(ids)
    for i in range(len(ids)):
        item = ds[i]
        want = {k.split("/", 2)[2]: z[k] for k in z.files if k.startswith(f"{case}/{i}/")}
        assert set(item) == set(want), (case, i, set(item) ^ set(want))
        for k, v in want.items():
            got = item[k].numpy()
            assert got.dtype == v.dtype and got.shape == v.shape, (case, i, k, got.dtype, v.dtype, got.shape, v.shape)
            assert np.array_equal(got, v), (case, i, k)


def test_dataset_default_collate_stacks():
    from mmgl_amd.wikiweb2m import WikiWeb2M
    from mmgl_amd.wikiweb2m.synthetic import synthetic_id_list, synthetic_pages, synthetic_tokenizer
    df = synthetic_pages(3, seed=5)
    ds = WikiWeb2M(data_args(position_type="laplacian"), df, synthetic_id_list(df), synthetic_tokenizer(), None)
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=3)))
    assert batch["input_ids"].shape == (3, 44) and batch["neighbor_input_ids"].shape == (3, 5, 32)
    assert batch["lpe"].shape == (3, 8, 3) and batch["neighbor_images"].shape == (3, 2, 3, 224, 224)
    # slot 0 (page info) is always a valid text neighbor; padding slots come last
    assert (batch["neighbor_pos_ids"][:, 0] == 1).all() and (batch["text_locations"][:, 0] == 0).all()
    locs = torch.cat([batch["text_locations"], batch["image_locations"]], dim=1).sort(dim=1).values
    assert torch.equal(locs, torch.arange(7).expand(3, -1))


def test_graph_pe_shapes_and_properties():
    from mmgl_amd.wikiweb2m import graph_pe
    edges = torch.tensor([[0, 0, 1, 2, 3], [1, 2, 2, 3, 4]])
    lpe = graph_pe.compute_LPE(edges, 17)
    assert lpe.shape == (17, 12) and torch.isfinite(lpe).all()
    assert (lpe[5:] == 0).all()                      # isolated (padding) nodes
    g = graph_pe.normalize_graph(graph_pe.dense_adjacency(edges, 6))
    assert torch.allclose(g, g.t()) and g.shape == (6, 6)
    A = graph_pe.dense_adjacency(edges, 6) + torch.eye(6, dtype=torch.float64)
    d = A.sum(1)
    assert torch.allclose(g.double(), A / (d[:, None] * d[None, :]).sqrt(), atol=1e-6)


def test_g7_cider_golden():
    from mmgl_amd.wikiweb2m.cider import Cider
    with open(os.path.join(GOLDEN, "g7_cider.json")) as f:
        cases = json.load(f)
    for name, c in cases.items():
        gts = {i: [r] for i, r in enumerate(c["refs"])}
        res = {i: [h] for i, h in enumerate(c["hyps"])}
        score, scores = Cider().compute_score(gts, res)
        assert abs(score - c["score"]) < 1e-12, name
        assert np.allclose(scores, c["scores"], rtol=0, atol=1e-12), name
    # SURVEY.md known-answer test: candidates == references on a 3-sentence corpus
    refs = ["the cat sat on the mat", "a dog barks at the mailman", "birds fly south in winter"]
    s, _ = Cider().compute_score({i: [r] for i, r in enumerate(refs)}, {i: [r] for i, r in enumerate(refs)})
    assert abs(s - 10.0) < 1e-9
