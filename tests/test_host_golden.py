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


IMAGE_CASES = {
    "emb_all_dec_img": dict(),
    "emb_all_dec_wide_img": dict(max_text_neighbors=11, max_image_neighbors=5, max_input_length=48),
    "emb_all_dec_tight_img": dict(max_text_neighbors=3, max_image_neighbors=1),
    "emb_all_encdec_img": dict(decoder_only=False),
    "raw_section_all_img": dict(neighbor_mode="raw", context="section_all"),
    "raw_all_img": dict(neighbor_mode="raw", context="all", max_input_length=96),
    "raw_all_encdec_img": dict(neighbor_mode="raw", context="all", max_input_length=96, decoder_only=False),
}


@pytest.mark.parametrize("case", sorted(IMAGE_CASES))
def test_g6_dataset_items_with_image_neighbors_bit_exact(case, tmp_path):
    """The WITH-IMAGE branch of the collate against the reference itself (wikiweb2m/data.py:118-144, 363-420; raw modes :158-240):
    slot order section image -> caption -> other sections' text / image / caption under the text / image caps, a corrupt file skipped
    in favour of the section's next image, blank pixels + position 0 for padding slots.  Fixture: tests/golden/make_golden_host.py ran
    the reference's WikiWeb2M over the same synthetic pages and image files (temp image dir + chdir, stubbed feature extractor)."""
    from mmgl_amd.wikiweb2m import WikiWeb2M
    from mmgl_amd.wikiweb2m.synthetic import (synthetic_id_list, synthetic_images, synthetic_pages, synthetic_pixel_values,
                                             synthetic_tokenizer)
    z = np.load(os.path.join(GOLDEN, "g6_data_images.npz"))
    df = synthetic_pages(4, seed=3)
    ids = synthetic_id_list(df)
    written = synthetic_images(df, str(tmp_path / "images"), seed=11)
    assert "corrupt" in written.values() and "image" in written.values()
    ds = WikiWeb2M(data_args(**IMAGE_CASES[case]), df, ids, synthetic_tokenizer(), synthetic_pixel_values, image_dir=str(tmp_path / "images"))
    ds.visual_feature_extractor = synthetic_pixel_values          # (the constructor only keeps it for context section_all / all, like the reference)
    n_img = 0
    for i in range(len(ids)):
        item = ds[i]
        want = {k.split("/", 2)[2]: z[k] for k in z.files if k.startswith(f"{case}/{i}/")}
        assert set(item) == set(want), (case, i, set(item) ^ set(want))
        for k, v in want.items():
            got = item[k].numpy()
            if v.dtype == np.uint8:                                # pixel stacks are stored as the exact uint8 they came from
                assert got.dtype == np.float32
                v = (v.astype(np.float32) / 255.0).astype(np.float32)
                n_img += int((v.reshape(v.shape[0], -1).max(1) > 0).sum())
            assert got.dtype == v.dtype and got.shape == v.shape, (case, i, k, got.dtype, v.dtype, got.shape, v.shape)
            assert np.array_equal(got, v), (case, i, k)
    assert n_img > 0, "the fixture case holds no image neighbor at all"


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


def test_on_disk_format_round_trip(tmp_path):
    """the files the reference's preprocess_data.py writes -- three parquet files with bytes / per-section array
    columns (:116-145), the id pickle keyed by split (:147-181), images named {page}_{section}_{idx}.{ext} (:201-202) -- read
    back through load_wikiweb2m and WikiWeb2M.  Text-only items must equal the in-memory ones bit for bit; a page that has an
    image file gets its pixels in the neighbor bundle at the slot its caption occupies (reference data.py:118-144, 363-381)."""
    import pickle
    from PIL import Image
    from mmgl_amd.wikiweb2m import WikiWeb2M
    from mmgl_amd.wikiweb2m.data import load_wikiweb2m
    from mmgl_amd.wikiweb2m.synthetic import synthetic_id_list, synthetic_pages, synthetic_tokenizer
    raw = tmp_path / "raw"
    (raw / "images").mkdir(parents=True)
    cols = ["page_id", "page_url", "page_title", "page_description", "section_title", "section_depth", "section_heading",
            "section_parent_index", "section_summary", "section_rest_sentence", "image_url", "image_caption"]
    splits = {"train": synthetic_pages(4, seed=3), "val": synthetic_pages(2, seed=4), "test": synthetic_pages(2, seed=5)}
    for name, df in splits.items():
        assert list(df.columns) == cols                                   # the schema of preprocess_data.py:120-121
        df.to_parquet(raw / f"wikiweb2m_{name}_large.parquet")
    with open(raw / "section_id_split_large.pkl", "wb") as f:
        pickle.dump({k: synthetic_id_list(df) for k, df in splits.items()}, f)
    train_df, val_df, test_df, id_list = load_wikiweb2m("section", root=str(raw))
    assert set(id_list) == {"train", "val", "test"} and len(train_df) == 4 and len(val_df) == 2 and len(test_df) == 2
    assert isinstance(train_df["page_title"].iloc[0], bytes) and isinstance(train_df["section_title"].iloc[0], np.ndarray)
    tok = synthetic_tokenizer()
    # (1) no image files yet: identical to the in-memory dataset (and therefore to the reference's golden items)
    ds_mem = WikiWeb2M(data_args(), splits["train"], id_list["train"], tok, None, image_dir=str(raw / "images"))
    ds_disk = WikiWeb2M(data_args(), train_df, id_list["train"], tok, None, image_dir=str(raw / "images"))
    for i in range(len(ds_mem)):
        a, b = ds_mem[i], ds_disk[i]
        assert set(a) == set(b)
        for k in a:
            assert torch.equal(a[k], b[k]), (i, k)
    # (2) an image for (page 1000, section 1, second url): {page}_{section}_{idx}.{ext of the url}
    page_id, section_id = id_list["train"][1]
    assert section_id == 1
    Image.fromarray((np.arange(20 * 30 * 3) % 255).astype(np.uint8).reshape(20, 30, 3)).save(raw / "images" / f"{page_id}_{section_id}_1.jpg")
    seen = []

    def extractor(img):                                                   # stands in for the CLIP feature extractor (no network)
        seen.append(img.size)
        return torch.full((3, 224, 224), 0.25)
    ds_img = WikiWeb2M(data_args(), train_df, id_list["train"], tok, extractor, image_dir=str(raw / "images"))
    with_img, without = ds_img[1], ds_disk[1]
    assert seen and seen[0] == (30, 20)
    assert int(with_img["neighbor_images_pos_ids"][0]) == 1 and int(without["neighbor_images_pos_ids"][0]) == 0
    assert torch.equal(with_img["neighbor_images"][0], torch.full((3, 224, 224), 0.25)) and float(without["neighbor_images"].abs().max()) == 0
    # the section image sits right after the page-info text and its caption follows it as a text neighbor (data.py:363-381)
    assert int(with_img["image_locations"][0]) == 1 and int(with_img["text_locations"][1]) == 2
    batch = next(iter(torch.utils.data.DataLoader(ds_img, batch_size=2)))
    assert batch["neighbor_images"].shape == (2, 2, 3, 224, 224)
