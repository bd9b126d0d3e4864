"""Golden vectors for the host-side rows: WikiWeb2M example construction (G6) and SelfAttentionModel fusion (G9),
produced by importing the reference with stubs for the packages this image lacks (torch_geometric, nltk, peft)."""
import importlib.util
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


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


def golden_data():
    from mmgl_amd.wikiweb2m.synthetic import synthetic_id_list, synthetic_pages, synthetic_tokenizer
    _stub("torch_geometric")
    _stub("torch_geometric.data", Data=lambda **k: SimpleNamespace(**k))
    pkg = _stub("language_modelling")
    pkg.__path__ = []
    pkg.utils = _stub("language_modelling.utils", get_feature_extractor_for_model=lambda n: None,
                      get_pixel_values_for_model=lambda fe, img: None)
    ref = _load("ref_data", f"{REF}/wikiweb2m/data.py")
    df = synthetic_pages(4, seed=3)
    ids = synthetic_id_list(df)
    tok = synthetic_tokenizer()
    flat = {}
    for case, kw in DATA_CASES.items():
        ds = ref.WikiWeb2M(data_args(**kw), df, ids, tok, None)
        for i in range(len(ids)):
            item = ds[i]
            for k, v in item.items():
                flat[f"{case}/{i}/{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "g6_data.npz"), **flat)
    print(f"wrote g6_data.npz: {len(flat)} arrays, {os.path.getsize(os.path.join(HERE, 'g6_data.npz'))/1024:.1f} KiB")
    golden_data_with_images(ref, df, ids, tok)


IMAGE_CASES = {
    # the with-image slot order of get_embedding_item (wikiweb2m/data.py:363-420) and the raw-mode image splicing (:158-240)
    "emb_all_dec_img": dict(),
    "emb_all_dec_wide_img": dict(max_text_neighbors=11, max_image_neighbors=5, max_input_length=48),
    "emb_all_dec_tight_img": dict(max_text_neighbors=3, max_image_neighbors=1),
    "emb_all_encdec_img": dict(decoder_only=False),
    "raw_section_all_img": dict(neighbor_mode="raw", context="section_all"),
    "raw_all_img": dict(neighbor_mode="raw", context="all", max_input_length=96),
    "raw_all_encdec_img": dict(neighbor_mode="raw", context="all", max_input_length=96, decoder_only=False),
}


def golden_data_with_images(ref, df, ids, tok):
    """G6 with image neighbors: the reference checks `self.image_path/{page}_{section}_{idx}.{ext}` and then opens
    `./wikiweb2m/raw/images/...` relative to the working directory (data.py:136-140) -- so the synthetic image files go into a temp
    `wikiweb2m/raw/images`, the process chdir()s next to it, `image_path` points at it, and the (absent) HF feature extractor is
    stubbed by mmgl_amd.wikiweb2m.synthetic.synthetic_pixel_values.  Pixel tensors are stored as float16-exact values
    (k / 255 is not, so they are compared through their uint8 source: stored as uint8 * 1, see below)."""
    import tempfile
    from mmgl_amd.wikiweb2m.synthetic import synthetic_images, synthetic_pixel_values
    sys.modules["language_modelling.utils"].get_pixel_values_for_model = lambda fe, img: synthetic_pixel_values(img)
    ref.utils.get_pixel_values_for_model = lambda fe, img: synthetic_pixel_values(img)
    flat, cwd = {}, os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        image_dir = os.path.join(tmp, "wikiweb2m", "raw", "images")
        written = synthetic_images(df, image_dir, seed=11)
        os.chdir(tmp)
        try:
            n_img = 0
            for case, kw in IMAGE_CASES.items():
                ds = ref.WikiWeb2M(data_args(**kw), df, ids, tok, None)
                ds.image_path = image_dir
                ds.visual_feature_extractor = object()           # the reference only sets it when given a model name (:62-63)
                for i in range(len(ids)):
                    item = ds[i]
                    for k, v in item.items():
                        a = v.numpy()
                        if a.dtype == np.float32 and a.ndim == 4:        # pixel stacks: k / 255 floats -> exact uint8 (4x smaller fixture)
                            q = np.rint(a * 255.0)
                            assert np.array_equal((q / 255.0).astype(np.float32), a)
                            a = q.astype(np.uint8)
                            n_img += int((a.reshape(a.shape[0], -1).max(1) > 0).sum())
                        flat[f"{case}/{i}/{k}"] = a
        finally:
            os.chdir(cwd)
    assert n_img > 20, n_img
    np.savez_compressed(os.path.join(HERE, "g6_data_images.npz"), **flat)
    print(f"wrote g6_data_images.npz: {len(flat)} arrays, {n_img} non-blank image slots, {sum(v == 'corrupt' for v in written.values())} corrupt "
          f"files of {len(written)}, {os.path.getsize(os.path.join(HERE, 'g6_data_images.npz'))/1024:.1f} KiB")


def golden_self_attention():
    """G9: SelfAttentionModel (OPT, peft none, neighbor_mode=embedding, context=all) with position_type none / laplacian."""
    from make_golden import make_batch, save, tiny_clip_vision_config, tiny_opt_config, tiny_roberta_config
    from transformers import CLIPVisionModel, OPTForCausalLM, RobertaModel
    _stub("peft", LoraConfig=None, PrefixTuningConfig=None, PromptTuningInit=None, PromptTuningConfig=None, TaskType=SimpleNamespace(
        SEQ_2_SEQ_LM=0, CAUSAL_LM=1), get_peft_model=None)
    pkg = _stub("ref_model")
    pkg.__path__ = [f"{REF}/model"]
    _load("ref_model.graph", f"{REF}/model/graph.py")
    sa = _load("ref_model.modelling_self_attention", f"{REF}/model/modelling_self_attention.py")
    oc, rc, vc = tiny_opt_config(), tiny_roberta_config(), tiny_clip_vision_config()
    for tag, ptype in (("none", "none"), ("laplacian", "laplacian")):
        torch.manual_seed(9)
        saved = (sa.AutoConfig.from_pretrained, sa.AutoModelForCausalLM.from_pretrained, sa.RobertaModel.from_pretrained,
                 sa.CLIPVisionModel.from_pretrained)
        sa.AutoConfig.from_pretrained = staticmethod(lambda name, *a, **k: rc if "roberta" in name else oc)
        sa.AutoModelForCausalLM.from_pretrained = staticmethod(lambda *a, **k: OPTForCausalLM(oc))
        sa.RobertaModel.from_pretrained = staticmethod(lambda *a, **k: RobertaModel(rc, add_pooling_layer=False))
        sa.CLIPVisionModel.from_pretrained = staticmethod(lambda *a, **k: CLIPVisionModel(vc))
        try:
            args = SimpleNamespace(context="all", decoder_only=True, neighbor_mode="embedding", position_type=ptype, n_text_tokens=2,
                                   n_visual_tokens=2, model_name_or_path="opt-tiny", peft_type="none", text_model="roberta-tiny",
                                   visual_model="clip-vit-tiny", max_output_length=8, max_text_neighbors=3, max_image_neighbors=2,
                                   freeze_lm=False, lora_r=4, lora_alpha=1, lora_dropout=0.0)
            w = sa.SelfAttentionModel(args, None)
        finally:
            (sa.AutoConfig.from_pretrained, sa.AutoModelForCausalLM.from_pretrained, sa.RobertaModel.from_pretrained,
             sa.CLIPVisionModel.from_pretrained) = saved
        w.eval()
        g = torch.Generator().manual_seed(91)
        b = make_batch(g)
        if ptype == "laplacian":
            b["lpe"] = torch.randn(2, 1 + 3 + 2, 1 + 3 + 2 - 5, generator=g)
        o = w(**b)
        o.loss.backward()
        grads = {k: v.grad for k, v in w.named_parameters() if v.grad is not None and not k.startswith("lm.")}
        save(f"g9_selfattn_{tag}.npz", dict(position_type=ptype), p=w.state_dict(), **{"in": b},
             out=dict(logits=o.logits, loss=o.loss), grad=grads)


def main():
    golden_data()
    golden_self_attention()


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    main()
