"""Fixture loading + comparison helpers shared by the test modules."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        self.meta = json.loads(str(z["meta"]))
        self.groups = {}
        for k in z.files:
            if k == "meta":
                continue
            g, n = k.split("/", 1)
            self.groups.setdefault(g, {})[n] = torch.from_numpy(np.array(z[k]))

    def __getattr__(self, g):
        try:
            return self.__dict__["groups"][g]
        except KeyError:
            raise AttributeError(g)

    @property
    def inp(self):
        return self.groups["in"]


def rel_err(a: torch.Tensor, b: torch.Tensor, floor: float = 1e-6) -> float:
    """max|a-b| / max(max|b|, floor) -- the 'relative fp32' measure BASELINE.json's 1e-3 tolerance is stated in.
    The floor keeps analytically-zero tensors (e.g. d k_proj.bias: softmax is invariant to a per-row score shift)
    from turning 1e-10 round-off into a relative error."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), floor)


def elementwise_err(a: torch.Tensor, b: torch.Tensor, floor_frac: float = 1e-3, abs_floor: float = 1e-6) -> float:
    """max over elements of |a - b| / (|b| + floor_frac * max|b|): an element-wise relative error (rel_err above is a max-norm measure:
    a small element of a tensor with a large maximum can be wrong by many times its own size and pass).  The floor -- a fraction of the
    tensor's largest magnitude -- keeps elements that are analytically ~0 from dividing round-off by nothing."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    floor = max(floor_frac * b.abs().max().item(), abs_floor)      # abs_floor: analytically-zero tensors (d k_proj.bias) are round-off on both sides
    return ((a - b).abs() / (b.abs() + floor)).max().item()


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
    return e


# ------------------------------------------------------------------------------------------ tiny model builders
def tiny_opt_config(pre_ln=True, proj=None, dropout=0.1):
    from transformers import OPTConfig
    return OPTConfig(vocab_size=128, hidden_size=64, num_attention_heads=4, ffn_dim=128, num_hidden_layers=4,
                     max_position_embeddings=64, word_embed_proj_dim=proj or 64, do_layer_norm_before=pre_ln,
                     dropout=dropout, attention_dropout=0.0, pad_token_id=1, bos_token_id=2, eos_token_id=2, init_std=0.08)


def tiny_roberta_config():
    from transformers import RobertaConfig
    return RobertaConfig(vocab_size=128, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                         max_position_embeddings=40, pad_token_id=1, type_vocab_size=1, hidden_dropout_prob=0.0,
                         attention_probs_dropout_prob=0.0)


def tiny_clip_vision_config():
    from transformers import CLIPVisionConfig
    return CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, image_size=32,
                            patch_size=16)


def mpt_args(**kw):
    from types import SimpleNamespace
    a = dict(context="all", neighbor_mode="cross_attention", n_text_tokens=2, n_visual_tokens=2, text_model="roberta-tiny",
             visual_model="clip-vit-tiny", max_output_length=8, freeze_lm=False, model_name_or_path="opt-tiny",
             peft_type="flamingo", lora_r=4, lora_alpha=1.0, lora_dropout=0.0, neighbor_layer_wise=2, decoder_only=True,
             position_type="none", max_text_neighbors=3, max_image_neighbors=2)
    a.update(kw)
    return SimpleNamespace(**a)


def load_exact(module, state, prefix=""):
    """load_state_dict(strict) from a fixture group, ignoring fixture keys outside `prefix`."""
    sd = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    missing = [k for k in missing if "position_ids" not in k]
    assert not missing, f"missing keys: {missing[:5]}"
    assert not [k for k in unexpected if "position_ids" not in k], f"unexpected keys: {unexpected[:5]}"
