"""-m gpu: BASELINE.json's acceptance check as a test -- "CIDEr within +-0.2 of reference" for the bf16 HIP path against the fp32
CPU oracle on the same weights and batches, through evaluate_loop's own slicing / argmax / decoding / scoring
(reference language_modelling/run_generation.py:584-606, 624-629, 668-671).

There are no pretrained weights here (no network), and a random-init LM predicts near-uniformly: its argmax is decided by the
fifth significant digit, so agreement between ANY two implementations would be a coin toss and both CIDEr scores ~0.  The test
therefore builds a model that is CONFIDENT by construction at the real OPT-125m dimensions (d = 768, 12 heads of 64, 12 + 4
layers): the tied token embedding is scaled up so that the residual stream is dominated by the current token and the tied
lm_head echoes it, while every self-attention / gated cross-attention / FFN layer (gates 0.5) still perturbs the logits.  The
predicted captions then share most words with the reference summaries (shifted by one token), CIDEr is far from zero, and a
numerical defect anywhere in the path moves both the argmax agreement and the score."""
import os

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

N_SAMPLES = 32


class _OracleModel(nn.Module):
    """The CPU oracle behind the model(**batch) -> (.loss, .logits) surface evaluate_loop drives."""

    def __init__(self, hip_model, lm_cfg, args):
        super().__init__()
        from oracle import lm_ref
        self.anchor = nn.Parameter(torch.zeros(1))                      # evaluate_loop asks next(model.parameters()).device
        self.sd = {k: v.detach().float().cpu() for k, v in hip_model.state_dict().items()}
        self.text_model, self.visual_model = hip_model.text_model, hip_model.visual_model      # HF modules (fp32, CPU at this point)
        self.cfg = lm_ref.LMConfig(vocab_size=lm_cfg.vocab_size, hidden_size=lm_cfg.hidden_size, num_attention_heads=lm_cfg.num_attention_heads,
                                   ffn_dim=lm_cfg.ffn_dim, num_hidden_layers=lm_cfg.num_hidden_layers, word_embed_proj_dim=lm_cfg.word_embed_proj_dim,
                                   neighbor_layer_wise=args.neighbor_layer_wise, pad_token_id=lm_cfg.pad_token_id)
        self.args = args
        self.cache = []

    def encode(self, batch):
        with torch.no_grad():
            L = batch["neighbor_input_ids"].shape[-1]
            tl = self.text_model(input_ids=batch["neighbor_input_ids"].reshape(-1, L),
                                 attention_mask=batch["neighbor_attention_mask"].reshape(-1, L)).last_hidden_state
            im = batch["neighbor_images"]
            vp = self.visual_model(im.reshape(-1, *im.shape[-3:])).pooler_output
        return tl, vp

    def forward(self, **batch):
        tl, vp = self.cache.pop(0) if self.cache else self.encode(batch)
        return self.run(batch, tl, vp)

    def run(self, batch, tl, vp):
        from types import SimpleNamespace
        from oracle import wrapper_ref
        with torch.no_grad():
            logits, loss = wrapper_ref.cross_attention_model_forward(self.sd, self.cfg, batch, tl, vp, "all", self.args.n_text_tokens)
        return SimpleNamespace(logits=logits, loss=loss)


def test_cider_and_argmax_bf16_hip_vs_fp32_oracle(tmp_path):
    from torch.utils.data import DataLoader, Subset
    from mmgl_amd.language_modelling.run_generation import Arguments, build_datasets, build_model, evaluate_loop, _summary_slices
    from mmgl_amd.wikiweb2m.synthetic import synthetic_tokenizer
    torch.manual_seed(7)
    tokenizer = synthetic_tokenizer()
    args = Arguments(model_name_or_path="mpt-125m", dataset="synthetic", context="all", neighbor_mode="embedding", peft_type="flamingo",
                     max_input_length=48, max_output_length=16, max_text_neighbors=5, max_image_neighbors=2, n_text_tokens=4,
                     n_visual_tokens=4, per_device_val_batch_size=4, dataloader_num_workers=0, val_steps_per_epoch=N_SAMPLES // 4,
                     print_freq=100, log_dir=str(tmp_path), seed=0)
    args.neighbor_layer_wise = 3
    args.image_size = 224
    with torch.device("cpu"):
        model = build_model(args, tokenizer, offline=True).float().eval()
    lm_cfg = model.lm.config
    with torch.no_grad():
        model.lm.model.decoder.embed_tokens.weight.mul_(50.0)           # tied with lm_head: the confident "echo" model of the docstring
        for n_, p in model.named_parameters():
            if n_.endswith("gating1") or n_.endswith("gating2"):
                p.fill_(0.5)
    assert model.lm.lm_head.weight.data_ptr() == model.lm.model.decoder.embed_tokens.weight.data_ptr()
    _, val_ds, _ = build_datasets(args, tokenizer)
    val_ds = Subset(val_ds, list(range(N_SAMPLES)))
    loader = lambda: DataLoader(val_ds, batch_size=4, shuffle=False, num_workers=0, drop_last=True)

    # fp32 CPU oracle through evaluate_loop, on the very same module parameters
    oracle = _OracleModel(model, lm_cfg, args)
    ref_tokens, ref_labels, enc = [], [], []
    for batch in loader():
        tl, vp = oracle.encode(batch)
        enc.append((tl, vp))                                            # re-used by the evaluate_loop pass below
        out = oracle.run(batch, tl, vp)
        lg, lb = _summary_slices(args, out.logits, batch["labels"])
        ref_tokens.append(lg.argmax(-1))
        ref_labels.append(lb)
    oracle.cache = list(enc)
    args.fuse_eval_batches = False                                      # the oracle's encoder cache is per validation batch
    evaluate_loop(loader(), oracle, tokenizer, 0, args, prefix="oracle-fp32")
    ref = dict(evaluate_loop.last)
    args.fuse_eval_batches = True                                       # the HIP pass below: all 8 batches share one forward (default)

    # the bf16 HIP path through the same loop
    hip = model.to(torch.bfloat16).cuda().eval()
    evaluate_loop(loader(), hip, tokenizer, 0, args, prefix="hip-bf16")
    got = dict(evaluate_loop.last)
    agree, total = 0, 0
    with torch.no_grad():
        for i, batch in enumerate(loader()):
            out = hip(**{k: v.cuda() for k, v in batch.items()})
            lg, lb = _summary_slices(args, out.logits, batch["labels"].cuda())
            real = (ref_labels[i] != tokenizer.pad_token_id)
            agree += int(((lg.argmax(-1).cpu() == ref_tokens[i]) & real).sum())
            total += int(real.sum())
    rate = agree / max(total, 1)
    print(f"CIDEr: HIP bf16 {got['cider']:.4f} vs CPU oracle fp32 {ref['cider']:.4f};  BLEU-4 {got['bleu4']:.4f} vs {ref['bleu4']:.4f};  "
          f"summary loss {got['loss']:.4f} vs {ref['loss']:.4f};  teacher-forced argmax agreement {rate:.4f} over {total} summary tokens")
    assert ref["cider"] > 0.5, "the constructed model must produce captions that overlap the references (else the check is vacuous)"
    assert abs(got["cider"] - ref["cider"]) <= 0.2, (got["cider"], ref["cider"])          # BASELINE.json: CIDEr within +-0.2
    assert rate >= 0.97, rate
    assert abs(got["loss"] - ref["loss"]) <= 2e-2 * abs(ref["loss"]), (got["loss"], ref["loss"])
