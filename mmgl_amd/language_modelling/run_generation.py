#!/usr/bin/env python
"""Fine-tuning / evaluation driver with the reference's `Arguments` surface and step order
(language_modelling/run_generation.py), re-wired for MI355X:

  * one process per GPU (torchrun or the built-in spawn), torch.distributed backend "nccl" == RCCL over xGMI;
    no hard-coded rendezvous (reference :283 pins tcp://127.0.0.1:1337);
  * `DataParallelEngine` (mmgl_amd/distributed.py) replaces DDP + torch.optim.AdamW: flat gradient buckets exchanged
    once per optimizer step, overlapped with backward, fused AdamW kernel;
  * "mpt" model names (or peft_type flamingo) select CrossAttentionModel exactly as :286-301 dispatch on substrings;
  * `neighbor_layer_wise` exists (the reference reads it but never defines it);
  * the step loop reproduces :462-524: loss / accum, step + scheduler every `grad_accumulation_steps`, NO gradient clipping
    unless grad_clip > 2 (:492), meters all-reduced every print_freq optimizer steps, examples_per_sec = per_device_batch /
    batch_time * n_gpus (:503) -- batch_time from HIP events, resolved when the meters are read (_Meters);
  * the `grad_accumulation_steps` micro-batches of one optimizer step run as ONE forward / backward pass over the concatenated
    samples when that is the same number (train_loop, _fusable; `fuse_grad_accumulation=False` = one pass per micro-batch);
    evaluate_loop shares a forward pass between validation batches the same way;
  * validation CIDEr / BLEU on teacher-forced argmax tokens (:604-606), predictions all-gathered (:608-616);
  * checkpoint dict layout of :402-416 (frozen encoders stripped, `module.` prefix kept) so checkpoints interchange.
wandb / torchmetrics / warmup_scheduler are absent here: logging goes to stdout, BLEU is a local corpus-BLEU,
ROUGE is not computed, the warm-up is restated as linear 0 -> lr over lr_warmup_steps then StepLR (parity unpinned).
"""
import os
import random
import sys
import time
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.distributed as dist
import torch.nn as nn


def _f(default, help_):
    return field(default=default, metadata={"help": help_})


@dataclass
class Arguments:
    """Same field names and defaults as the reference's Arguments (:66-229) + `neighbor_layer_wise`."""
    overwrite_cache: Optional[bool] = _f(False, "Overwrite the cached preprocessed datasets or not.")
    dataset: Optional[str] = _f("wikiweb2m", "The name of the dataset to use.")
    task: Optional[str] = _f("section", "One of three generation tasks in WikiWeb2M")
    context: Optional[str] = _f("section_only", "Range of neighbor context: section_only, section_all, text_only, all")
    max_input_length: Optional[int] = _f(512, "maximum token length of input text")
    max_output_length: Optional[int] = _f(128, "maximum token length of output text")

    wandb_project: Optional[str] = _f("MMGL", "wandb project name (logging is stdout-only in this build)")
    wandb_run: Optional[str] = _f("default", "run name")
    log_dir: Optional[str] = _f("log", "logging dir")
    save_dir: Optional[str] = _f(None, "save dir")
    resume: Optional[str] = _f(None, "path to latest checkpoint (default: none)")

    seed: Optional[int] = _f(None, "seed for initializing training.")
    fp16: Optional[bool] = _f(False, "fp32 compute (the reference's --fp16 calls model.float(), :304-305)")
    bf16: Optional[bool] = _f(False, "bf16 compute (model.bfloat16(), :306-307)")

    test: Optional[bool] = _f(False, "evaluate model on validation set.")

    per_device_train_batch_size: Optional[int] = _f(4, "Batch size per device during training.")
    per_device_val_batch_size: Optional[int] = _f(4, "Batch size per device during validation/test.")
    dataloader_num_workers: Optional[int] = _f(4, "Number of threads to read data.")

    start_epoch: Optional[int] = _f(0, "Starting epoch.")
    epochs: Optional[int] = _f(90, "Total number of epochs.")
    steps_per_epoch: Optional[int] = _f(2000, "Number of training steps per epoch.")
    val_steps_per_epoch: Optional[int] = _f(1000, "Number of validation/test steps per epoch.")
    print_freq: Optional[int] = _f(50, "print frequency")

    learning_rate: Optional[float] = _f(0.001, "initial learning rate.")
    adam_beta1: Optional[float] = _f(0.9, "beta1 for Adam.")
    adam_beta2: Optional[float] = _f(0.95, "beta2 for AdamDecay.")
    weight_decay: Optional[float] = _f(0.01, "Weight decay parameter.")
    grad_accumulation_steps: Optional[int] = _f(4, "number of gradient accumulation steps.")
    grad_clip: Optional[float] = _f(1.0, "gradient clipping amount.")
    lr_warmup_steps: Optional[int] = _f(2000, "Number of steps to warm up lr.")
    lr_schedule_step_size: Optional[int] = _f(5, "Number of steps before decaying lr.")
    lr_schedule_gamma: Optional[float] = _f(0.1, "Decay parameter for learning rate scheduler.")

    model_name_or_path: str = _f(None, "Path to pretrained model or model identifier from huggingface.co/models")
    decoder_only: Optional[bool] = _f(False, "whether LM models are decoder-only: opt or mpt")
    cross_attention: Optional[bool] = _f(False, "whether LM models use cross-attention: mpt")
    text_model: str = _f("roberta-base", "text model to encode neighbor texts")
    visual_model: str = _f("openai/clip-vit-base-patch16", "visual model to encode neighbor images")
    n_text_tokens: int = _f(4, "number of tokens for text embeddings")
    n_visual_tokens: int = _f(4, "number of tokens for visual embeddings")
    freeze_lm: Optional[bool] = _f(False, "whether to freeze LM parameters")
    neighbor_mode: str = _f("raw", "how to encode neighbor information: raw, embedding")
    max_text_neighbors: int = _f(11, "maxinum number of text neighbors")
    max_image_neighbors: int = _f(5, "maximum number of image neighbors")
    position_type: str = _f("none", "position id type for text/image neighbors")

    num_neighbor_layers: int = _f(4, "number of cross-attention layers to encode neighbor information")
    neighbor_layer_wise: Optional[int] = _f(None, "insert a cross-attention layer after every this-many LM layers "
                                                  "(default: num_hidden_layers // num_neighbor_layers)")
    peft_type: str = _f("none", "peft type: none, prefix, prompt, lora, flamingo")
    lora_r: int = _f(64, "lora row rank")
    lora_alpha: float = _f(1, "lora scaling factor")
    lora_dropout: float = _f(0.0, "lora dropout rate")

    # not in the reference: how its batch protocol is executed (train_loop)
    fuse_grad_accumulation: Optional[bool] = _f(True, "run the grad_accumulation_steps micro-batches of one optimizer step as ONE "
                                                "forward/backward pass over the concatenated samples (same samples, order, "
                                                "optimizer steps and loss); False = one pass per micro-batch, literally")
    fuse_eval_batches: Optional[bool] = _f(True, "evaluate_loop: several validation batches share one forward pass (bounded by "
                                           "fused_pass_tokens); meters, gathers and caption order stay per batch")
    fused_pass_tokens: int = _f(49152, "upper bound on samples x sequence length of one fused pass (memory); a group that "
                                "exceeds it is cut into the fewest equal passes")


# =============================================================================================== schedule / metrics
class WarmupStepLR:
    """lr(step): linear 0 -> base over `warmup` optimizer steps (GradualWarmupScheduler, multiplier 1.0), then StepLR
    (x gamma every `step_size` steps, counted from the hand-over) -- reference :332-333.  Parity with the third-party
    warmup_scheduler package is UNPINNED (absent here)."""

    def __init__(self, base_lr, warmup, step_size, gamma):
        self.base_lr, self.warmup, self.step_size, self.gamma = base_lr, max(int(warmup), 0), max(int(step_size), 1), gamma
        self.last_step = 0      # number of scheduler.step() calls so far: optimizer step n runs at lr_at(n - 1), as in the reference

    def lr_at(self, step):
        if self.warmup and step <= self.warmup:
            return self.base_lr * step / self.warmup
        return self.base_lr * self.gamma ** ((step - self.warmup) // self.step_size)

    def step(self):
        self.last_step += 1
        return self.lr_at(self.last_step)

    def get_last_lr(self):
        return [self.lr_at(self.last_step)]

    def state_dict(self):
        return dict(last_step=self.last_step, base_lr=self.base_lr, warmup=self.warmup, step_size=self.step_size, gamma=self.gamma)

    def load_state_dict(self, sd):
        if "last_step" in sd:
            self.last_step = sd["last_step"]
        elif "last_epoch" in sd:              # torch / warmup_scheduler checkpoints of the reference count calls as last_epoch
            self.last_step = sd["last_epoch"]
        else:
            raise ValueError(f"scheduler checkpoint has neither last_step nor last_epoch (keys: {sorted(sd)})")


def corpus_bleu(preds, refs, n_gram=4):
    """Corpus BLEU-n with brevity penalty, uniform weights, whitespace tokens (the quantity torchmetrics.BLEUScore
    computes; torchmetrics is absent here).  refs: list of lists of reference strings."""
    import math
    from collections import Counter
    num = [0] * n_gram
    den = [0] * n_gram
    pred_len = ref_len = 0
    for p, rs in zip(preds, refs):
        pt = p.split()
        rts = [r.split() for r in rs]
        pred_len += len(pt)
        ref_len += min((abs(len(r) - len(pt)), len(r)) for r in rts)[1] if rts else 0
        for k in range(1, n_gram + 1):
            pc = Counter(tuple(pt[i:i + k]) for i in range(len(pt) - k + 1))
            mx = Counter()
            for r in rts:
                rc = Counter(tuple(r[i:i + k]) for i in range(len(r) - k + 1))
                for g, c in rc.items():
                    mx[g] = max(mx[g], c)
            num[k - 1] += sum(min(c, mx[g]) for g, c in pc.items())
            den[k - 1] += max(len(pt) - k + 1, 0)
    if min(num) == 0 or pred_len == 0:
        return 0.0
    logp = sum(math.log(nm / dn) for nm, dn in zip(num, den)) / n_gram
    bp = 1.0 if pred_len > ref_len else math.exp(1 - ref_len / pred_len)
    return bp * math.exp(logp)


# =============================================================================================== model / data factories
OFFLINE_LM_DIMS = {  # architectures that can be built without a checkpoint (`--dataset synthetic`, benchmarks)
    "opt-125m": dict(hidden_size=768, num_attention_heads=12, ffn_dim=3072, num_hidden_layers=12, word_embed_proj_dim=768),
    "opt-350m": dict(hidden_size=1024, num_attention_heads=16, ffn_dim=4096, num_hidden_layers=24, word_embed_proj_dim=512,
                     do_layer_norm_before=False),
    "opt-1.3b": dict(hidden_size=2048, num_attention_heads=32, ffn_dim=8192, num_hidden_layers=24, word_embed_proj_dim=2048),
    "opt-tiny": dict(hidden_size=64, num_attention_heads=4, ffn_dim=128, num_hidden_layers=4, word_embed_proj_dim=64,
                     max_position_embeddings=256),
    "t5-small": dict(d_model=512, d_kv=64, d_ff=2048, num_layers=6, num_heads=8),
    "t5-tiny": dict(d_model=64, d_kv=16, d_ff=128, num_layers=2, num_heads=4),
}


def offline_configs(args, vocab_size):
    """Random-init HF configs for the named architecture (no network / checkpoints in this environment)."""
    from transformers import CLIPVisionConfig, OPTConfig, RobertaConfig, T5Config
    name = args.model_name_or_path.replace("mpt", "opt").split("/")[-1]
    dims = dict(OFFLINE_LM_DIMS[name])
    if "t5" in name:
        lm = T5Config(vocab_size=vocab_size, decoder_start_token_id=1, pad_token_id=1, eos_token_id=2, **dims)
    else:
        dims.setdefault("max_position_embeddings", 2048)
        lm = OPTConfig(vocab_size=vocab_size, pad_token_id=1, bos_token_id=2, eos_token_id=2, **dims)
    tiny = "tiny" in name
    txt = RobertaConfig(vocab_size=vocab_size, hidden_size=32 if tiny else 768, num_hidden_layers=2 if tiny else 12,
                        num_attention_heads=2 if tiny else 12, intermediate_size=64 if tiny else 3072,
                        max_position_embeddings=args.max_input_length + 2, pad_token_id=1, type_vocab_size=1)
    vis = CLIPVisionConfig(hidden_size=32 if tiny else 768, intermediate_size=64 if tiny else 3072, num_hidden_layers=2 if tiny else 12,
                           num_attention_heads=2 if tiny else 12, image_size=getattr(args, "image_size", 224), patch_size=16)
    return lm, txt, vis


def build_model(args, tokenizer, offline=False):
    """Model dispatch on substrings of model_name_or_path (reference :286-301) -- "t5"/"opt" -> SelfAttentionModel,
    "mpt" -> CrossAttentionModel on the corresponding OPT; an "opt" name with peft_type flamingo also selects the
    cross-attention model (README pairing)."""
    from ..model import CrossAttentionModel, SelfAttentionModel
    name = args.model_name_or_path
    cfgs = {}
    if offline:
        lm, txt, vis = offline_configs(args, len(tokenizer))
        cfgs = dict(lm_config=lm, text_config=txt, visual_config=vis)
    if "t5" in name:
        args.decoder_only = False
        return SelfAttentionModel(args, tokenizer, **cfgs)
    if "mpt" in name or "llama" in name.lower() or ("opt" in name and args.peft_type == "flamingo"):
        args.decoder_only = True
        args.model_name_or_path = name.replace("mpt", "opt")
        return CrossAttentionModel(args, tokenizer, **cfgs)
    if "opt" in name:
        args.decoder_only = True
        return SelfAttentionModel(args, tokenizer, **cfgs)
    raise ValueError(f"unsupported model_name_or_path {name!r}: expected a t5 / opt / mpt name (reference :286-301)")


def build_datasets(args, tokenizer):
    from ..wikiweb2m import WikiWeb2M, load_wikiweb2m
    if args.dataset == "synthetic":
        from ..wikiweb2m.synthetic import synthetic_id_list, synthetic_pages
        dfs = [synthetic_pages(24, seed=s) for s in (11, 12, 13)]
        ids = {k: synthetic_id_list(df) for k, df in zip(("train", "val", "test"), dfs)}
        vis = None
    else:
        train_df, val_df, test_df, ids = load_wikiweb2m(args.task)
        dfs, vis = [train_df, val_df, test_df], args.visual_model
    return [WikiWeb2M(args, df, ids[k], tokenizer, vis) for df, k in zip(dfs, ("train", "val", "test"))]


# =============================================================================================== loops
def _device_of(model):
    return next(model.parameters()).device


def _sync(device):
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def _summary_slices(args, logits, labels):
    """Decoder-only: score only the reference summary, logits[..., L_in:-1] vs labels[..., L_in+1:] (reference :473-478)."""
    lg = logits[..., args.max_input_length:-1, :]
    lb = labels[..., (args.max_input_length + 1):]
    if lg.shape[1] - lb.shape[1] > 0:
        lg = lg[..., :-(lg.shape[1] - lb.shape[1]), :]
    return lg, lb


def _summary_cross_entropy(lg, lb, pad_id):
    """The running meter's loss over the summary positions (reference :473-480).  On the GPU: the HIP cross-entropy kernel straight on
    the logits slice (one pass over 0.8 GB at B = 64 instead of an fp32 copy, a log-softmax and a gather: 0.9 ms per step)."""
    lg2, lb1 = lg.reshape(-1, lg.size(-1)), lb.reshape(-1)
    if lg2.is_cuda:
        from .. import ops
        with torch.no_grad():
            return ops.cross_entropy(lg2.contiguous(), lb1.contiguous(), pad_id)
    return nn.functional.cross_entropy(lg2.float(), lb1, ignore_index=pad_id)


def _takes_logits_slice(model):
    from ..model.modelling_cross_attention import CrossAttentionModel
    return isinstance(model, CrossAttentionModel)


def _host_meta(model, batch):
    """Host-side facts about a collated batch (which neighbor slots are real, how long each neighbor text is) for models that
    take them (`CrossAttentionModel.forward(host_meta=)`): the forward pass then never synchronises with the device."""
    from ..model.modelling_cross_attention import CrossAttentionModel, host_metadata
    if isinstance(model, CrossAttentionModel) and all(not v.is_cuda for v in batch.values()):
        return {"host_meta": host_metadata(batch, use_images=model.context in ("section_all", "all"))}
    return {}


def _is_opt_self_attention(model):
    """SelfAttentionModel over this build's OPT fork: its forward takes logits_slice too (a frozen head -- prefix / prompt tuning --
    then never materialises the full logits in a training step)."""
    from ..model.modelling_self_attention import SelfAttentionModel
    return isinstance(model, SelfAttentionModel) and hasattr(model.lm, "model") and hasattr(model.lm.model, "decoder")


def _fusable(model, args, micro_batches):
    """May the micro-batches of one optimizer step run as ONE forward / backward pass with the same result?  The reference's step is
    sum_k mean_k(CE) / accum (:483-485); one pass over the concatenated samples computes a mean over ALL of them, which is the same
    number exactly when every micro-batch scores the same number of positions: decoder-only labels without -100 (labels = input_ids,
    data.py:331-333 -- every position counts) and equal micro-batch sizes (drop_last).  Encoder-decoder labels carry -100 at the pads
    (unequal counts) and LayerDrop flips one coin per forward call (:581-584): those run the literal loop."""
    if not getattr(args, "fuse_grad_accumulation", True) or not getattr(args, "decoder_only", False) or len(micro_batches) < 2:
        return False
    if any(getattr(m, "layerdrop", 0) and m.training for m in model.modules()):
        return False
    shapes = {k: tuple(v.shape) for k, v in micro_batches[0].items()}
    for mb in micro_batches:
        if {k: tuple(v.shape) for k, v in mb.items()} != shapes or bool((mb["labels"] == -100).any()):
            return False
    return True


def _pass_sizes(n_micro, samples, tokens_per_sample, budget_tokens):
    """Cut `n_micro` micro-batches into the fewest passes of at most `budget_tokens` tokens each, as even as possible (16 at a budget of
    9 -> 8 + 8, not 9 + 7: the GEMM grids of both passes then fill the same number of tile rounds)."""
    per = max(1, int(budget_tokens) // max(1, samples * tokens_per_sample)) if budget_tokens else n_micro
    n_pass = -(-n_micro // per)
    base, extra = divmod(n_micro, n_pass)
    return [base + (1 if k < extra else 0) for k in range(n_pass)]


_META_KEYS = ("attention_mask", "neighbor_pos_ids", "neighbor_attention_mask", "neighbor_images_pos_ids")


class _GroupFeeder:
    """The loader's micro-batches, grouped by optimizer step (`grad_accumulation_steps` consecutive ones; the epoch's last group may be
    short, :485), each group already cut into passes and on the device.  `prefetch()` stages the NEXT group while the current one
    computes: its host-to-device copies run on a side stream (the SDMA engines) under the backward pass instead of in front of the
    next forward -- 3 MB of pixels per sample, ~8 ms per 64 samples on the critical path otherwise."""

    def __init__(self, loader, model, args, device, accum):
        self.it, self.model, self.args, self.device, self.accum = iter(loader), model, args, device, accum
        self.i = 0                      # index of the next micro-batch
        self.done = False
        self.staged = None
        self.stream = torch.cuda.Stream(device) if device.type == "cuda" else None

    def _fetch(self):
        if self.done:
            return None
        mbs, first = [], self.i
        while True:
            try:
                mbs.append(next(self.it))
            except StopIteration:
                self.done = True
                break
            self.i += 1
            if self.i % self.accum == 0:
                break
            if self.i == self.args.steps_per_epoch:
                break
        if self.i >= self.args.steps_per_epoch:
            self.done = True
        if not mbs:
            return None
        boundary = (self.i % self.accum == 0) or (self.i == self.args.steps_per_epoch)
        n = len(mbs)
        if _fusable(self.model, self.args, mbs):
            B, T = mbs[0]["input_ids"].shape[:2]
            sizes = _pass_sizes(n, B, T, getattr(self.args, "fused_pass_tokens", 49152))
        else:
            sizes = [1] * n
        passes, k = [], 0
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else None
        for sz in sizes:
            part = mbs[k:k + sz]
            k += sz
            # host-side facts of the concatenated pass (small integer tensors only), read before the copy
            small = {key: (torch.cat([mb[key] for mb in part]) if sz > 1 else part[0][key]) for key in _META_KEYS if key in part[0]}
            extra = _host_meta(self.model, small) if "attention_mask" in small else {}
            if ctx is not None:
                with ctx:
                    dev = [{key: v.to(self.device, non_blocking=True) for key, v in mb.items()} for mb in part]
                    batch = dev[0] if sz == 1 else {key: torch.cat([d[key] for d in dev]) for key in dev[0]}
            else:
                dev = [{key: v.to(self.device) for key, v in mb.items()} for mb in part]
                batch = dev[0] if sz == 1 else {key: torch.cat([d[key] for d in dev]) for key in dev[0]}
            passes.append(dict(batch=batch, extra=extra, n_micro=sz, micro_size=part[0]["input_ids"].size(0)))
        ready = None
        if self.stream is not None:
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return dict(passes=passes, n_micro=n, last_index=self.i - 1, first_index=first, boundary=boundary, ready=ready)

    def prefetch(self):
        if self.staged is None and not self.done:
            self.staged = self._fetch()

    def __iter__(self):
        return self

    def __next__(self):
        g, self.staged = (self.staged, None) if self.staged is not None else (self._fetch(), None)
        if g is None:
            raise StopIteration
        if g["ready"] is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(g["ready"])
            for p in g["passes"]:
                for t in p["batch"].values():
                    t.record_stream(cur)
        return g


class _Meters:
    """The reference's four running meters (:447-450) without its per-micro-batch host synchronisation.  The reference calls
    `summary_loss.item()` on every micro-batch (:480) and reads unsynchronised wall clocks; reading them right would cost a device
    sync per step, and with syncs the host cannot launch the next pass while the current one computes -- at the reference's batch sizes
    the launch stream is then the bottleneck.  Here the loss stays a device scalar and the times are HIP events; they are RESOLVED --
    one synchronisation -- when somebody looks: every print_freq optimizer steps (:498-502) and at the end of the epoch.  The values
    that reach the meters, and their order, are those of the per-step version."""

    def __init__(self, device, utils):
        self.cuda = device.type == "cuda"
        self.batch_time = utils.AverageMeter("Time", ":6.3f")
        self.data_time = utils.AverageMeter("Data", ":6.3f")
        self.forward_time = utils.AverageMeter("Forward", ":6.3f")
        self.losses = utils.AverageMeter("Loss", ":.4e")
        self.pending = []

    def all(self):
        return (self.losses, self.batch_time, self.data_time, self.forward_time)

    def mark(self):
        if not self.cuda:
            return time.time()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def span(self, meter, t0, t1, n):
        self.pending.append((meter, t0, t1, n))

    def loss(self, values, n):
        """`values`: list of device (or host) scalars, one per micro-batch of `n` samples"""
        self.pending.append((self.losses, values, None, n))

    def resolve(self):
        if self.cuda and self.pending:
            torch.cuda.synchronize()
        for meter, a, b, n in self.pending:
            if meter is self.losses:
                vals = torch.stack([v.float().reshape(()) for v in a]).tolist() if torch.is_tensor(a[0]) else a
                for v in vals:
                    meter.update(v, n)
            else:
                dt = a.elapsed_time(b) * 1e-3 if self.cuda else b - a
                meter.update(dt / n, n)
        self.pending = []


def train_loop(train_loader, model, tokenizer, engine, epoch, scheduler, args, run=None):
    """One epoch (reference :430-524).  `engine` = DataParallelEngine (replaces DDP + optimizer).

    The reference's protocol is `per_device_train_batch_size` x `grad_accumulation_steps` (2 x 16 in script/train_generation.sh:26-29,
    4 x 4 by default): `accum` forward / backward passes of a few samples each per optimizer step.  Equal-size micro-batches with a
    mean loss / accum ARE one batch of B * accum samples, and an MI355X holds it (288 GB): by default the group runs as ONE pass over
    the concatenated samples -- same samples, same order, same optimizer / scheduler step count, the per-micro-batch summary-loss
    meter kept per chunk (it ignores pads, so a mean of chunk means is not the mean over the pass) and the short last group still
    scaled by 1 / accum (:485) -- cut only where `fused_pass_tokens` says memory demands.  `fuse_grad_accumulation=False` is the
    literal loop (one pass per micro-batch); `_fusable` says when the literal loop runs regardless.  The loop itself never waits for
    the device (see _Meters): the next pass is launched while the current one computes."""
    from . import utils
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    device = _device_of(model)
    mt = _Meters(device, utils)
    progress = utils.ProgressMeter(args.steps_per_epoch, [mt.batch_time, mt.losses], prefix=f"Epoch: [{epoch}]")
    pad_id = tokenizer.pad_token_id if tokenizer is not None else 1
    accum = max(1, args.grad_accumulation_steps)
    model.train()
    engine.zero_grad()
    history = []
    sliced = args.decoder_only and (_takes_logits_slice(model) or _is_opt_self_attention(model))
    feeder = _GroupFeeder(train_loader, model, args, device, accum)
    _sync(device)
    end, end_host = mt.mark(), time.time()
    for group in feeder:
        g = group["n_micro"]
        mt.data_time.update((time.time() - end_host) / g, g)      # host time spent waiting for the loader
        for pi, ps in enumerate(group["passes"]):
            batch, extra, k = ps["batch"], dict(ps["extra"]), ps["n_micro"]
            if sliced:                                 # the running summary loss below reads positions L_in .. T-2 only: the training
                # step then never builds the [B, T, V] logits (explicit stop: SelfAttentionModel appends neighbor tokens after position T-1)
                extra["logits_slice"] = slice(args.max_input_length, batch["input_ids"].shape[1] - 1)
            engine.sync = group["boundary"] and pi == len(group["passes"]) - 1   # gradients cross xGMI once per optimizer step
            f0 = mt.mark()
            outputs = model(**batch, **extra)
            mt.span(mt.forward_time, f0, mt.mark(), k)
            loss = outputs.loss
            mb = ps["micro_size"]
            if args.decoder_only:
                if sliced:
                    lg, lb = outputs.logits.detach(), batch["labels"][..., (args.max_input_length + 1):]
                else:
                    lg, lb = _summary_slices(args, outputs.logits.detach(), batch["labels"])
                # one meter entry per MICRO-batch, as the literal loop records them (:473-480)
                mt.loss([_summary_cross_entropy(lg[c * mb:(c + 1) * mb], lb[c * mb:(c + 1) * mb], pad_id) for c in range(k)], mb)
            else:
                mt.loss([loss.detach()], mb)
            # the pass's loss is the mean over its k micro-batches' positions = (1 / k) * sum_k mean_k: times k / accum (:483)
            (loss * (k / accum)).backward()
            engine.finish_backward()
        lr = None
        if group["boundary"]:
            # reference order (:486-494): optimizer.step() at the current lr, THEN scheduler.step() -- the warm-up starts at 0
            lr = scheduler.get_last_lr()[0] if scheduler is not None else None
            engine.step(lr)
            if scheduler is not None:
                scheduler.step()
            # the reference clips only if grad_clip > 2, AFTER the step, i.e. to no effect (:490-493): nothing to do
            engine.zero_grad()
        now = mt.mark()
        mt.span(mt.batch_time, end, now, g)
        end, end_host = now, time.time()
        feeder.prefetch()                              # the next group's H2D copies run under this group's passes
        if group["boundary"]:
            actual_step = (epoch * args.steps_per_epoch + group["last_index"] + 1) // accum
            if actual_step == 1 or actual_step % args.print_freq == 0:
                mt.resolve()
                for m in mt.all():
                    m.all_reduce()
                ex_per_sec = (args.per_device_train_batch_size / max(mt.batch_time.avg, 1e-9)) * world_size
                history.append(dict(step=actual_step, loss=mt.losses.avg, examples_per_sec=ex_per_sec, lr=lr,
                                    passes=[p["n_micro"] for p in group["passes"]]))
                if rank == 0:
                    progress.display(group["last_index"] + 1)
                    print(f"  step {actual_step}: loss {mt.losses.avg:.4f}  examples/sec {ex_per_sec:.2f}  "
                          f"data {mt.data_time.avg:.3f}s  fwd {mt.forward_time.avg:.3f}s  lr {lr}")
                for m in mt.all():
                    m.reset()
                end, end_host = mt.mark(), time.time()
    mt.resolve()
    _sync(device)
    return history


def _eval_groups(loader, args, limit):
    """Consecutive validation batches, `k` at a time (k * B * T <= fused_pass_tokens; samples are independent in eval mode, so one
    forward over k batches is k forwards) -- at per_device_val_batch_size 2 (script/train_generation.sh) a forward per batch leaves
    the GPU waiting for the launch stream."""
    group, n = [], 0
    for batch in loader:
        n += 1
        if group and any(tuple(v.shape) != tuple(group[0][key].shape) for key, v in batch.items()):
            yield group
            group = []
        group.append(batch)
        B, T = batch["input_ids"].shape[:2]
        k = 1                                           # encoder-decoder: the wrapper's own loss is the meter (:577-579) -> one batch per pass
        if getattr(args, "fuse_eval_batches", True) and args.decoder_only:
            k = max(1, int(getattr(args, "fused_pass_tokens", 49152)) // max(1, B * T))
        if len(group) >= k:
            yield group
            group = []
        if n == limit:
            break
    if group:
        yield group


def evaluate_loop(val_loader, model, tokenizer, epoch, args, run=None, prefix="val"):
    """Teacher-forced evaluation (reference :527-703): argmax tokens on the summary span, all-gathered, decoded,
    truncated at the first '.', scored with BLEU-1..4 and CIDEr.  Returns BLEU-4 (the model-selection metric, :703).
    Several validation batches share one forward pass (_eval_groups); meters, gathers and caption order are per batch, as in the
    reference.  `evaluate_loop.last` also carries `samples_per_sec` (this rank's samples / wall time of the loop)."""
    from . import utils
    from ..wikiweb2m.cider import Cider
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    device = _device_of(model)
    batch_time = utils.AverageMeter("Time", ":6.3f", utils.Summary.AVERAGE)
    losses = utils.AverageMeter("Loss", ":.4e", utils.Summary.AVERAGE)
    progress = utils.ProgressMeter(args.val_steps_per_epoch, [batch_time, losses], prefix=f"{prefix}: ")
    pad_id = tokenizer.pad_token_id
    model.eval()
    gen_caps, gt_caps = [], []
    sliced = args.decoder_only and (_takes_logits_slice(model) or _is_opt_self_attention(model))
    n_samples, seen = 0, 0
    _sync(device)
    t_loop = time.time()
    with torch.no_grad():
        end = time.time()
        for group in _eval_groups(val_loader, args, args.val_steps_per_epoch):
            k, mb = len(group), group[0]["input_ids"].size(0)
            small = {key: (torch.cat([b[key] for b in group]) if k > 1 else group[0][key]) for key in _META_KEYS if key in group[0]}
            extra = _host_meta(model, small) if "attention_mask" in small else {}
            dev = [{key: v.to(device, non_blocking=True) for key, v in b.items()} for b in group]
            batch = dev[0] if k == 1 else {key: torch.cat([d[key] for d in dev]) for key in dev[0]}
            if sliced:                                 # only the summary positions' logits are read below (:584-591)
                extra["logits_slice"] = slice(args.max_input_length, batch["input_ids"].shape[1] - 1)
            outputs = model(**batch, **extra)
            logits = outputs.logits
            if args.decoder_only:
                if sliced:
                    labels = batch["labels"][..., (args.max_input_length + 1):]
                else:
                    logits, labels = _summary_slices(args, logits, batch["labels"])
                chunk = [_summary_cross_entropy(logits[c * mb:(c + 1) * mb], labels[c * mb:(c + 1) * mb], pad_id) for c in range(k)]
                chunk = torch.stack([c.float().reshape(()) for c in chunk]).tolist()
            else:
                labels = batch["labels"]
                chunk = [outputs.loss.item()]
            for v in chunk:
                losses.update(v, mb)
            if prefix == "test" and hasattr(model, "generate"):
                generated_ids = model.generate(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], max_new_tokens=32)
            else:
                generated_ids = torch.argmax(logits, dim=-1)            # the reference's wrappers have no generate() (:600)
            labels = labels.contiguous()
            generated_ids = generated_ids.contiguous()
            if world_size > 1:
                gl = [torch.zeros_like(generated_ids) for _ in range(world_size)]
                tl = [torch.zeros_like(labels) for _ in range(world_size)]
                dist.all_gather(gl, generated_ids)
                dist.all_gather(tl, labels)
                # the reference gathers per batch (:608-616): batch c of every rank, rank-major, then batch c + 1
                generated_ids = torch.cat([g[c * mb:(c + 1) * mb] for c in range(k) for g in gl])
                labels = torch.cat([t[c * mb:(c + 1) * mb] for c in range(k) for t in tl])
            if not args.decoder_only:
                labels = labels.masked_fill(labels == -100, pad_id)
            preds = tokenizer.batch_decode(generated_ids, skip_special_tokens=True)
            gts = tokenizer.batch_decode(labels, skip_special_tokens=True)
            for p, g in zip(preds, gts):
                stop = p.find(".")
                gen_caps.append(p[:stop] if stop > 5 else p)
                gt_caps.append([g])
            n_samples += k * mb
            batch_time.update((time.time() - end) / k, k)
            end = time.time()
            if any((seen + c) % args.print_freq == 0 for c in range(k)) and rank == 0:
                progress.display(seen + k)
            seen += k
    _sync(device)
    t_loop = time.time() - t_loop
    bleu = [corpus_bleu(gen_caps, gt_caps, n) for n in (1, 2, 3, 4)]
    cands = {idx: [p] for idx, p in enumerate(gen_caps)}
    refs = {idx: g for idx, g in enumerate(gt_caps)}
    cider_score, _ = Cider().compute_score(refs, cands) if gen_caps else (0.0, None)
    meters = {}
    for name, val in [("loss", losses.avg), ("bleu1", bleu[0]), ("bleu2", bleu[1]), ("bleu3", bleu[2]), ("bleu4", bleu[3]), ("cider", cider_score)]:
        m = utils.AverageMeter(name, ":6.4f", utils.Summary.AVERAGE)
        m.update(val, 1)
        m.all_reduce()
        meters[name] = m.avg
    if rank == 0:
        print(f"[{prefix}] epoch {epoch}: " + "  ".join(f"{k} {v:.4f}" for k, v in meters.items()) + f"  ({len(gen_caps)} captions)")
    evaluate_loop.last = meters
    evaluate_loop.samples_per_sec = n_samples / max(t_loop, 1e-9)
    return meters["bleu4"]


# =============================================================================================== main / worker
best_acc1 = 0


def save_checkpoint(path, model, engine, scheduler, epoch, acc1):
    """Reference layout (:402-416): frozen encoders stripped, keys sorted, `module.` prefix as under DDP."""
    sd = {"module." + k: v for k, v in model.state_dict().items() if ".text_model" not in "." + k and ".visual_model" not in "." + k}
    state = {"epoch": epoch, "best_acc1": acc1, "state_dict": OrderedDict(sorted(sd.items())), "optimizer": engine.state_dict()}
    if scheduler is not None:
        state["scheduler"] = scheduler.state_dict()
    torch.save(state, path)


def load_checkpoint(path, model, engine, scheduler, map_location, trust=None):
    """Checkpoints written by save_checkpoint hold tensors and plain containers only and load through torch's weights-only
    unpickler.  The REFERENCE's checkpoints (:402-416) do not: `GradualWarmupScheduler.state_dict()` keeps its `after_scheduler`
    -- a pickled StepLR object, which carries the optimizer -- and the weights-only unpickler rejects that by design.  They load
    with `trust=True` (or MMGL_TRUST_CHECKPOINT=1): the file is then unpickled the way the reference itself loads it (:340,
    arbitrary code in the file WILL run), the StepLR object is dropped and its call count (`last_epoch`) taken over; the
    optimizer state is matched by parameter name (DataParallelEngine.load_state_dict)."""
    import pickle
    if trust is None:
        trust = os.environ.get("MMGL_TRUST_CHECKPOINT", "0") == "1"
    try:
        with torch.serialization.safe_globals([OrderedDict]):
            ck = torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as e:
        if not trust:
            raise RuntimeError(
                f"{path}: not a tensors-and-plain-containers checkpoint ({str(e).splitlines()[0]}).  A checkpoint written by the "
                "reference embeds pickled scheduler / optimizer objects; if you trust the file, load it with "
                "MMGL_TRUST_CHECKPOINT=1 (load_checkpoint(..., trust=True))") from e
        ck = torch.load(path, map_location=map_location, weights_only=False)
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in ck["state_dict"].items()}
    model.load_state_dict(sd, strict=False)
    engine.sync_master_from_params()
    if "optimizer" in ck:
        engine.load_state_dict(ck["optimizer"])
    if scheduler is not None and "scheduler" in ck:
        sch = ck["scheduler"]
        if not isinstance(sch, dict):
            sch = sch.state_dict()
        scheduler.load_state_dict({k: v for k, v in sch.items() if k != "after_scheduler"})
    return ck


def main_worker(gpu, world_size, args, log_dir, run=None, tokenizer=None, offline=None, datasets=None, backend=None):
    """One process per GPU (reference :269-428)."""
    global best_acc1
    from ..distributed import DataParallelEngine
    use_cuda = torch.cuda.is_available() and backend != "gloo"
    backend = backend or ("nccl" if use_cuda else "gloo")
    if not dist.is_initialized():
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29517")
        kw = dict(device_id=torch.device("cuda", gpu)) if backend == "nccl" else {}
        dist.init_process_group(backend=backend, world_size=world_size, rank=int(os.environ.get("RANK", gpu)), **kw)
    offline = (args.dataset == "synthetic") if offline is None else offline
    if tokenizer is None:
        if offline:
            from ..wikiweb2m.synthetic import synthetic_tokenizer
            tokenizer = synthetic_tokenizer()
        else:
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(args.model_name_or_path.replace("mpt", "opt"), use_fast=False)
    model = build_model(args, tokenizer, offline=offline)
    if args.fp16:
        model = model.float()
    elif args.bf16:
        model = model.bfloat16()
    device = torch.device("cuda", gpu) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(gpu)
    model.to(device)
    from . import utils
    if dist.get_rank() == 0:
        _, ntrain, nfrozen = utils.get_params_count(model)
        print(f"total_params {ntrain + nfrozen}  trainable_params {ntrain}  non_trainable_params {nfrozen}")
    engine = DataParallelEngine(model, lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2), eps=1e-8,
                                weight_decay=args.weight_decay, optimizer="adafactor" if "t5" in args.model_name_or_path else "adamw",
                                master_weights=False if "t5" in args.model_name_or_path else None)
    scheduler = None
    if "t5" not in args.model_name_or_path:
        scheduler = WarmupStepLR(args.learning_rate, args.lr_warmup_steps,
                                 (args.lr_schedule_step_size * args.steps_per_epoch) // max(1, args.grad_accumulation_steps),
                                 args.lr_schedule_gamma)
    if args.resume:
        path = os.path.join(args.log_dir, args.resume, "ckpt.pth.tar")
        if os.path.isfile(path):
            ck = load_checkpoint(path, model, engine, scheduler, device)
            args.start_epoch, best_acc1 = ck["epoch"], ck["best_acc1"]
            print(f"=> loaded checkpoint '{path}' (epoch {ck['epoch']}, best_acc {ck['best_acc1']})")
        else:
            print(f"=> no checkpoint found at '{path}'")

    train_ds, val_ds, test_ds = datasets if datasets is not None else build_datasets(args, tokenizer)
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    samplers = [DistributedSampler(train_ds, drop_last=True), DistributedSampler(val_ds, shuffle=False, drop_last=True),
                DistributedSampler(test_ds, shuffle=False, drop_last=True)]
    nw = args.dataloader_num_workers
    mk = lambda ds, bs, smp: DataLoader(ds, batch_size=bs, shuffle=False, num_workers=nw, prefetch_factor=(10 if nw else None),
                                        pin_memory=use_cuda, sampler=smp, drop_last=True)
    train_loader = mk(train_ds, args.per_device_train_batch_size, samplers[0])
    val_loader = mk(val_ds, args.per_device_val_batch_size, samplers[1])
    test_loader = mk(test_ds, args.per_device_val_batch_size, samplers[2])
    if args.test:
        return evaluate_loop(test_loader, model, tokenizer, args.start_epoch, args, run, "test")
    results = dict(history=[], val=[])
    for epoch in range(args.start_epoch, args.epochs):
        t0 = time.time()
        if epoch == 0:
            evaluate_loop(val_loader, model, tokenizer, epoch - 1, args, run)
        samplers[0].set_epoch(epoch)
        results["history"] += train_loop(train_loader, model, tokenizer, engine, epoch, scheduler, args, run)
        acc1 = evaluate_loop(val_loader, model, tokenizer, epoch, args, run)
        results["val"].append(dict(evaluate_loop.last))
        is_best = acc1 > best_acc1
        best_acc1 = max(acc1, best_acc1)
        if dist.get_rank() == 0 and (is_best or epoch == 0) and args.save_dir:
            print("=> save best val model ...", args.save_dir)
            save_checkpoint(args.save_dir, model, engine, scheduler, epoch, acc1)
        if dist.get_rank() == 0:
            print(f"Epoch {epoch} time: {time.time() - t0:.1f}s")
    results["engine"] = engine
    results["model"] = model
    return results


def _spawn_entry(gpu, world_size, args, log_dir):
    os.environ["RANK"] = str(gpu)
    main_worker(gpu, world_size, args, log_dir)


def main():
    from transformers import HfArgumentParser
    args = HfArgumentParser((Arguments,)).parse_args_into_dataclasses()[0]
    i = 0
    while os.path.exists(os.path.join(args.log_dir, f"{args.wandb_run}_{i}")):
        i += 1
    log_dir = os.path.join(args.log_dir, f"{args.wandb_run}_{i}")
    os.makedirs(log_dir, exist_ok=True)
    args.save_dir = os.path.join(log_dir, "ckpt.pth.tar")
    print(f"Logging to {log_dir}.")
    if args.seed is not None:
        random.seed(args.seed)
        torch.manual_seed(args.seed)
    if "WORLD_SIZE" in os.environ:                      # torchrun: one process per GPU already exists
        main_worker(int(os.environ.get("LOCAL_RANK", 0)), int(os.environ["WORLD_SIZE"]), args, log_dir)
    else:
        n = max(1, torch.cuda.device_count())
        if n == 1:
            os.environ.setdefault("RANK", "0")
            main_worker(0, 1, args, log_dir)
        else:
            import torch.multiprocessing as mp
            mp.spawn(_spawn_entry, nprocs=n, args=(n, args, log_dir))


if __name__ == "__main__":
    main()
