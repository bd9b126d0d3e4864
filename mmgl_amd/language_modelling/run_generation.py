#!/usr/bin/env python
"""Fine-tuning / evaluation driver with the reference's `Arguments` surface and step order
(language_modelling/run_generation.py), re-wired for MI355X:

  * one process per GPU (torchrun or the built-in spawn), torch.distributed backend "nccl" == RCCL over xGMI;
    no hard-coded rendezvous (reference :283 pins tcp://127.0.0.1:1337);
  * `DataParallelEngine` (mmgl_amd/distributed.py) replaces DDP + torch.optim.AdamW: flat gradient buckets exchanged
    once per optimizer step, overlapped with backward, fused AdamW kernel;
  * "mpt" model names (or peft_type flamingo) select CrossAttentionModel exactly as :286-301 dispatch on substrings;
  * `neighbor_layer_wise` exists (the reference reads it but never defines it);
  * the step loop reproduces :462-524: loss / accum, backward every micro-batch, step + scheduler every
    `grad_accumulation_steps`, NO gradient clipping unless grad_clip > 2 (:492), meters all-reduced every print_freq
    optimizer steps, examples_per_sec = per_device_batch / batch_time * n_gpus (:503) -- with a device sync before the
    clock is read;
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
