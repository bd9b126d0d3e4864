"""Two optimizer steps of the trainer for every (model family, context, neighbor_mode, peft_type, position_type) combination the
reference's Arguments allow, on synthetic pages: a crash sweep of the wrapper / trainer plumbing.   python tools/probes/mode_sweep.py"""
import itertools
import os
import sys
import tempfile
import traceback

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mmgl_amd.language_modelling.run_generation import Arguments, main_worker  # noqa: E402

os.environ.update(MASTER_ADDR="127.0.0.1", RANK="0")
combos = []
for ctx, peft in itertools.product(["section_only", "section_all", "text_only", "all"], ["none", "flamingo", "lora"]):
    combos.append(("mpt-tiny", ctx, "embedding", peft, "none"))
for ctx, mode, peft in itertools.product(["section_only", "section_all", "text_only", "all"], ["raw", "embedding"], ["none", "lora", "prefix", "prompt"]):
    combos.append(("opt-tiny", ctx, mode, peft, "none"))
for pos in ["laplacian", "gnn"]:
    combos.append(("opt-tiny", "all", "embedding", "none", pos))
for ctx, mode, peft in itertools.product(["section_only", "all"], ["raw", "embedding"], ["none", "lora", "prompt"]):
    combos.append(("t5-tiny", ctx, mode, peft, "none"))
bad = []
for i, (name, ctx, mode, peft, pos) in enumerate(combos):
    os.environ["MASTER_PORT"] = str(29600 + i % 300)
    tmp = tempfile.mkdtemp()
    try:
        args = Arguments(model_name_or_path=name, dataset="synthetic", context=ctx, neighbor_mode=mode, peft_type=peft, position_type=pos,
                         max_input_length=32, max_output_length=12, max_text_neighbors=5, max_image_neighbors=2, n_text_tokens=2,
                         n_visual_tokens=2, per_device_train_batch_size=2, per_device_val_batch_size=2, dataloader_num_workers=0, epochs=1,
                         steps_per_epoch=2, val_steps_per_epoch=1, print_freq=1, grad_accumulation_steps=1, learning_rate=1e-3,
                         lr_warmup_steps=1, log_dir=tmp, seed=0, bf16="t5" not in name, decoder_only="t5" not in name)
        args.image_size = 32
        args.save_dir = os.path.join(tmp, "ckpt.pth.tar")
        res = main_worker(0, 1, args, tmp)
        ok = all(torch.isfinite(torch.tensor(h["loss"])) for h in res["history"])
        print(f"{'ok ' if ok else 'NAN'} {name} {ctx} {mode} {peft} {pos}", flush=True)
        if not ok:
            bad.append((name, ctx, mode, peft, pos, "non-finite loss"))
    except Exception as e:                                    # noqa: BLE001
        msg = traceback.format_exc().strip().splitlines()
        print(f"ERR {name} {ctx} {mode} {peft} {pos}: {type(e).__name__}: {str(e)[:160]} @ {msg[-3].strip()[:120]}", flush=True)
        bad.append((name, ctx, mode, peft, pos, repr(e)[:200]))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
print(f"{len(combos) - len(bad)} / {len(combos)} combinations ran")
