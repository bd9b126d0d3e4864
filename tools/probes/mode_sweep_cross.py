import os, sys, tempfile, traceback, itertools, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mmgl_amd.language_modelling.run_generation import Arguments, main_worker
os.environ.update(MASTER_ADDR="127.0.0.1", RANK="0")
for i, (ctx, peft) in enumerate(itertools.product(["text_only", "section_all", "all"], ["none", "lora", "flamingo"])):
    os.environ["MASTER_PORT"] = str(29700 + i)
    tmp = tempfile.mkdtemp()
    try:
        args = Arguments(model_name_or_path="mpt-tiny", dataset="synthetic", context=ctx, neighbor_mode="cross_attention", peft_type=peft,
                         max_input_length=32, max_output_length=12, max_text_neighbors=5, max_image_neighbors=2, n_text_tokens=2,
                         n_visual_tokens=2, per_device_train_batch_size=2, per_device_val_batch_size=2, dataloader_num_workers=0, epochs=1,
                         steps_per_epoch=2, val_steps_per_epoch=1, print_freq=1, grad_accumulation_steps=1, learning_rate=1e-3,
                         lr_warmup_steps=1, log_dir=tmp, seed=0, bf16=True)
        args.image_size = 32
        args.save_dir = os.path.join(tmp, "ckpt.pth.tar")
        res = main_worker(0, 1, args, tmp)
        print("ok ", ctx, peft, [round(h["loss"], 3) for h in res["history"]][:2], len(res["engine"].names), flush=True)
    except Exception as e:
        msg = traceback.format_exc().strip().splitlines()
        print("ERR", ctx, peft, type(e).__name__, str(e)[:200], "@", msg[-3].strip()[:100], flush=True)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
