"""-m gpu: the trainer on one MI355X with synthetic pages (config-2-shaped but tiny), the self-attention wrapper against
the reference's golden vectors, LoRA injection, and the fused optimizer through the engine."""
import os

import pytest
import torch
import torch.distributed as dist

from helpers import (Fixture, assert_close, load_exact, mpt_args, tiny_clip_vision_config, tiny_opt_config,
                     tiny_roberta_config)

pytestmark = pytest.mark.gpu


def _sa_args(**kw):
    base = dict(neighbor_mode="embedding", peft_type="none")
    base.update(kw)
    return mpt_args(**base)


@pytest.mark.parametrize("tag", ["none", "laplacian"])
def test_g9_self_attention_model_golden(tag):
    from mmgl_amd.model import SelfAttentionModel
    fx = Fixture(f"g9_selfattn_{tag}.npz")
    w = SelfAttentionModel(_sa_args(position_type=fx.meta["position_type"]), None, lm_config=tiny_opt_config(),
                           text_config=tiny_roberta_config(), visual_config=tiny_clip_vision_config())
    load_exact(w, fx.p)
    w = w.cuda().eval()
    b = {k: v.cuda() for k, v in fx.inp.items()}
    o = w(**b)
    assert o.logits.shape == fx.out["logits"].shape           # [B, T + S, V]: neighbors appended after the sequence
    assert_close(o.logits, fx.out["logits"], 1e-3, "logits")
    assert_close(o.loss, fx.out["loss"], 1e-3, "loss")
    o.loss.backward()
    params = dict(w.named_parameters())
    for k, g in fx.grad.items():
        if params[k].grad is None:
            assert g.abs().max() == 0, k
            continue
        assert_close(params[k].grad, g, 2e-3, f"d {k}", ) if g.abs().max() > 1e-7 else None


def test_lora_injection_trains_only_adapters_and_head():
    from mmgl_amd.model import SelfAttentionModel
    from mmgl_amd.model.modelling_self_attention import LoRALinear
    torch.manual_seed(0)
    w = SelfAttentionModel(_sa_args(peft_type="lora", lora_r=8, lora_alpha=16.0, context="all"), None, lm_config=tiny_opt_config(dropout=0.0),
                           text_config=tiny_roberta_config(), visual_config=tiny_clip_vision_config()).cuda().eval()
    n_lora = sum(isinstance(m, LoRALinear) for m in w.modules())
    assert n_lora == 2 * 4                                    # q_proj + v_proj in each of the 4 layers
    lm_trainable = sorted(n for n, p in w.lm.named_parameters() if p.requires_grad)
    assert all(("lora_" in n) or n.startswith("lm_head") for n in lm_trainable) and any("lora_A" in n for n in lm_trainable)
    fx = Fixture("g9_selfattn_none.npz")
    b = {k: v.cuda() for k, v in fx.inp.items()}
    base = w(**b)
    # lora_B = 0 at init => identical to the un-adapted model; then a non-zero B changes the logits
    with torch.no_grad():
        for m in w.modules():
            if isinstance(m, LoRALinear):
                ref = torch.nn.functional.linear(torch.ones(1, m.base_layer.in_features, device="cuda"), m.base_layer.weight, m.base_layer.bias)
                assert_close(m(torch.ones(1, m.base_layer.in_features, device="cuda")), ref, 1e-5, "B=0")
                m.lora_B.normal_(std=0.05)
    out = w(**b)
    assert (out.logits - base.logits).abs().max() > 1e-4
    out.loss.backward()
    for n, p in w.named_parameters():
        if "lora_" in n:
            assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0, n


def test_prefix_tuning_wrapper_trains_only_the_prefix_table():
    """peft_type="prefix" on the decoder-only OPT: the only trainable LM-side parameter is peft's prefix table
    [20, 2 * n_layers * d]; logits keep the sequence length (the prefix lives inside the attention of every layer, not in the
    sequence); a zero-gradient-free backward reaches the table through all layers; prompt tuning stays an input prefix."""
    from mmgl_amd.model import SelfAttentionModel
    from mmgl_amd.model.modelling_self_attention import NUM_VIRTUAL_TOKENS
    torch.manual_seed(0)
    oc = tiny_opt_config(dropout=0.0)
    w = SelfAttentionModel(_sa_args(peft_type="prefix", context="all"), None, lm_config=oc, text_config=tiny_roberta_config(),
                           visual_config=tiny_clip_vision_config()).cuda().eval()
    assert w.prompt_embeddings is None
    assert tuple(w.prefix_encoder.weight.shape) == (NUM_VIRTUAL_TOKENS, 2 * oc.num_hidden_layers * oc.hidden_size)
    assert not any(p.requires_grad for p in w.lm.parameters())
    fx = Fixture("g9_selfattn_none.npz")
    b = {k: v.cuda() for k, v in fx.inp.items()}
    out = w(**b)
    assert out.logits.shape == fx.out["logits"].shape          # no virtual tokens in the sequence
    out.loss.backward()
    g = w.prefix_encoder.weight.grad
    assert g is not None and torch.isfinite(g).all()
    per_layer = g.view(NUM_VIRTUAL_TOKENS, oc.num_hidden_layers, 2, oc.hidden_size).abs().amax(dim=(0, 3))
    assert (per_layer > 0).all(), "every layer's key and value prefix must receive a gradient"
    # the prefix changes the output (it is not a no-op) ...
    with torch.no_grad():
        w.prefix_encoder.weight.mul_(0.0)
        zeroed = w(**b)
    assert (zeroed.logits - out.logits).abs().max() > 1e-4
    # ... and prompt tuning remains the input-embedding prefix (sequence grows by the virtual tokens)
    wp = SelfAttentionModel(_sa_args(peft_type="prompt", context="all"), None, lm_config=oc, text_config=tiny_roberta_config(),
                            visual_config=tiny_clip_vision_config()).cuda().eval()
    assert wp.prefix_encoder is None and wp(**b).logits.shape[1] == fx.out["logits"].shape[1] + NUM_VIRTUAL_TOKENS


def test_trainer_flamingo_synthetic_one_gpu(tmp_path):
    """run_generation on cuda:0: mpt-tiny, context=all, neighbor_mode=embedding, flamingo; loss goes down, checkpoint
    round-trips, a resumed model reproduces the validation metrics."""
    from mmgl_amd.language_modelling.run_generation import Arguments, evaluate_loop, load_checkpoint, main_worker
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0")
    args = Arguments(model_name_or_path="mpt-tiny", dataset="synthetic", context="all", neighbor_mode="embedding", peft_type="flamingo",
                     max_input_length=32, max_output_length=12, max_text_neighbors=5, max_image_neighbors=2, n_text_tokens=2,
                     n_visual_tokens=2, per_device_train_batch_size=4, per_device_val_batch_size=4, dataloader_num_workers=0, epochs=2,
                     steps_per_epoch=12, val_steps_per_epoch=3, print_freq=2, grad_accumulation_steps=2, learning_rate=3e-3,
                     lr_warmup_steps=2, log_dir=str(tmp_path), seed=0, fp16=True)
    args.image_size = 32
    args.save_dir = str(tmp_path / "ckpt.pth.tar")
    torch.manual_seed(0)
    try:
        res = main_worker(0, 1, args, str(tmp_path))
        hist = res["history"]
        assert len(hist) >= 4 and hist[-1]["loss"] < hist[0]["loss"], [h["loss"] for h in hist]
        model, engine = res["model"], res["engine"]
        assert sorted(engine.names)[0].startswith("lm.model.decoder.neighbor_layers")
        ck = torch.load(args.save_dir, weights_only=False)
        assert {"epoch", "best_acc1", "state_dict", "optimizer", "scheduler"} <= set(ck)
        assert "module.lm.model.decoder.neighbor_layers.0.gating1" in ck["state_dict"]
        assert "module.text_embeddings.weight" in ck["state_dict"]
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_trainer_prompt_tuning_two_steps(tmp_path):
    """peft_type="prompt" through the trainer: the frozen head runs the fused lm_head + cross-entropy, the running summary meter
    reads a logits slice shifted by the virtual tokens."""
    from mmgl_amd.language_modelling.run_generation import Arguments, main_worker
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29545", RANK="0")
    args = Arguments(model_name_or_path="opt-tiny", dataset="synthetic", context="all", neighbor_mode="embedding", peft_type="prompt",
                     max_input_length=32, max_output_length=12, max_text_neighbors=5, max_image_neighbors=2, n_text_tokens=2,
                     n_visual_tokens=2, per_device_train_batch_size=4, per_device_val_batch_size=4, dataloader_num_workers=0, epochs=1,
                     steps_per_epoch=4, val_steps_per_epoch=2, print_freq=2, grad_accumulation_steps=1, learning_rate=1e-2,
                     lr_warmup_steps=1, log_dir=str(tmp_path), seed=0, bf16=True)
    args.image_size = 32
    args.save_dir = str(tmp_path / "ckpt.pth.tar")
    try:
        res = main_worker(0, 1, args, str(tmp_path))
        assert all(torch.isfinite(torch.tensor(h["loss"])) for h in res["history"])
        assert any(n.startswith("prompt_embeddings") for n in res["engine"].names)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_trainer_prefix_tuning_synthetic_one_gpu(tmp_path):
    """run_generation with peft_type="prefix" on the decoder-only OPT (SelfAttentionModel): the trainable LM-side state is the
    prefix table, the loss goes down, the checkpoint carries it."""
    from mmgl_amd.language_modelling.run_generation import Arguments, main_worker
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", RANK="0")
    args = Arguments(model_name_or_path="opt-tiny", dataset="synthetic", context="all", neighbor_mode="embedding", peft_type="prefix",
                     max_input_length=32, max_output_length=12, max_text_neighbors=5, max_image_neighbors=2, n_text_tokens=2,
                     n_visual_tokens=2, per_device_train_batch_size=4, per_device_val_batch_size=4, dataloader_num_workers=0, epochs=2,
                     steps_per_epoch=12, val_steps_per_epoch=3, print_freq=2, grad_accumulation_steps=2, learning_rate=3e-2,
                     lr_warmup_steps=2, log_dir=str(tmp_path), seed=0, bf16=True)
    args.image_size = 32
    args.save_dir = str(tmp_path / "ckpt.pth.tar")
    torch.manual_seed(0)
    try:
        res = main_worker(0, 1, args, str(tmp_path))
        hist = res["history"]
        assert len(hist) >= 4 and hist[-1]["loss"] < hist[0]["loss"], [h["loss"] for h in hist]
        assert any(n.startswith("prefix_encoder") for n in res["engine"].names)
        assert not any(n.startswith("lm.") for n in res["engine"].names)
        ck = torch.load(args.save_dir, weights_only=False)
        assert "module.prefix_encoder.weight" in ck["state_dict"]
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_engine_fused_adamw_matches_torch_on_gpu():
    from mmgl_amd.distributed import DataParallelEngine
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 8)).cuda()
    m2 = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 8)).cuda()
    m2.load_state_dict(m.state_dict())
    eng = DataParallelEngine(m, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.01)
    assert eng.fused
    opt = torch.optim.AdamW(m2.parameters(), lr=1e-2, betas=(0.9, 0.95), weight_decay=0.01, eps=1e-8)
    for _ in range(3):
        x = torch.randn(16, 64, device="cuda")
        eng.zero_grad(); m(x).pow(2).mean().backward(); eng.finish_backward(); eng.step()
        opt.zero_grad(); m2(x).pow(2).mean().backward(); opt.step()
    for a, b in zip(m.parameters(), m2.parameters()):
        assert_close(a, b, 1e-5, "fused adamw")


def test_gcn_module_matches_reference_golden():
    from mmgl_amd.model.graph import GCN
    fx = Fixture("g8_gcn.npz")
    net = GCN(input_dim=12, output_dim=12, hidden_dim=7)
    # 12 and 7 are not whole 16-byte chunks: ops.linear zero-pads odd feature counts
    net.load_state_dict(fx.p)
    net = net.cuda()
    out = net(fx.inp["X"].cuda(), fx.inp["adj"].cuda())
    assert_close(out, fx.out["out"], 1e-3, "gcn")


def test_self_attention_model_gnn_position_type_runs():
    from mmgl_amd.model import SelfAttentionModel
    fx = Fixture("g9_selfattn_none.npz")
    torch.manual_seed(0)
    w = SelfAttentionModel(_sa_args(position_type="gnn"), None, lm_config=tiny_opt_config(dropout=0.0), text_config=tiny_roberta_config(),
                           visual_config=tiny_clip_vision_config()).cuda().eval()
    b = {k: v.cuda() for k, v in fx.inp.items()}
    from mmgl_amd.wikiweb2m import graph_pe
    edges = torch.tensor([[0, 0, 1, 2], [1, 2, 2, 3]])
    g = graph_pe.normalize_graph(graph_pe.dense_adjacency(edges, 6))
    b["graph"] = g[None].expand(2, -1, -1).contiguous().cuda()
    o = w(**b)
    o.loss.backward()
    # For a causal LM the neighbors sit AFTER the sequence with labels -100 (reference modelling_self_attention.py:323-330),
    # so they cannot influence the loss: the GNN gets an exactly-zero gradient (SURVEY.md 3.4 "no-op for causal LMs").
    assert torch.isfinite(o.loss) and o.logits.shape[1] == b["input_ids"].shape[1] + 5 * 2
    assert w.gnn.w1.weight.grad is not None and float(w.gnn.w1.weight.grad.abs().max()) == 0.0


def test_cli_entry_point_synthetic(tmp_path):
    """`python -m mmgl_amd.language_modelling.run_generation` end to end on one GPU (HfArgumentParser surface)."""
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29557")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "mmgl_amd.language_modelling.run_generation", "--model_name_or_path", "mpt-tiny", "--dataset", "synthetic",
           "--context", "text_only", "--neighbor_mode", "embedding", "--peft_type", "flamingo", "--max_input_length", "32",
           "--max_output_length", "12", "--max_text_neighbors", "4", "--n_text_tokens", "2", "--n_visual_tokens", "2",
           "--per_device_train_batch_size", "2", "--per_device_val_batch_size", "2", "--dataloader_num_workers", "0", "--epochs", "1",
           "--steps_per_epoch", "4", "--val_steps_per_epoch", "2", "--print_freq", "1", "--grad_accumulation_steps", "2",
           "--log_dir", str(tmp_path), "--bf16", "True"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "examples/sec" in r.stdout and "cider" in r.stdout
    assert os.path.exists(os.path.join(str(tmp_path), "default_0", "ckpt.pth.tar"))


def test_six_optimizer_steps_follow_the_oracle_trajectory():
    """Trainer trajectory parity (reference language_modelling/run_generation.py:321-333, 466-494): `train_loop` on the fp32 HIP path
    (dropout 0) -- 12 micro-batches, grad accumulation 2, linear warm-up over 2 optimizer steps, StepLR(step 2, gamma 0.5), fused AdamW
    over the flat buffers -- against the CPU oracle's autograd + torch.optim.AdamW + torch's own StepLR on the same batches: the lr
    sequence is equal, and after 6 optimizer steps every trainable parameter agrees to 1e-4 of its largest element (Adam divides by
    sqrt(v): the test is as tight as a first-order method allows; k_proj.bias, whose gradient is analytically zero -- softmax ignores a
    per-row score shift -- receives pure round-off and Adam turns round-off into +-lr steps, so it is compared to the INITIAL value
    plus at most 6 lr instead)."""
    import copy
    from types import SimpleNamespace
    from mmgl_amd.distributed import DataParallelEngine
    from mmgl_amd.language_modelling.run_generation import WarmupStepLR, train_loop
    from mmgl_amd.model import CrossAttentionModel
    from oracle import lm_ref, wrapper_ref

    fx = Fixture("g1_wrapper_all.npz")
    base_lr, accum, n_opt, warmup, step_size, gamma = 2e-3, 2, 6, 2, 2, 0.5
    w = CrossAttentionModel(mpt_args(context="all"), tokenizer=None, lm_config=tiny_opt_config(dropout=0.0), text_config=tiny_roberta_config(),
                            visual_config=tiny_clip_vision_config())
    load_exact(w, fx.p)
    w = w.cuda().train()
    trainable = [n for n, p in w.named_parameters() if p.requires_grad]
    T = fx.inp["input_ids"].shape[1]
    lin = T - 8                                                       # mpt_args: max_output_length 8

    def micro_batch(i):
        g = torch.Generator().manual_seed(700 + i)
        b = {k: v.clone() for k, v in fx.inp.items()}
        b["input_ids"] = torch.where(b["attention_mask"].bool(), torch.randint(3, 128, b["input_ids"].shape, generator=g), b["input_ids"])
        b["labels"] = b["input_ids"].clone()
        return b

    batches = [micro_batch(i) for i in range(accum * n_opt)]
    args = SimpleNamespace(steps_per_epoch=len(batches), grad_accumulation_steps=accum, decoder_only=True, max_input_length=lin, print_freq=1,
                           per_device_train_batch_size=batches[0]["input_ids"].shape[0])
    engine = DataParallelEngine(w, lr=base_lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01)
    sched = WarmupStepLR(base_lr, warmup, step_size, gamma)
    hist = train_loop(batches, w, None, engine, 0, sched, args)
    lrs_hip = [h["lr"] for h in hist]
    assert len(lrs_hip) == n_opt

    # ---- the oracle's trajectory: functional fp32 forward on CPU, autograd, torch.optim.AdamW, torch's StepLR after a linear warm-up
    p = {k: v.clone() for k, v in fx.p.items()}
    for k in trainable:
        p[k].requires_grad_()
    cfg = lm_ref.LMConfig(vocab_size=128, hidden_size=64, num_attention_heads=4, ffn_dim=128, num_hidden_layers=4, word_embed_proj_dim=64,
                          neighbor_layer_wise=2)
    opt = torch.optim.AdamW([p[k] for k in trainable], lr=base_lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01)
    after = torch.optim.lr_scheduler.StepLR(opt, step_size=step_size, gamma=gamma)
    lrs_ref, n_sched = [], 0
    for s in range(n_opt):
        opt.zero_grad()
        for a in range(accum):
            b = batches[s * accum + a]
            _, loss = wrapper_ref.cross_attention_model_forward(p, cfg, b, fx.out["text_last_hidden"], fx.out["visual_pooled"], "all", 2)
            (loss / accum).backward()
        # the reference's order: optimizer.step() at the current lr, then scheduler.step() -- GradualWarmupScheduler(multiplier 1) hands
        # lr = base * n / warmup for its first `warmup` calls, then defers to StepLR
        lr = base_lr * n_sched / warmup if n_sched <= warmup else after.get_last_lr()[0]
        for g in opt.param_groups:
            g["lr"] = lr
        lrs_ref.append(lr)
        opt.step()
        n_sched += 1
        if n_sched > warmup:
            after.step()
    assert lrs_hip == pytest.approx(lrs_ref, rel=1e-12, abs=0), (lrs_hip, lrs_ref)
    assert lrs_ref == pytest.approx([0.0, 1e-3, 2e-3, 2e-3, 1e-3, 1e-3], rel=1e-12)

    got = dict(w.named_parameters())
    worst = ("", 0.0)
    for k in trainable:
        ref, g0 = p[k].detach(), got[k].detach().cpu().float()
        if k.endswith("k_proj.bias"):
            assert (g0 - fx.p[k]).abs().max() <= sum(lrs_ref) * 1.01 + 1e-6, k
            continue
        err = float((g0 - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
        moved = float((ref - fx.p[k]).abs().max() / ref.abs().max().clamp_min(1e-12))
        if err > worst[1]:
            worst = (k, err, moved)
        assert err < 1e-4, (k, err, moved)
    print(f"trajectory: worst parameter {worst}")
    moved_any = max(float((p[k].detach() - fx.p[k]).abs().max()) for k in trainable)
    assert moved_any > 3e-3                                           # the six steps really moved the parameters (~ sum of lrs)


def _protocol_run(fuse, accum, n_micro, budget=49152):
    """`train_loop` over the same seeded micro-batches on the fp32 HIP path (dropout 0), fused or literal"""
    from types import SimpleNamespace
    from mmgl_amd.distributed import DataParallelEngine
    from mmgl_amd.language_modelling.run_generation import WarmupStepLR, train_loop
    from mmgl_amd.model import CrossAttentionModel
    fx = Fixture("g1_wrapper_all.npz")
    w = CrossAttentionModel(mpt_args(context="all"), tokenizer=None, lm_config=tiny_opt_config(dropout=0.0), text_config=tiny_roberta_config(),
                            visual_config=tiny_clip_vision_config())
    load_exact(w, fx.p)
    w = w.cuda().train()
    T = fx.inp["input_ids"].shape[1]

    def micro_batch(i):
        g = torch.Generator().manual_seed(900 + i)
        b = {k: v.clone() for k, v in fx.inp.items()}
        b["input_ids"] = torch.where(b["attention_mask"].bool(), torch.randint(3, 128, b["input_ids"].shape, generator=g), b["input_ids"])
        b["labels"] = b["input_ids"].clone()
        if i % 3 == 1:                     # ragged neighbor sets from one micro-batch to the next: the concatenated pass packs them all
            b["neighbor_pos_ids"] = b["neighbor_pos_ids"].clone()
            b["neighbor_pos_ids"][:, -1] = 0
        return b

    batches = [micro_batch(i) for i in range(n_micro)]
    args = SimpleNamespace(steps_per_epoch=n_micro, grad_accumulation_steps=accum, decoder_only=True, max_input_length=T - 8, print_freq=1,
                           per_device_train_batch_size=batches[0]["input_ids"].shape[0], fuse_grad_accumulation=fuse, fused_pass_tokens=budget)
    engine = DataParallelEngine(w, lr=2e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01)
    sched = WarmupStepLR(2e-3, 2, 2, 0.5)
    hist = train_loop(batches, w, None, engine, 0, sched, args)
    return hist, {n: p.detach().float().cpu().clone() for n, p in w.named_parameters() if p.requires_grad}, engine.step_count


@pytest.mark.parametrize("accum,n_micro,budget", [(4, 23, 49152), (16, 32, 49152), (4, 24, 2 * 2 * 24)])
def test_accumulation_group_as_one_pass_equals_the_literal_loop(accum, n_micro, budget):
    """The reference's batch protocol (run_generation.py:462-494; 2 x 16 in script/train_generation.sh:26-29, 4 x 4 by default) as ONE
    forward / backward pass per optimizer step against the literal per-micro-batch loop, fp32, dropout 0, 6 (or 2) optimizer steps
    incl. a short last group (23 = 5 x 4 + 3, still scaled by 1 / accum, :485) and a token budget that cuts each group in two: same
    optimizer / scheduler step counts, equal lr sequence, equal per-micro-batch summary-loss meters (to fp32 round-off: the GEMMs
    tile M = B * T differently), parameters within 1e-5 of their largest element."""
    lit, p_lit, n_lit = _protocol_run(False, accum, n_micro)
    fus, p_fus, n_fus = _protocol_run(True, accum, n_micro, budget)
    assert n_fus == n_lit == -(-n_micro // accum)
    assert all(h["passes"] == [1] * len(h["passes"]) for h in lit)
    assert all(len(h["passes"]) == (1 if budget == 49152 else 2) for h in fus)
    assert [sum(h["passes"]) for h in fus] == [len(h["passes"]) for h in lit]
    assert [h["lr"] for h in fus] == [h["lr"] for h in lit]
    assert [h["step"] for h in fus] == [h["step"] for h in lit]
    for a, b in zip(fus, lit):
        assert abs(a["loss"] - b["loss"]) <= 2e-6 * abs(b["loss"]), (a, b)
    worst = ("", 0.0)
    for k, ref in p_lit.items():
        if k.endswith("k_proj.bias"):      # analytically zero gradient: Adam turns round-off into +-lr steps (see the trajectory test)
            continue
        err = float((p_fus[k] - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
        if err > worst[1]:
            worst = (k, err)
        assert err < 1e-5, (k, err)
    print(f"fused vs literal: worst parameter {worst}")


def test_accumulation_group_as_one_pass_lora_self_attention_model():
    """The same fused-vs-literal check on the self-attention fusion path with LoRA adapters (BASELINE config 4's model class;
    reference model/modelling_self_attention.py:80-87, 282-332): the wrapper appends the neighbor tokens after the sequence and pads
    the labels with -100 ITSELF -- the input labels carry none, every micro-batch scores the same number of positions, so the group
    runs as one pass; 4 optimizer steps at accum 3, fp32, dropout 0."""
    from types import SimpleNamespace
    from mmgl_amd.distributed import DataParallelEngine
    from mmgl_amd.language_modelling.run_generation import WarmupStepLR, train_loop
    from mmgl_amd.model import SelfAttentionModel
    from mmgl_amd.model.modelling_self_attention import LoRALinear
    fx = Fixture("g9_selfattn_none.npz")
    T = fx.inp["input_ids"].shape[1]

    def run(fuse):
        torch.manual_seed(3)
        w = SelfAttentionModel(_sa_args(peft_type="lora", lora_r=8, lora_alpha=16.0, context="all"), None, lm_config=tiny_opt_config(dropout=0.0),
                               text_config=tiny_roberta_config(), visual_config=tiny_clip_vision_config()).cuda().train()
        with torch.no_grad():
            for m in w.modules():
                if isinstance(m, LoRALinear):
                    m.lora_B.normal_(std=0.05)
        batches = []
        for i in range(12):
            g = torch.Generator().manual_seed(40 + i)
            b = {k: v.clone() for k, v in fx.inp.items()}
            b["input_ids"] = torch.where(b["attention_mask"].bool(), torch.randint(3, 128, b["input_ids"].shape, generator=g), b["input_ids"])
            b["labels"] = b["input_ids"].clone()
            batches.append(b)
        args = SimpleNamespace(steps_per_epoch=12, grad_accumulation_steps=3, decoder_only=True, max_input_length=T - 8, print_freq=1,
                               per_device_train_batch_size=batches[0]["input_ids"].shape[0], fuse_grad_accumulation=fuse, fused_pass_tokens=49152)
        engine = DataParallelEngine(w, lr=2e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01)
        hist = train_loop(batches, w, None, engine, 0, WarmupStepLR(2e-3, 1, 2, 0.5), args)
        return hist, {n: p.detach().float().cpu().clone() for n, p in w.named_parameters() if p.requires_grad}

    lit, p_lit = run(False)
    fus, p_fus = run(True)
    assert [h["passes"] for h in fus] == [[3]] * 4 and [h["passes"] for h in lit] == [[1, 1, 1]] * 4
    assert [h["lr"] for h in fus] == [h["lr"] for h in lit]
    for a, b in zip(fus, lit):
        assert abs(a["loss"] - b["loss"]) <= 5e-6 * abs(b["loss"]), (a, b)
    for k, ref in p_lit.items():
        err = float((p_fus[k] - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
        assert err < 2e-5, (k, err)
