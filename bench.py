#!/usr/bin/env python
"""bench.py -- the reference's headline metric on MI355X.

    python bench.py --gpus 1 --steps 8 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

metric   : train samples/sec (whole job) of the neighbor-fusion fine-tune -- frozen RoBERTa-base + CLIP ViT-B/16
           over every neighbor, OPT-1.3B with 4 gated cross-attention layers (flamingo), 16 neighbors x 4 tokens,
           T = 512 + 128 -- one "step" = forward + backward + gradient exchange + AdamW on one synthetic
           WikiWeb2M-shaped batch per GPU (BASELINE.json configs[2] per GPU, which fits one GPU).
           samples/sec/GPU = value / n_gpus (the reference's examples_per_sec / ngpus, run_generation.py:503).
roofline : the masked cross-attention core (mmgl_xattn_fwd), the north-star kernel: HBM-bound.  achieved =
           algorithmic bytes per launch (sum over samples of 2*T*d*e + 2*S_valid*d*e) / average
           launch duration measured with HIP events on the launch stream inside the timed steps.
cpu_baseline: the CPU oracle (oracle/, fp32 torch restatement pinned to the reference's golden vectors) running the
           same step (encoders + LM forward/backward) on a bounded sample on the host cores; kind "port".
Data are synthetic (seeded) and weights random-init of the named architectures: no network in this environment.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 measured-achievable
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16
MFMA_F32_PEAK_TF = 157.3

CONFIGS = {
    # name: LM dims, neighbors, model kind.  BASELINE.json configs[2] (default, the one `metric` is quoted on), [1], [3], [4].
    "opt-1.3b": dict(kind="flamingo", lm=dict(vocab_size=50272, hidden_size=2048, num_attention_heads=32, ffn_dim=8192, num_hidden_layers=24,
                                              max_position_embeddings=2048, word_embed_proj_dim=2048), nt=11, ni=5, wise=6,
                     model_name="facebook/mpt-1.3b", batch=64, lin=512, lout=128, vocab=50272,
                     metric="train samples/sec (OPT-1.3B flamingo, 16 neighbors)"),
    "opt-125m": dict(kind="flamingo", lm=dict(vocab_size=50272, hidden_size=768, num_attention_heads=12, ffn_dim=3072, num_hidden_layers=12,
                                              max_position_embeddings=2048, word_embed_proj_dim=768), nt=2, ni=2, wise=3,
                     model_name="facebook/mpt-125m", batch=64, lin=512, lout=128, vocab=50272,
                     metric="train samples/sec (OPT-125m flamingo, 4 neighbors)"),
    # configs[3]: LoRA r=16 on q_proj / v_proj of OPT-1.3B, neighbors concatenated into the sequence (T = 640 + 64), lm_head trainable
    "opt-1.3b-lora": dict(kind="lora", lm=dict(vocab_size=50272, hidden_size=2048, num_attention_heads=32, ffn_dim=8192, num_hidden_layers=24,
                                               max_position_embeddings=2048, word_embed_proj_dim=2048), nt=11, ni=5, wise=6,
                          model_name="facebook/opt-1.3b", batch=64, lin=512, lout=128, vocab=50272, lora_r=16,
                          metric="train samples/sec (OPT-1.3B LoRA r=16, 16 neighbors, self-attention fusion)"),
    # configs[4]: Llama-2-7B dims, 32 neighbors (22 text + 10 image) x 4 tokens, max_input_length 2048 -> T = 2176, S = 128
    "llama-2-7b": dict(kind="llama", lm=dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                                             num_attention_heads=32, num_key_value_heads=32, max_position_embeddings=4096), nt=22, ni=10,
                       wise=8, model_name="meta-llama/Llama-2-7b-hf", batch=8, lin=2048, lout=128, vocab=32000,
                       metric="train samples/sec (Llama-2-7B flamingo, 32 neighbors, T=2176)"),
}


def hf_configs(cfg):
    from transformers import CLIPVisionConfig, LlamaConfig, OPTConfig, RobertaConfig
    if cfg["kind"] == "llama":
        lm = LlamaConfig(pad_token_id=0, bos_token_id=1, eos_token_id=2, attention_dropout=0.0, **cfg["lm"])
    else:
        lm = OPTConfig(do_layer_norm_before=True, dropout=0.1, attention_dropout=0.0, pad_token_id=1, bos_token_id=2,
                       eos_token_id=2, **cfg["lm"])
    txt = RobertaConfig(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                        max_position_embeddings=514, pad_token_id=1, type_vocab_size=1)
    vis = CLIPVisionConfig(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                           image_size=224, patch_size=16, projection_dim=512)
    return lm, txt, vis


def make_args(cfg, **kw):
    from mmgl_amd.language_modelling.run_generation import Arguments
    peft = {"flamingo": "flamingo", "llama": "flamingo", "lora": "lora"}[cfg["kind"]]
    a = Arguments(model_name_or_path=cfg["model_name"], context="all", neighbor_mode="embedding", peft_type=peft,
                  max_text_neighbors=cfg["nt"], max_image_neighbors=cfg["ni"], decoder_only=True, max_input_length=cfg["lin"],
                  max_output_length=cfg["lout"])
    a.neighbor_layer_wise = cfg["wise"]
    if cfg["kind"] == "lora":
        a.lora_r, a.lora_alpha, a.lora_dropout, a.position_type = cfg["lora_r"], 32.0, 0.0, "none"
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def synthetic_batch(B, cfg, seed, device, Ln=512):
    """WikiWeb2M-shaped batch of SURVEY.md 8(d): ragged prompt / summary / neighbor lengths, ragged neighbor counts,
    random interleave of the valid slots, padding slots last."""
    g = torch.Generator().manual_seed(seed)
    Nt, Ni = cfg["nt"], cfg["ni"]
    Lin, Lout, vocab = cfg["lin"], cfg["lout"], cfg["vocab"]
    T = Lin + Lout
    ids = torch.randint(3, vocab, (B, T), generator=g)
    am = torch.ones(B, T, dtype=torch.long)
    nids = torch.randint(3, 50265, (B, Nt, Ln), generator=g)
    nam = torch.ones(B, Nt, Ln, dtype=torch.long)
    npos = torch.zeros(B, Nt, dtype=torch.long)
    ipos = torch.zeros(B, Ni, dtype=torch.long)
    tloc = torch.zeros(B, Nt, dtype=torch.long)
    iloc = torch.zeros(B, Ni, dtype=torch.long)
    imgs = torch.zeros(B, Ni, 3, 224, 224)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for b in range(B):
        lp, ls = ri(64, Lin), ri(8, 64)
        ids[b, lp:Lin] = 1; am[b, lp:Lin] = 0
        ids[b, Lin + ls - 1] = 2; ids[b, Lin + ls:] = 1; am[b, Lin + ls:] = 0
        nt, ni = ri(1, Nt), ri(0, Ni)
        for j in range(Nt):
            if j < nt:
                ln = ri(16, Ln)
                nids[b, j, 0] = 0; nids[b, j, ln - 1] = 2; nids[b, j, ln:] = 1; nam[b, j, ln:] = 0
                npos[b, j] = j + 1
            else:
                nids[b, j] = 1; nids[b, j, 0] = 0; nids[b, j, 1] = 2; nam[b, j, 2:] = 0
        for j in range(ni):
            imgs[b, j] = torch.randn(3, 224, 224, generator=g)
            ipos[b, j] = j + 1
        kinds = ["t"] * (nt - 1) + ["i"] * ni
        perm = torch.randperm(len(kinds), generator=g).tolist()
        order = ["t"] + [kinds[q] for q in perm]
        ti = ii = 0
        for loc, kd in enumerate(order):
            if kd == "t":
                tloc[b, ti] = loc; ti += 1
            else:
                iloc[b, ii] = loc; ii += 1
        loc = len(order)
        for j in range(nt, Nt):
            tloc[b, j] = loc; loc += 1
        for j in range(ni, Ni):
            iloc[b, j] = loc; loc += 1
    batch = dict(input_ids=ids, attention_mask=am, labels=ids.clone(), neighbor_input_ids=nids, neighbor_attention_mask=nam,
                 neighbor_pos_ids=npos, text_locations=tloc, neighbor_images=imgs, neighbor_images_pos_ids=ipos,
                 image_locations=iloc)
    valid_keys = [(int((npos[b] > 0).sum()) + int((ipos[b] > 0).sum())) * 4 for b in range(B)]
    from mmgl_amd.model.modelling_cross_attention import host_metadata
    meta = host_metadata(batch)                      # what the trainer reads off the collated batch before the H2D copy
    batch = {k: v.to(device) for k, v in batch.items()}
    batch["host_meta"] = meta
    return batch, valid_keys


def _cpu_threads():
    # threads: torch/MKL fp32 GEMM peaks at ~32 threads on the GPU box's 2 x 64-core EPYC 9575F and collapses beyond
    # (tools/probes/cpu_threads.py: 1.44 TFLOP/s @32, 0.64 @64, 0.07 @256) -> min(32, cpu_count) threads, BOTH numbers reported
    return min(32, os.cpu_count() or 1)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median_time(fn, warmups, iters):
    for _ in range(warmups):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts


def _cpu_step(model, cfg, lm_cfg, batch, n_samples):
    """closure running the CPU oracle's train step (frozen encoders fwd, LM fwd + bwd w.r.t. the trainable set) on `n_samples`
    samples of `batch`; returns the loss"""
    from oracle import lm_ref, wrapper_ref
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    trainable = [k for k, p in model.named_parameters() if p.requires_grad]
    for k in trainable:
        sd[k].requires_grad_()
    text_model = model.text_model.float().cpu()
    visual_model = model.visual_model.float().cpu()
    b = {k: v[:n_samples].cpu() for k, v in batch.items() if k != "host_meta"}
    ocfg = lm_ref.LMConfig(vocab_size=lm_cfg.vocab_size, hidden_size=lm_cfg.hidden_size, num_attention_heads=lm_cfg.num_attention_heads,
                           ffn_dim=lm_cfg.ffn_dim, num_hidden_layers=lm_cfg.num_hidden_layers,
                           word_embed_proj_dim=lm_cfg.word_embed_proj_dim, neighbor_layer_wise=cfg["wise"])

    def once():
        for k in trainable:
            sd[k].grad = None
        with torch.no_grad():
            L = b["neighbor_input_ids"].shape[-1]
            tl = text_model(input_ids=b["neighbor_input_ids"].reshape(-1, L), attention_mask=b["neighbor_attention_mask"].reshape(-1, L)).last_hidden_state
            vp = visual_model(b["neighbor_images"].reshape(-1, 3, 224, 224)).pooler_output
        logits, loss = wrapper_ref.cross_attention_model_forward(sd, ocfg, b, tl, vp, "all", 4)
        loss.backward()
        return float(loss.detach())
    return once


def cpu_protocol(cores):
    """BASELINE.md section 3, items (i)-(iv), on the host cores with the CPU oracle (fp32): attention core fwd and fwd + bwd, one gated
    cross-attention layer fwd + bwd, the 4-layer gated stack at OPT-1.3B dims, the full train step at OPT-125m dims (config 2).
    Median of 5 after 2 warm-ups each; B = 2, T = 640, S = 64 (40 valid keys in sample 0), seed 1234; FLOPs = SURVEY.md 8(d)."""
    from oracle import lm_ref
    B, T, S, d, H, ffn = 2, 640, 64, 2048, 32, 8192
    g = torch.Generator().manual_seed(1234)
    rn = lambda *sh, std=1.0: torch.randn(*sh, generator=g) * std
    valid = torch.ones(B, S, dtype=torch.long)
    valid[0, 40:] = 0
    s_valid = [40, 64]
    add_mask = lm_ref.expand_mask(valid, torch.float32, T)
    ocfg = lm_ref.LMConfig(vocab_size=50272, hidden_size=d, num_attention_heads=H, ffn_dim=ffn, num_hidden_layers=24, word_embed_proj_dim=d,
                           neighbor_layer_wise=6)

    def layer_params(pre):
        p = {}
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            p[f"{pre}self_attn.{n}.weight"], p[f"{pre}self_attn.{n}.bias"] = rn(d, d, std=0.02), rn(d, std=0.02)
        p[pre + "fc1.weight"], p[pre + "fc1.bias"] = rn(ffn, d, std=0.02), rn(ffn, std=0.02)
        p[pre + "fc2.weight"], p[pre + "fc2.bias"] = rn(d, ffn, std=0.02), rn(d, std=0.02)
        for n in ("self_attn_layer_norm", "final_layer_norm"):
            p[f"{pre}{n}.weight"], p[f"{pre}{n}.bias"] = torch.ones(d), torch.zeros(d)
        p[pre + "gating1"], p[pre + "gating2"] = torch.tensor([0.5]), torch.tensor([0.5])
        for v in p.values():
            v.requires_grad_()
        return p

    core_flops = sum(4.0 * T * sv * d for sv in s_valid)
    block_flops = sum(4.0 * T * d * d + 4.0 * sv * d * d for sv in s_valid) + core_flops
    layer_flops = block_flops + B * 4.0 * T * d * ffn
    q, k, v = rn(B, T, d).requires_grad_(), rn(B, S, d).requires_grad_(), rn(B, S, d).requires_grad_()
    hidden, ne = rn(B, T, d), rn(B, S, d)
    out = {}

    def rec(name, fn, flops, samples):
        med, ts = _median_time(fn, 2, 5)
        out[name] = {"ms": round(med * 1e3, 3), "gflops": round(flops / med / 1e9, 1), "samples_per_s": round(samples / med, 3),
                     "runs_ms": [round(t * 1e3, 2) for t in ts]}

    def core_fwd():
        with torch.no_grad():
            lm_ref.attention_core(q, k, v, add_mask, H)

    def core_fwd_bwd():
        for t in (q, k, v):
            t.grad = None
        lm_ref.attention_core(q, k, v, add_mask, H).sum().backward()

    rec("attention_core_fwd", core_fwd, core_flops, B)
    rec("attention_core_fwd_bwd", core_fwd_bwd, 3 * core_flops, B)
    layers = [layer_params(f"l{i}.") for i in range(4)]

    def stack(n):
        def fn():
            h = hidden
            for i in range(n):
                for t in layers[i].values():
                    t.grad = None
                h = lm_ref.decoder_layer(layers[i], f"l{i}.", h, None, ocfg, ne, add_mask, cross=True)
            h.sum().backward()
        return fn

    rec("gated_layer_fwd_bwd", stack(1), 3 * layer_flops, B)
    rec("gated_stack4_fwd_bwd", stack(4), 4 * 3 * layer_flops, B)
    del layers

    # (iv) the full train step at config 2's dimensions (OPT-125m, 2 + 2 neighbors), 2 samples
    from mmgl_amd.model import CrossAttentionModel
    cfg2 = CONFIGS["opt-125m"]
    lm2, txt2, vis2 = hf_configs(cfg2)
    torch.manual_seed(1234)
    with torch.device("cpu"):
        m2 = CrossAttentionModel(make_args(cfg2), tokenizer=None, lm_config=lm2, text_config=txt2, visual_config=vis2)
    with torch.no_grad():
        for n_, p_ in m2.named_parameters():
            if n_.endswith("gating1") or n_.endswith("gating2"):
                p_.fill_(0.5)
    b2, _ = synthetic_batch(2, cfg2, seed=1234, device=torch.device("cpu"))
    step2 = _cpu_step(m2, cfg2, lm2, b2, 2)
    med, ts = _median_time(step2, 2, 5)
    out["config2_full_step"] = {"ms": round(med * 1e3, 1), "samples_per_s": round(2 / med, 3), "runs_ms": [round(t * 1e3, 1) for t in ts],
                                "what": "opt-125m flamingo, 2+2 neighbors, T=640: frozen encoders fwd + LM fwd + bwd, 2 samples"}
    out["protocol"] = f"BASELINE.md section 3: fp32 CPU oracle, B=2, T=640, S=64 (40 / 64 valid keys), d=2048, H=32, ffn=8192; median of 5 after 2 warm-ups; {cores} threads"
    return out


def cpu_baseline(model, cfg, lm_cfg, batch, n_samples=1, repeats=3, protocol=True, all_threads_too=False):
    """The CPU oracle on the host cores, same bench run (kind "port"; the reference's Python does not travel).  `value`: the bench's own
    workload -- config 3's train step (frozen encoders fwd, LM fwd + bwd w.r.t. the trainable set) on a bounded sample: one warm-up,
    then the median of `repeats` runs of `n_samples` sample(s) (6 s each: the section-3 count of 2 + 5 would be 40 s).  `section3`:
    BASELINE.md section 3's items (i)-(iv) at their own sizes, median of 5 after 2 warm-ups.  fp32."""
    cores = _cpu_threads()
    torch.set_num_threads(cores)
    once = _cpu_step(model, cfg, lm_cfg, batch, n_samples)
    loss = once()                                       # warm-up (thread pool, allocator)
    dt, runs = _median_time(once, 0, repeats)
    all_threads = None
    if all_threads_too and os.cpu_count() and os.cpu_count() != cores:
        # BASELINE.md section 3 says os.cpu_count() threads: one run of the same sample with every logical CPU, beside the figure
        # above (the reason `cores` is 32: torch's fp32 GEMM does not scale past that on this host class).  Opt-in (--cpu-all-threads):
        # it takes minutes on a 256-thread host, the default run has to finish in a few; the recorded figure is under profiles/.
        torch.set_num_threads(os.cpu_count())
        dt_all, _ = _median_time(once, 0, 1)
        all_threads = dict(cores=os.cpu_count(), value=n_samples / dt_all, seconds=round(dt_all, 2))
        torch.set_num_threads(cores)
    model.text_model.to(batch["input_ids"].device)
    model.visual_model.to(batch["input_ids"].device)
    out = dict(value=n_samples / dt, unit="samples/s", cores=cores, cpu_count=os.cpu_count(), cpu_model=_cpu_model(), kind="port",
               seconds=round(dt, 2), runs_s=[round(r, 2) for r in runs],
               sample=f"median of {repeats} runs (after 1 warm-up) of {n_samples} sample(s) of the same synthetic batch: frozen encoders fwd + LM "
                      f"fwd + bwd (no optimizer step), fp32 torch oracle (oracle/), {cores} threads of {os.cpu_count()} logical CPUs "
                      f"(torch's fp32 GEMM collapses beyond 32 threads on this host class)", loss=loss)
    if all_threads is not None:
        out["all_threads"] = all_threads
    if protocol:
        out["section3"] = cpu_protocol(cores)
    return out


def pmc_traffic(B, lm_cfg, cfg, dtype):
    """HBM bytes per launch of xattn_fwd_kernel from the committed PMC collection (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes, gfx950 x2 fetch correction applied: profiles/r<round>_pmc_xattn_*.json) when it was taken at this exact
    shape, and the file it came from; (None, None) otherwise -- PMC counters cannot be read from inside this process, so `traffic` is
    a committed measurement of the same kernel at the same shape, NOT a counter of this run (`traffic_source` says which file)."""
    for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):                     # the latest round's collection first
        rel = os.path.join("profiles", f"{rnd}_pmc_xattn_B{B}_{dtype}.json")
        try:
            with open(os.path.join(ROOT, rel)) as f:
                d = json.load(f)
            c = d["config"]
            if (c["B"], c["H"], c["S"], c["D"]) == (B, lm_cfg.num_attention_heads, (cfg["nt"] + cfg["ni"]) * 4,
                                                   lm_cfg.hidden_size // lm_cfg.num_attention_heads) and c["T"] == 640:
                return d["kernels"]["xattn_fwd_kernel"]["hbm_bytes_per_launch"], rel
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="opt-1.3b", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="0 = the config's default (64 for opt-1.3b).  " + "per-GPU batch (reference default 4).  Sized for the 288 GB HBM and the 256-CU GEMM grid: 16 -> 197, 32 -> 208, 48 -> 233, 56 -> 229, 64 -> 243, 72 -> 233, 80 -> 238 samples/s on one MI355X (the dips are library-GEMM heuristics)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=1)
    ap.add_argument("--cpu-all-threads", action="store_true", help="cpu_baseline: also time ONE run of the same sample with os.cpu_count() threads "
                    "(`all_threads`; minutes on a 256-thread host)")
    ap.add_argument("--force-exchange", action="store_true", help="--gpus 1 only: create a world_size-1 RCCL ('nccl') process group on the GPU and run "
                    "the data-parallel engine with its gradient exchange forced on (hook-launched async all-reduces, work.wait(), dynamic GEMM "
                    "tile schedule), reporting the `exchange` block: the device-side N>1 path on a 1-GPU box")
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the `batch_sweep` block (the same step at per-GPU batches 4 / 16 / 32 / 64 up "
                    "to the run's own batch, outside the timed region)")
    ap.add_argument("--no-protocol", action="store_true", help="skip `at_reference_protocol` (the reference's 4 x 4 and 2 x 16 batch protocols "
                    "through the trainer's train_loop, fused and literal)")
    ap.add_argument("--ref-batch", type=int, default=4, help="also report the step at the reference's default per-device batch (Arguments default 4, run_generation.py:124-126); 0 = skip")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched bare (`python bench.py --gpus N`): spawn the N ranks ourselves, one process per GPU, the way the reference
        # does with mp.spawn (language_modelling/run_generation.py:265-266); the re-executed ranks land in the branch below
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` or torchrun --nproc-per-node N")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("MMGL_DIST_BACKEND", "nccl")        # "gloo": dry run of the N-rank path on fewer GPUs than ranks
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} visible GPU(s)")
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist
    forced = bool(args.force_exchange) and world == 1
    if args.force_exchange and world != 1:
        raise SystemExit("--force-exchange is the 1-GPU stand-in for the N>1 exchange: use it with --gpus 1")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    elif forced:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local_rank))
        dist.init_process_group(backend, rank=0, world_size=1, **(dict(device_id=device) if backend == "nccl" else {}))
    comm = world > 1 or forced                        # collectives are issued (RCCL) -- with one rank only under --force-exchange

    from mmgl_amd import _lib
    from mmgl_amd.distributed import DataParallelEngine
    from mmgl_amd.language_modelling.run_generation import _summary_cross_entropy
    from mmgl_amd.model import CrossAttentionModel, SelfAttentionModel
    _lib.lib()                                        # fail loudly if the HIP extension is missing

    cfg = CONFIGS[args.config]
    if not args.batch:
        args.batch = cfg["batch"]
    lm_cfg, txt_cfg, vis_cfg = hf_configs(cfg)
    margs = make_args(cfg)
    torch.manual_seed(1234)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cls = SelfAttentionModel if cfg["kind"] == "lora" else CrossAttentionModel
    if cfg["kind"] == "llama":                        # 7 B parameters: random-initialised directly on the GPU in the compute dtype
        torch.set_default_dtype(dtype)
        with torch.device(device):
            model = cls(margs, tokenizer=None, lm_config=lm_cfg, text_config=txt_cfg, visual_config=vis_cfg)
        torch.set_default_dtype(torch.float32)
    else:
        with torch.device("cpu"):
            model = cls(margs, tokenizer=None, lm_config=lm_cfg, text_config=txt_cfg, visual_config=vis_cfg)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith("gating1") or n_.endswith("gating2"):
                p.fill_(0.5)                          # numerically live cross-attention (init value 0 = identity)
            if n_.endswith("lora_B"):
                p.normal_(std=0.02)                   # numerically live adapters (init value 0 = identity)
    model = model.to(dtype).to(device)
    model.train()
    engine = DataParallelEngine(model, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.01, force_exchange=forced)
    n_train = sum(p.numel() for p in engine.params)

    batch, valid_keys = synthetic_batch(args.batch, cfg, seed=1234 + rank, device=device)
    T = batch["input_ids"].shape[1]
    d = lm_cfg.hidden_size
    esize = 2 if dtype == torch.bfloat16 else 4
    # the roofline kernel: the north-star cross-attention core (HBM-bound); the LoRA config has none -- its dominant kernel is the
    # ping-pong GEMM that carries the base projections and the low-rank update in its epilogue (MFMA-bound)
    roof_name = "mmgl_gemm_nt" if cfg["kind"] == "lora" else "mmgl_xattn_fwd"

    n_steps_run = [0]
    last_meter = [0.0]

    # the trainer's step (reference run_generation.py:466-485; mmgl_amd train_loop): besides the token loss that is differentiated,
    # the logits of the summary positions and their cross-entropy for the running meter.  The meter value stays a device scalar, as in
    # train_loop (_Meters: resolved every print_freq optimizer steps, not per step): the host launches the next step while this one
    # computes.  `at_reference_protocol` below goes through train_loop itself.
    lin = cfg["lin"]
    summary = slice(lin, T - 1)

    def step():
        out = model(**batch, logits_slice=summary)
        lg = out.logits.detach()
        last_meter[0] = _summary_cross_entropy(lg, batch["labels"][..., lin + 1:], 1)
        out.loss.backward()
        engine.finish_backward()
        engine.step()
        engine.zero_grad()
        n_steps_run[0] += 1
        return out.loss

    for _ in range(args.warmup):
        loss = step()
    timing = not args.no_kernel_timing
    _lib.KernelTimer.reset()
    _lib.KernelTimer.enabled = timing
    _lib.KernelTimer.only = {roof_name}               # timed region: HIP events around the roofline kernel only
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.KernelTimer.enabled = False
    roof = _lib.KernelTimer.summary().get(roof_name) if timing else None
    table_steps = 0
    if timing:                                        # per-entry-point table: two more steps with events around EVERY call (the
        _lib.KernelTimer.reset()                      # ~1700 event pairs per step cost 1-2 % of throughput, so not in `value`)
        _lib.KernelTimer.only = None
        _lib.KernelTimer.enabled = True
        table_steps = 2
        for _ in range(table_steps):
            loss = step()
        torch.cuda.synchronize()
        _lib.KernelTimer.enabled = False
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- the same step at other per-GPU batches, outside the timed region: the reference's own default (launch-bound regime: GEMM
    # M = 4 * 640) as `at_reference_batch`, and {4, 16, 32, 64} below the run's batch as `batch_sweep` (round 4 picked bench batches
    # that land on whole rounds of 256x256 GEMM tiles; the default is a plain 64 again and the dependence is on the record)
    def at_batch(bsz, seed, warm, n):
        nonlocal batch
        keep = batch
        batch, _ = synthetic_batch(bsz, cfg, seed=seed + rank, device=device)
        for _ in range(warm):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        t1 = torch.tensor([time.perf_counter() - t1], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t1, op=dist.ReduceOp.MAX)
        t1 = float(t1.item())
        batch = keep
        return {"per_gpu_batch": bsz, "value": round(world * bsz * n / t1, 3), "unit": "samples/s", "ms_per_step": round(1e3 * t1 / n, 3), "steps": n}

    # ---- the reference's own batch protocol THROUGH THE TRAINER (mmgl_amd train_loop, the function a user of the reference's
    # settings runs): per_device_train_batch_size x grad_accumulation_steps = 4 x 4 (Arguments defaults, run_generation.py:124-126)
    # and 2 x 16 (script/train_generation.sh:26-29), host micro-batches in pinned memory as the DataLoader hands them over (the H2D
    # copies are inside the measurement), the group as ONE pass (default) beside the literal per-micro-batch loop
    def at_protocol(per_device, accum, fuse, n_opt, warm_opt):
        from types import SimpleNamespace
        from mmgl_amd.language_modelling.run_generation import WarmupStepLR, train_loop
        mbs = []
        for i in range(accum * max(n_opt, warm_opt)):
            hb, _ = synthetic_batch(per_device, cfg, seed=9000 + 100 * rank + i, device=torch.device("cpu"))
            hb.pop("host_meta")
            mbs.append({k: v.pin_memory() for k, v in hb.items()})
        sched = WarmupStepLR(1e-4, 0, 1 << 30, 1.0)

        def run(n_groups):
            targs = SimpleNamespace(steps_per_epoch=accum * n_groups, grad_accumulation_steps=accum, decoder_only=True,
                                    max_input_length=lin, print_freq=1 << 30, per_device_train_batch_size=per_device,
                                    fuse_grad_accumulation=fuse, fused_pass_tokens=49152)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            train_loop(mbs[:accum * n_groups], model, None, engine, 0, sched, targs)
            torch.cuda.synchronize()
            return time.perf_counter() - t1
        run(warm_opt)
        t1 = torch.tensor([run(n_opt)], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t1, op=dist.ReduceOp.MAX)
        t1 = float(t1.item())
        return {"per_device_train_batch_size": per_device, "grad_accumulation_steps": accum, "one_pass_per_optimizer_step": fuse,
                "value": round(world * per_device * accum * n_opt / t1, 3), "unit": "samples/s",
                "ms_per_optimizer_step": round(1e3 * t1 / n_opt, 3), "optimizer_steps": n_opt}

    protocol = None
    # (single-GPU diagnostics: the N > 1 lines of the scaling run carry `value`, `exchange` and the sweep only)
    if not args.no_protocol and world == 1:
        protocol = {"through": "mmgl_amd.language_modelling.run_generation.train_loop (host micro-batches in pinned memory, H2D inside)"}
        for per_device, accum in ((4, 4), (2, 16)):
            # warm-up = one pass over the SAME micro-batches (the packed encoder buffers have data-dependent sizes: the first time a
            # group's shapes are seen the caching allocator grows -- hipMalloc, a device sync -- which a 2000-step epoch amortises)
            n_f, n_l = 256 // (per_device * accum), 64 // (per_device * accum)
            if cfg["kind"] == "llama":                # 2176-token samples: ~0.85 s per 16 of them
                n_f, n_l = 2, 1
            protocol[f"{per_device}x{accum}"] = at_protocol(per_device, accum, True, n_f, n_f)
            protocol[f"{per_device}x{accum}_literal"] = at_protocol(per_device, accum, False, n_l, n_l)

    # ---- evaluate_loop (reference run_generation.py:527-703) at the reference's per_device_val_batch_size 2: teacher-forced forward,
    # summary-loss meter, argmax, decode, BLEU / CIDEr over the captions -- several validation batches per forward pass (default)
    # beside one forward per batch.  Token ids are decoded as numbers (no vocabulary here): the scoring cost is the real one.
    def at_eval(per_device, n_batches, fuse):
        from types import SimpleNamespace
        from mmgl_amd.language_modelling.run_generation import evaluate_loop

        class IdTokenizer:
            pad_token_id = 1

            def batch_decode(self, ids, skip_special_tokens=True):
                return [" ".join(str(t) for t in row if t > 2) for row in ids.tolist()]

        mbs = []
        for i in range(n_batches):
            hb, _ = synthetic_batch(per_device, cfg, seed=7000 + 100 * rank + i, device=torch.device("cpu"))
            hb.pop("host_meta")
            mbs.append({k: v.pin_memory() for k, v in hb.items()})
        eargs = SimpleNamespace(val_steps_per_epoch=n_batches, print_freq=1 << 30, decoder_only=True, max_input_length=lin,
                                fuse_eval_batches=fuse, fused_pass_tokens=49152)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            evaluate_loop(mbs[:max(2, n_batches // 4)], model, IdTokenizer(), 0, eargs, prefix="bench-warmup")
            evaluate_loop(mbs, model, IdTokenizer(), 0, eargs, prefix="bench")
        model.train()
        return {"per_device_val_batch_size": per_device, "batches": n_batches, "batches_share_a_forward": fuse,
                "value": round(world * evaluate_loop.samples_per_sec, 3), "unit": "samples/s"}

    eval_line = None
    if not args.no_protocol and world == 1 and cfg["kind"] == "flamingo":
        eval_line = {"through": "mmgl_amd.language_modelling.run_generation.evaluate_loop (incl. decode + BLEU / CIDEr scoring on the host)",
                     "grouped": at_eval(2, 64, True), "literal": at_eval(2, 32, False)}

    ref_line = None
    if args.ref_batch and args.ref_batch < args.batch:
        ref_line = at_batch(args.ref_batch, 4321, 3, 10)
    sweep = None
    if not args.no_batch_sweep:
        sweep = {}
        for bsz in (4, 16, 32, 64):
            if bsz < args.batch:
                sweep[str(bsz)] = ref_line["value"] if (ref_line and bsz == args.ref_batch) else at_batch(bsz, 4321, 2, 6)["value"]
        sweep[str(args.batch)] = round(world * args.batch * args.steps / dt, 3)

    # ---- gradient-exchange report (outside the timed region; every rank runs the same collectives)
    exchange = None
    if comm:
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)                                         # = number of ranks RCCL actually connected
        bytes_per_step = engine.exchange_bytes / max(1, n_steps_run[0])

        def timed(fn, n):
            dist.barrier()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t) / n], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        def bare_exchange():                                          # the same bucket all-reduces with nothing to overlap with
            for b in engine.buckets:
                dist.all_reduce(engine.flat_grad[b["start"]:b["end"]])
        bare_exchange()
        ar_s = timed(bare_exchange, 3)
        engine.sync = False                                           # the same step without the exchange
        step()
        nosync_s = timed(step, 3)
        engine.sync = True
        step_s = dt / args.steps
        exposed = max(0.0, step_s - nosync_s)
        exchange = {"rccl_ranks": int(ones.item()), "backend": backend, "forced": forced, "buckets": len(engine.buckets), "bucket_mb": round(engine.bucket_mb, 2),
                    "first_bucket_fraction_of_gradient": round((engine.buckets[0]["end"] - engine.buckets[0]["start"]) / engine.numel, 4),
                    "exchange_bytes_per_step_per_gpu": int(bytes_per_step),
                    "wire_bytes_per_step_per_gpu": int(2 * (world - 1) / world * bytes_per_step),
                    "allreduce_ms_alone": round(ar_s * 1e3, 3), "step_ms_without_exchange": round(nosync_s * 1e3, 3),
                    "exposed_ms": round(exposed * 1e3, 3),
                    "overlap_fraction": round(1.0 - min(1.0, exposed / ar_s), 4) if ar_s > 0 else None,
                    "algbw_GBps": round(bytes_per_step / ar_s / 1e9, 1) if ar_s > 0 else None}

    if rank == 0:
        value = world * args.batch * args.steps / dt
        line = {
            "metric": cfg["metric"],
            "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic WikiWeb2M-shaped batch (seeded), random-init weights",
            "samples_per_sec_per_gpu": round(value / world, 3),
            "config": {"workload": f"{args.config} context=all neighbor_mode=embedding peft={margs.peft_type}"
                                   + (f" r={cfg['lora_r']} (q_proj, v_proj; lm_head trainable), neighbors concatenated into the sequence" if cfg["kind"] == "lora" else "")
                                   + f", {cfg['nt']}+{cfg['ni']} neighbors x 4 tokens, T={cfg['lin']}+{cfg['lout']}, roberta-base + clip-vit-base-patch16 frozen "
                                     f"encoders, full train step (fwd+bwd+exchange+AdamW)",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "seq_len": T, "neighbor_keys": (cfg["nt"] + cfg["ni"]) * 4,
                       "trainable_params": n_train, "parallelism": f"dp{world}", "loss": round(float(loss.detach()), 4)},
        }
        if timing:
            ks = _lib.KernelTimer.summary()
            x = roof
            if x and cfg["kind"] == "lora":
                tf = x["flops"] / (x["ms_total"] * 1e-3) / 1e12
                pk = MFMA_BF16_PEAK_TF if dtype == torch.bfloat16 else MFMA_F32_PEAK_TF
                line["roofline"] = {"kernel": "gemm8p_kernel (mmgl_gemm_nt: frozen projections, LoRA update in the epilogue)", "bound": "mfma", "achieved": round(tf, 1),
                                    "peak": pk, "unit": "TFLOP/s", "frac": round(tf / pk, 4), "traffic": None,
                                    "us_per_launch": round(x["ms_avg"] * 1e3, 2), "launches": x["calls"],
                                    "algorithmic_flops_per_launch": x["flops"] / x["calls"]}
            elif x:
                alg = sum(2.0 * T * d * esize + 2.0 * sv * d * esize for sv in valid_keys)     # bytes per launch
                traffic, traffic_src = pmc_traffic(args.batch, lm_cfg, cfg, args.dtype)
                flops = sum(4.0 * T * sv * d for sv in valid_keys)
                sec = x["ms_avg"] * 1e-3
                gbs = alg / sec / 1e9
                line["roofline"] = {"kernel": "xattn_fwd_kernel (mmgl_xattn_fwd)", "bound": "hbm", "achieved": round(gbs, 1),
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                    "traffic": traffic, "traffic_source": traffic_src,
                                    "us_per_launch": round(x["ms_avg"] * 1e3, 2), "launches": x["calls"],
                                    "algorithmic_bytes_per_launch": alg, "tflops": round(flops / sec / 1e12, 2),
                                    "frac_of_bf16_mfma_peak": round(flops / sec / 1e12 / MFMA_BF16_PEAK_TF, 4)}
            peak_tf = MFMA_BF16_PEAK_TF if dtype == torch.bfloat16 else MFMA_F32_PEAK_TF
            kern = {}
            for name, s in sorted(ks.items(), key=lambda kv: -kv[1]["ms_total"]):
                e = {"calls": s["calls"], "ms_avg": round(s["ms_avg"], 4), "ms_total": round(s["ms_total"], 2)}
                if s["flops"]:
                    tf = s["flops"] / (s["ms_total"] * 1e-3) / 1e12
                    e.update(bound="mfma", tflops=round(tf, 1), frac=round(tf / peak_tf, 4))
                elif s["bytes"]:
                    gb = s["bytes"] / (s["ms_total"] * 1e-3) / 1e9
                    e.update(bound="hbm", gbs=round(gb, 1), frac=round(gb / HBM_PEAK_GBS, 4))
                kern[name] = e
            xb = ks.get("mmgl_xattn_bwd")
            if xb:
                alg_b = sum(3.0 * T * d * esize + 4.0 * sv * d * esize for sv in valid_keys)
                gb = alg_b / (xb["ms_avg"] * 1e-3) / 1e9
                kern["mmgl_xattn_bwd"].update(bound="hbm", gbs=round(gb, 1), frac=round(gb / HBM_PEAK_GBS, 4))
            line["kernels"] = kern
            # what bounds the MFMA entries above (a committed measurement of this kernel, not a counter of this run): per 256x256 tile
            # t = a K + X -- the steady state of the ping-pong phases plus a K-independent tile boundary; the L2->LDS operand path is at
            # 0.20-0.30 of its 63.6 B/clk/CU at every shape (tools/probes/gemm_tile_model.py)
            kdim = lm_cfg.hidden_size
            line["gemm_bound"] = {"source": "profiles/r6_gemm_tile_model.txt", "kernel": "gemm8p_kernel (mmgl_gemm_nt / mmgl_linear_*)",
                                  "bound": "mfma steady state (0.55-0.62 of the nominal-clock pipe: 1.8 GHz sustained x ping-pong phases) + "
                                           "tile boundary (3.3-5.9 us per 256x256 tile); lds-dma operand path 0.20-0.30 utilised",
                                  "ceiling_frac_of_peak_by_K": {"768": [0.42, 0.51], "2048": [0.50, 0.56], "3072": [0.51, 0.58], "8192": [0.54, 0.60]},
                                  "this_config_K": [kdim, getattr(lm_cfg, "ffn_dim", None) or getattr(lm_cfg, "intermediate_size", None)]}
            # BASELINE.json's second metric, "cross-attn TFLOPS % of peak": every mmgl_linear_* call of this step belongs to the
            # gated cross-attention layers (projections + FFN; the frozen layers' GEMMs go through mmgl_gemm_nt), so the layer-level
            # rate is (their FLOPs + the attention core's) / (their time + the core's time), forward and backward.
            lf, lb_, xf, xb2 = (ks.get(n) for n in ("mmgl_linear_fwd", "mmgl_linear_bwd", "mmgl_xattn_fwd", "mmgl_xattn_bwd"))
            if lf and lb_ and xf and xb2:
                core_f = sum(4.0 * T * sv * d for sv in valid_keys) * xf["calls"]
                fwd_tf = (lf["flops"] + core_f) / ((lf["ms_total"] + xf["ms_total"]) * 1e-3) / 1e12
                bwd_tf = (lb_["flops"] + 2.0 * core_f) / ((lb_["ms_total"] + xb2["ms_total"]) * 1e-3) / 1e12
                line["cross_attention_layers"] = {"fwd_tflops": round(fwd_tf, 1), "fwd_frac_of_peak": round(fwd_tf / peak_tf, 4),
                                                  "bwd_tflops": round(bwd_tf, 1), "bwd_frac_of_peak": round(bwd_tf / peak_tf, 4),
                                                  "peak_tflops": peak_tf, "scope": "GEMMs + attention core of the 4 gated cross-attention layers"}
            line["kernels_note"] = f"per C-ABI entry point over {table_steps} extra steps after the timed region"
            line["hip_path_ms_per_step"] = round(sum(s["ms_total"] for s in ks.values()) / table_steps, 2)
        if protocol is not None:
            line["at_reference_protocol"] = protocol
        if eval_line is not None:
            line["evaluate_loop"] = eval_line
        if ref_line is not None:
            line["at_reference_batch"] = ref_line
        if sweep is not None:
            line["batch_sweep"] = {"unit": "samples/s (whole job) by per-GPU batch", **sweep}
        if exchange is not None:
            line["exchange"] = exchange
        if world == 1 and not args.no_cpu_baseline and cfg["kind"] == "flamingo":
            line["cpu_baseline"] = cpu_baseline(model, cfg, lm_cfg, batch, args.cpu_samples, all_threads_too=args.cpu_all_threads)
    else:
        line = None
    if comm:
        dist.destroy_process_group()
    if line is not None:
        # RCCL writes its version banner to the C stdout of every rank (buffered: it used to land AFTER the JSON line when stdout is
        # a file).  Tear the communicator down first, flush the C streams, and only then print the one JSON line -- the last line.
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
