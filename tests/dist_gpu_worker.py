"""Worker of tests/test_rccl_gpu.py::test_two_ranks_on_one_gpu_*: one of N ranks of the data-parallel engine with DEVICE tensors.
Launched by torch.distributed.run; MMGL_DIST_BACKEND = nccl (one GPU per rank) or gloo (all ranks share GPU 0: the N-rank path
-- hooks, async bucket all-reduces, fused AdamW on the flat buffers -- on the hardware a 1-GPU box has).

Checks, on every rank:
  1. after the exchange, the flat gradient equals the sum of the per-rank gradients recomputed locally without any exchange;
  2. the bucket all-reduces were issued in the same order on every rank (the engine's own check, forced on every step);
  3. after two optimizer steps every rank holds bit-identical parameters.
Prints one JSON line from rank 0."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("MMGL_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, **(dict(device_id=dev) if backend == "nccl" else {}))
    from helpers import Fixture, load_exact, mpt_args, tiny_clip_vision_config, tiny_opt_config, tiny_roberta_config
    from mmgl_amd.distributed import DataParallelEngine
    from mmgl_amd.model import CrossAttentionModel

    fx = Fixture("g1_wrapper_all.npz")
    torch.manual_seed(100 + rank)                      # deliberately different initial weights: the constructor broadcast must fix that
    model = CrossAttentionModel(mpt_args(context="all"), tokenizer=None, lm_config=tiny_opt_config(dropout=0.0), text_config=tiny_roberta_config(),
                                visual_config=tiny_clip_vision_config())
    if rank == 0:
        load_exact(model, fx.p)
    model = model.to(dev).train()
    # small buckets: several all-reduces per step, so that the launch-order check has something to compare
    # world == 1: the exchange is FORCED ON (a world_size-1 RCCL group on the one GPU of the box): hooks, async all-reduces on RCCL's
    # stream, work.wait(), the dynamic GEMM tile schedule -- everything but the wire
    engine = DataParallelEngine(model, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.01, bucket_mb=0.02, tail_mb=0.005, force_exchange=True)
    assert len(engine.buckets) >= 3, len(engine.buckets)
    assert engine.exchange
    from mmgl_amd import _lib
    assert _lib.lib().mmgl_gemm_get_tile_counter(), "the engine must switch the persistent GEMM to the dynamic tile schedule"

    def batch_of(r, step):
        g = torch.Generator().manual_seed(1000 * step + r)
        b = {k: v.clone() for k, v in fx.inp.items()}
        ids = b["input_ids"]
        b["input_ids"] = torch.where(b["attention_mask"].bool(), torch.randint(3, 128, ids.shape, generator=g), ids)
        b["labels"] = b["input_ids"].clone()
        return {k: v.to(dev) for k, v in b.items()}

    report = dict(world=world, backend=backend, buckets=len(engine.buckets), steps=[], launch_order=None)
    for step in range(2):
        # (a) the exchanged gradient
        engine.sync = True
        model(**batch_of(rank, step)).loss.backward()
        engine.finish_backward()
        assert engine.last_launch_order == list(range(len(engine.buckets))), engine.last_launch_order
        report["launch_order"] = list(engine.last_launch_order)
        got = engine.flat_grad.clone()
        # (b) the same sum without any exchange: every rank's batch, accumulated locally
        engine.zero_grad()
        engine.sync = False
        for r in range(world):
            model(**batch_of(r, step)).loss.backward()
            engine.finish_backward()
        want = engine.flat_grad.clone()
        err = ((got - want).abs().max() / want.abs().max().clamp_min(1e-12)).item()
        assert err < 2e-5, f"rank {rank} step {step}: exchanged gradient differs from the local sum: {err}"
        if world == 1:                                 # one rank: the all-reduce is the identity, so the two runs must agree to the bit
            assert torch.equal(got, want), "world_size 1: the exchanged gradient differs from the one computed without the exchange"
        engine.flat_grad.copy_(got)
        engine.sync = True
        engine.step()
        engine.zero_grad()
        report["steps"].append(dict(grad_rel_err=err))
    mine = engine.flat_param.detach().float()
    mine = mine.cpu() if backend == "gloo" else mine
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    both = [t.cpu() for t in both]
    assert all(torch.equal(both[0], t) for t in both[1:]), "parameters diverged across ranks"
    report["params_equal"] = True
    report["exchange_bytes"] = engine.exchange_bytes
    if rank == 0:
        print(json.dumps(report), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
