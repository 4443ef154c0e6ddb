#!/usr/bin/env python
"""World 2 with the REAL kernels on ONE GPU (round 6; the round-5 review: "the combination native two-stage backward x world 2 exists
nowhere").  RCCL refuses two ranks on one device, gloo does not: two processes, both on cuda:0, backend gloo (all-reduce of CUDA tensors
staged through the host), each with the curriculum generator (H = 256 FiLM-SIREN, 32 x 96^3 grid, both mapping networks) and its OWN
latents.  Checked on every rank, parameter by parameter:

  (a) fenerf_amd.dist.GeneratorDataParallel + prepare_for_ddp (two-stage native backward, the grid's all-reduce started from its hook)
      leaves the MEAN of the two ranks' bare-module gradients in every .grad;
  (b) the same through DistributedDataParallel(**RECOMMENDED_DDP_KWARGS) with prepare_for_ddp;
  (c) two micro-batches under no_sync() / fdist.micro_batch_sync: the mean over ranks of the per-rank sums, through both wrappers;
  (d) both ranks hold identical gradients afterwards.

Launched by tests/test_gpu_parity.py::test_world_2_on_one_gpu_* as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/world2_one_gpu.py
Rank 0 prints one JSON line.  reference: train_double_latent_semantic.py:63 (gloo is the reference's own backend), :148-150, :402-446.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--img-size", type=int, default=128)
    ap.add_argument("--num-steps", type=int, default=24)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    from torch.nn.parallel import DistributedDataParallel as DDP
    from fenerf_amd import dist as fdist, procedural as proc
    import bench

    rank, _, world = fdist.init_from_env(backend="gloo")
    assert world == 2 and dist.get_backend() == "gloo"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
    sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
    gen, cur, curriculums = bench.curriculum_generator(spec, sd, dev, "f16x3")        # same seed on both ranks: identical weights
    B, S, N = args.batch, args.img_size, args.num_steps
    md = {**curriculums.extract_metadata(cur, 60000), "img_size": S, "num_steps": N, "nerf_noise": 0.5}
    torch.manual_seed(4242 + rank)                                                     # every rank its own latents and loss weights
    zs = [(torch.randn(B, cur["latent_geo_dim"], device=dev), torch.randn(B, cur["latent_app_dim"], device=dev)) for _ in range(2)]
    ws = [torch.randn((B, cur["output_dim"] - 1, S, S), device=dev) / (B * S * S) for _ in range(2)]
    params = {n: p for n, p in gen.named_parameters() if p.requires_grad}

    def backward(model, mb, seed):
        torch.manual_seed(seed + rank)                # the render's own draws (pose, jitter, noise): per rank, the same for every wrapper
        px, _ = model(zs[mb][0], zs[mb][1], **md)
        (px * ws[mb]).sum().backward()

    def grads():
        return {n: p.grad.detach().float().cpu().clone() for n, p in params.items() if p.grad is not None}

    def mean_over_ranks(g):
        out = {}
        for n in sorted(g):
            both = [torch.zeros_like(g[n]) for _ in range(world)]
            dist.all_gather(both, g[n])               # CPU tensors (gloo gathers host memory)
            out[n] = (both[0] + both[1]) / 2          # one fp32 rounding, as the collective's sum; / 2 is exact
        return out

    def rel(a, b):
        return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))

    def run(model, micro_batches):
        gen.zero_grad(set_to_none=True)
        for mb in range(micro_batches):
            with fdist.micro_batch_sync(model, mb, micro_batches):
                backward(model, mb, 100 + mb)
        torch.cuda.synchronize()
        return grads()

    res = {"rank": rank, "world": world, "backend": dist.get_backend(), "points_per_image": S * S * 2 * N}
    # the bare module: this rank's own gradients (one and two micro-batches), then the expected means
    fdist.prepare_for_ddp(gen, False)
    own1, own2 = run(gen, 1), run(gen, 2)
    want1, want2 = mean_over_ranks(own1), mean_over_ranks(own2)
    res["own_vs_mean"] = max(rel(own1[n], want1[n]) for n in own1)          # the ranks' gradients DO differ (own latents): far from 0
    assert set(own1) == set(params), sorted(set(params) - set(own1))

    def check(tag, got, want):
        assert set(got) == set(want), (tag, sorted(set(want) ^ set(got)))
        errs = {n: rel(got[n], want[n]) for n in want}
        worst = max(errs, key=errs.get)
        # identical across ranks (d): MAX and MIN of every element over the ranks agree
        same = True
        for n in sorted(got):
            hi, lo = got[n].clone(), got[n].clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            same = same and bool(torch.equal(hi, lo))
        res[tag] = {"worst_rel_err": errs[worst], "worst": worst, "tensors": len(errs), "identical_on_both_ranks": same}

    # (a) GeneratorDataParallel on the two-stage native backward
    fdist.prepare_for_ddp(gen, True)
    assert gen.siren.split_backward
    gdp = fdist.GeneratorDataParallel(gen, check_ranks=True)
    check("gdp_split", run(gdp, 1), want1)
    res["gdp_split"]["collectives"] = gdp.last_sync["collectives"]
    check("gdp_split_2_micro_batches", run(gdp, 2), want2)
    fdist.prepare_for_ddp(gen, False)
    check("gdp_single_node_backward", run(gdp, 1), want1)
    # (a') the exact-sparsity backward (siren.sparse_backward, generators/autograd.py) under the same wrapper: same means
    gen.siren.sparse_backward = True
    check("gdp_sparse_backward", run(gdp, 1), want1)
    check("gdp_sparse_backward_2_micro_batches", run(gdp, 2), want2)
    fdist.prepare_for_ddp(gen, True)                   # both switches on: the sparse node goes first
    check("gdp_split_and_sparse_backward", run(gdp, 1), want1)
    fdist.prepare_for_ddp(gen, False)
    gen.siren.sparse_backward = False
    gdp.detach_hooks()
    del gdp
    # (b) DistributedDataParallel with the recommended arguments on the two-stage backward, and the reference's own wrapper arguments
    kw = fdist.prepare_for_ddp(gen, True)
    ddp = DDP(gen, device_ids=[0], **kw)
    run(ddp, 1)                                        # first step: DDP rebuilds its buckets in arrival order afterwards
    check("ddp_recommended_split", run(ddp, 1), want1)
    check("ddp_recommended_split_2_micro_batches", run(ddp, 2), want2)
    del ddp
    fdist.prepare_for_ddp(gen, False)
    ddp = DDP(gen, device_ids=[0], find_unused_parameters=True)          # train_double_latent_semantic.py:148
    check("ddp_reference_wrapper", run(ddp, 1), want1)
    gen.siren.sparse_backward = True
    check("ddp_reference_wrapper_sparse_backward", run(ddp, 1), want1)
    gen.siren.sparse_backward = False
    from fenerf_amd.generators import autograd as GA
    GA.SparseHierarchicalRenderFunction.verify()
    del ddp
    res["peak_GB"] = torch.cuda.max_memory_allocated() / 2**30
    if rank == 0:
        print(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
