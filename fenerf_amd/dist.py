"""Multi-GPU sharding of the render path: one process per GPU, shard by image, NO data-path collective
(rays/images are independent; the model -- 2.75 MB of weights + 113 MB grid -- is replicated).
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm, "gloo" in CPU tests) is used only to agree on timing
and, optionally, to gather finished images on one rank.

reference analogue: the rank-strided image loop of fid_evaluation.py:136-150 and DistributedSampler sharding
(datasets.py:96-113); the reference itself initialises gloo (train_double_latent_semantic.py:63).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None, force=False):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher (torch.distributed.run).  Returns (rank, local_rank, world).
    A process group is created for world > 1, or for a single rank too with `force` (a one-GPU box can then exercise the RCCL
    initialisation, barrier and reductions of the N > 1 path)."""
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def rank_strided(total, rank, world):
    """Image ids rank, rank+world, ... (< total) -- the reference's FID-dump split (fid_evaluation.py:136-150)."""
    return list(range(rank, total, world))


def contiguous_shard(total, rank, world):
    """[begin, end) of a balanced contiguous split (first `total % world` ranks get one extra)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def max_over_ranks(value, device="cpu"):
    """Wall-clock agreement for benchmarks: MAX of a python float over all ranks."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_images(local, dst=0):
    """Collects per-rank image batches [b_i, C, S, S] (b_i may differ) on rank `dst`, in rank order.  Returns the
    concatenated tensor on dst, None elsewhere.  Off the hot path (inference convenience)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    pad = max(counts)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[:c] for o, c in zip(out, counts)], 0)


def micro_batch_sync(ddp_module, split, n_splits):
    """Context manager for the `batch_split` loop of the reference's generator / discriminator steps
    (train_double_latent_semantic.py:296-338, 407-446):

        for split in range(batch_split):
            with fdist.micro_batch_sync(generator_ddp, split, batch_split):
                loss = ...generator_ddp(z[split])...
                scaler.scale(loss).backward()

    The reference calls `generator_ddp(...)` / `.backward()` once per micro-batch with no `no_sync()`, so DistributedDataParallel
    all-reduces ALL generator gradients -- 124 MB, 113 MB of it the 96^3 feature grid -- `batch_split` (= 4) times per optimizer step and the
    optimizer only ever sees their sum.  Summation commutes with the all-reduce: accumulating locally and reducing in the LAST micro-batch's
    backward gives the same gradients up to fp32 summation order with a quarter of the xGMI traffic (7 links x ~153 GB/s per GPU,
    point-to-point: a ring all-reduce of 124 MB is per-link bound) and a quarter of the exposed all-reduce time.  Opt-in (it is a change
    of the reference's call pattern, not of its arithmetic); bench.py reports the reference's pattern (`allreduce_per_micro_batch: true`).
    Not a DDP module (world 1 without a wrapper): a no-op."""
    import contextlib
    if split < n_splits - 1 and hasattr(ddp_module, "no_sync"):
        return ddp_module.no_sync()
    return contextlib.nullcontext()


# DistributedDataParallel arguments this package recommends over the reference's DDP(generator, find_unused_parameters=True)
# (train_double_latent_semantic.py:148): every generator parameter takes part in every step, so the autograd-graph traversal is not
# needed -- and without it DDP rebuilds its buckets after the first backward in the order the gradients actually arrive, which is what lets
# the grid's bucket go first (prepare_for_ddp); gradients are views into the buckets (no copy of 124 MB in and out per backward); buckets
# large enough that the 113-MB grid and everything else leave as two collectives (ring all-reduce over point-to-point xGMI links is
# per-link bound: fewer, larger messages).
RECOMMENDED_DDP_KWARGS = dict(find_unused_parameters=False, gradient_as_bucket_view=True, bucket_cap_mb=128)


def prepare_for_ddp(generator, enable=True):
    """Call before wrapping a generator in DistributedDataParallel(generator, device_ids=[rank], **RECOMMENDED_DDP_KWARGS).

    Switches the generator's hierarchical render to its two-node backward (generators/autograd.py): the gradient of the 96^3 feature grid
    -- 113 of the 124 MB DDP all-reduces per backward -- is handed to autograd as soon as the last chain launch has finished, so DDP
    starts its all-reduce while the weight-gradient kernels (a quarter of the step) are still running, instead of after everything.
    Costs memory (the d(theta) dumps of all backward chunks are alive together: + 4.4 GB per 128 x 128 x 24 pass) -- hence opt-in.
    Returns RECOMMENDED_DDP_KWARGS.  With the reference's own wrapper arguments the switch is harmless but buys nothing: DDP then keeps
    its static bucket order, in which the grid comes last."""
    generator.siren.split_backward = bool(enable)
    return dict(RECOMMENDED_DDP_KWARGS)
