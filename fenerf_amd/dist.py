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

    Switches the generator's hierarchical render to its two-node backward (generators/autograd.py over fenerf_render_backward_stage 1 / 2): the
    gradient of the 96^3 feature grid -- 113 of the 124 MB a rank all-reduces per backward -- is handed to autograd when the last chain launch
    has been issued, with the weight-gradient kernels of the last two backward chunks (3.6 ms at 128 x 128 x 24) still to come, so the wrapper's
    all-reduce of it runs beside them instead of after everything.  Costs the dumps of those two chunks (+ 4.4 GB each at H = 256: 66.9 instead
    of 56.4 GB peak at the reference's 6-image micro-batch) -- hence opt-in.  Gradients: bit-identical to the one-call backward but for the
    atomically scattered grid's (1e-8).  Returns RECOMMENDED_DDP_KWARGS.  With the reference's own wrapper arguments the switch is harmless
    but buys nothing: DDP then keeps its static bucket order, in which the grid comes last.  fenerf_amd.dist.GeneratorDataParallel uses the
    same switch (and needs no bucket order)."""
    generator.siren.split_backward = bool(enable)
    return dict(RECOMMENDED_DDP_KWARGS)


class GeneratorDataParallel(torch.nn.Module):
    """Data-parallel gradient averaging for a generator WITHOUT DistributedDataParallel's per-parameter work (round 5).

        generator_dp = fdist.GeneratorDataParallel(generator)          # where the reference writes DDP(generator, ...), train...py:148
        imgs, _ = generator_dp(z_geo, z_app, **metadata); loss.backward()         # .grad of every parameter = the mean over ranks

    Why: the render's gradients come out of ONE native call (fenerf_render_backward) a few microseconds apart, 37 tensors of which 36 are
    smaller than 300 KB.  DistributedDataParallel handles each of them on its own -- a copy-and-divide launch into its bucket per parameter,
    the bucket bookkeeping on the host before and after -- : measured at world 1 on an MI355X, 109 of the 169 launches of a generator step
    and 1.5 of its 13.5 - 13.9 ms are the wrapper's (55 of 115 with RECOMMENDED_DDP_KWARGS; profiles/r05_ddp_step_timeline_*.txt), against
    60 launches / 12.0 ms for the bare module and 65 / 12.1 ms through this class.
    Here the gradients are reduced when the backward pass has finished: tensors of at least `async_numel` elements (16 MB: the 96^3 feature
    grid, 113 of the 124 MB) in place, started from the parameter's own post-accumulate hook so that with prepare_for_ddp's two-node backward
    the collective runs beside the weight-gradient kernels; everything else concatenated into one flat buffer (one launch), reduced with
    one collective, and handed back as views of it.  Averaging is the collective's own (RCCL `avg`; sum + one division on gloo).
    The same arithmetic as DDP up to the summation order inside the collective; `no_sync()` as DDP's (local accumulation over micro-batches).
    Every rank must produce gradients for the same parameters (DDP's find_unused_parameters=False contract): what is reduced is the set of
    parameters whose gradient hook fired since the last synchronisation (so a stale .grad kept by zero_grad(set_to_none=False) is not averaged
    again, and gradients accumulated under no_sync() are); `check_ranks=True` verifies with one extra 16-byte collective per step that all
    ranks reduce the same number of tensors and elements and raises instead of hanging in a mismatched all-reduce.  Parameters and buffers
    are broadcast from rank 0 at construction only (DDP re-broadcasts buffers every forward; the generators have none that change).  After a
    step the small parameters' .grad are views of one flat buffer, as with DDP's gradient_as_bucket_view=True.  State dict keys carry DDP's
    `module.` prefix."""

    def __init__(self, module, process_group=None, async_numel=1 << 22, broadcast=True, check_ranks=False):
        super().__init__()
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GeneratorDataParallel needs an initialised process group (fenerf_amd.dist.init_from_env)")
        self.module = module
        self.process_group = process_group
        self.async_numel = int(async_numel)
        self.require_backward_grad_sync = True
        self.world = dist.get_world_size(process_group)
        self._avg = dist.get_backend(process_group) == "nccl"          # gloo has no ReduceOp.AVG
        self._params = [p for p in module.parameters() if p.requires_grad]
        self._started, self._queued, self._fired = [], False, set()
        self.check_ranks = bool(check_ranks)
        self.last_sync = {"collectives": 0, "bytes": 0, "flat_tensors": 0}
        if broadcast and self.world > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self._params]

    def forward(self, *args, **kwargs):
        self._queued, self._started = False, []        # a backward pass that raised half way must not leave the next one unsynchronised
        return self.module(*args, **kwargs)

    def detach_hooks(self):
        """unregisters the gradient hooks from the wrapped module (which can then be used bare, or wrapped again)"""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def no_sync(self):
        """as DistributedDataParallel.no_sync: backward passes inside accumulate locally; the next one outside reduces the sum"""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old, self.require_backward_grad_sync = self.require_backward_grad_sync, False
            try:
                yield
            finally:
                self.require_backward_grad_sync = old
        return ctx()

    def _all_reduce(self, t, async_op):
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self.last_sync["collectives"] += 1
        self.last_sync["bytes"] += t.numel() * t.element_size()
        return dist.all_reduce(t, op=op, group=self.process_group, async_op=async_op)

    def _on_grad(self, p):
        self._fired.add(id(p))        # (also under no_sync(): the locally accumulated gradient is reduced by the next synchronised pass)
        if not self.require_backward_grad_sync:
            return
        if not self._queued:          # once per backward pass: reduce the rest when the whole graph has run
            self._queued = True
            self.last_sync = {"collectives": 0, "bytes": 0, "flat_tensors": 0}
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
        if p.numel() >= self.async_numel and p.grad is not None:
            self._started.append((p, self._all_reduce(p.grad, True)))

    def _finish(self):
        self._queued = False
        started, self._started = self._started, []
        early = {id(p) for p, _ in started}
        fired, self._fired = self._fired, set()
        small = {}
        for p in self._params:        # module order: the same on every rank
            if id(p) in fired and p.grad is not None and id(p) not in early:
                small.setdefault(p.grad.dtype, []).append(p)
        if self.check_ranks and self.world > 1:
            n_t = len(started) + sum(len(ps) for ps in small.values())
            n_e = sum(p.numel() for p, _ in started) + sum(p.numel() for ps in small.values() for p in ps)
            dev = next((p.grad.device for p in self._params if p.grad is not None), torch.device("cpu"))
            sig = torch.tensor([n_t, -n_t, n_e, -n_e], dtype=torch.float64, device=dev)
            dist.all_reduce(sig, op=dist.ReduceOp.MAX, group=self.process_group)
            if sig[0].item() != -sig[1].item() or sig[2].item() != -sig[3].item():
                for _, work in started:
                    work.wait()
                raise RuntimeError(f"GeneratorDataParallel: ranks disagree on the gradients of this step (this rank: {n_t} tensors, {n_e} elements; "
                                   f"max over ranks {int(sig[0].item())} / {int(sig[2].item())}, min {int(-sig[1].item())} / {int(-sig[3].item())}): "
                                   "every rank must produce gradients for the same parameters")
        flats = []
        for ps in small.values():                 # one flat buffer per dtype (the generator: fp32 only)
            flat = torch.cat([p.grad.reshape(-1) for p in ps])
            self._all_reduce(flat, False)
            self.last_sync["flat_tensors"] += len(ps)
            flats.append((ps, flat))
        for p, work in started:
            work.wait()
            if not self._avg:
                p.grad.div_(self.world)
        for ps, flat in flats:
            if not self._avg:
                flat.div_(self.world)
            off = 0
            for p in ps:
                n = p.numel()
                p.grad = flat[off:off + n].view_as(p)
                off += n
