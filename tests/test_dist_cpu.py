"""world_size-2 gloo test (CPU) of the N>1 path: image sharding, max-over-ranks timing agreement, gather."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fenerf_amd import dist as fdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = fdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    total = 7
    ids = fdist.rank_strided(total, rank, world)
    b, e = fdist.contiguous_shard(total, rank, world)
    # every rank "renders" its own images (stand-in tensor whose value encodes the image id), no data-path collective
    imgs = torch.stack([torch.full((3, 2, 2), float(i)) for i in range(b, e)]) if e > b else torch.zeros((0, 3, 2, 2))
    slow = 0.25 if rank == 1 else 0.1
    agreed = fdist.max_over_ranks(slow)
    allimg = fdist.gather_images(imgs, dst=0)
    q.put((rank, ids, (b, e), agreed, None if allimg is None else allimg[:, 0, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ids0, sh0, t0, all0), (r1, ids1, sh1, t1, all1) = res
    assert sorted(ids0 + ids1) == list(range(7)) and not set(ids0) & set(ids1)
    assert sh0 == (0, 4) and sh1 == (4, 7)
    assert t0 == t1 == 0.25
    assert all0 == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0] and all1 is None


def test_single_process_passthrough():
    assert fdist.max_over_ranks(1.5) == 1.5
    assert fdist.contiguous_shard(10, 0, 1) == (0, 10)
    x = torch.zeros(2, 3)
    assert fdist.gather_images(x) is x


def test_bench_self_launch_rendezvous_world_size_2():
    """`python bench.py --gpus 2` with no launcher environment re-executes itself under torch.distributed.run (the driver's N > 1
    command is that launcher line itself); --dist-check stops after the rendezvous + all-reduce of ones, backend gloo on the CPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-check", "--dist-backend", "gloo"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["dist_check"] and d["world_size"] == 2 and d["n_ranks_seen"] == 2 and d["backend"] == "gloo"
    # the N > 1 line's generator-step leg (bench.ddp_timed_leg: DistributedDataParallel wrapper, barrier / max-over-ranks bracket, rank
    # census, byte count) on a stand-in module over gloo: the schema the GPU line carries, and DDP really synchronised the two ranks
    sys.path.insert(0, root)
    import bench
    leg = d["gstep_ddp"]
    assert tuple(leg) == bench.GSTEP_DDP_KEYS
    assert leg["n_ranks"] == leg["n_ranks_seen"] == 2 and leg["dist_backend"] == "gloo" and leg["allreduce_per_micro_batch"] is True
    assert leg["allreduce_bytes"] == 4 * (16 * 32 + 32 + 32 * 4 + 4) and leg["allreduce_bytes_largest_tensor"] == 4 * 16 * 32
    assert leg["ms"] > 0 and leg["ms_no_ddp"] > 0 and abs(leg["allreduce_ms_exposed"] - (leg["ms"] - leg["ms_no_ddp"])) < 1e-9
    # round 5: one optimizer step of four micro-batches, the reference's pattern (an all-reduce per micro-batch) beside micro_batch_sync (one)
    assert leg["ms_optimizer_step_4_micro_batches"] > 0 and leg["ms_optimizer_step_4_micro_batches_one_allreduce"] > 0
    # round 5: fenerf_amd.dist.GeneratorDataParallel beside the two DDP configurations; the stand-in's 4 small tensors leave as ONE collective
    assert leg["ms_generator_data_parallel"] > 0 and leg["collectives_per_step_generator_data_parallel"] == 1
    assert d["params_identical_across_ranks"] is True


def test_self_launch_command_is_the_drivers_launcher_line():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cmd = bench.self_launch_command(4, ["--gpus", "4", "--steps", "5"], port=12345)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "5"] and cmd[-5].endswith("bench.py")


def _mb_worker(rank, world, port, q):
    """batch_split micro-batches through DDP: the reference's pattern (all-reduce every micro-batch) vs fdist.micro_batch_sync (local
    accumulation, one all-reduce in the last micro-batch's backward)"""
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    fdist.init_from_env(backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    ddp = DDP(net, find_unused_parameters=True)
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(10 + rank))        # every rank its own data
    n_splits, res = 4, {}
    for mode in ("reference", "micro_batch_sync"):
        ddp.zero_grad(set_to_none=True)
        for split in range(n_splits):
            ctx = fdist.micro_batch_sync(ddp, split, n_splits) if mode == "micro_batch_sync" else fdist.micro_batch_sync(net, split, n_splits)
            with ctx:
                ddp(x[2 * split:2 * split + 2]).square().sum().backward()
        res[mode] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
    q.put((rank, res["reference"].tolist(), res["micro_batch_sync"].tolist()))      # plain lists: tensors would travel as shared-memory handles
    dist.barrier()
    dist.destroy_process_group()


def test_micro_batch_sync_equals_the_reference_pattern_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mb_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, ref0, mb0), (_, ref1, mb1) = res
    assert ref0 == ref1 and mb0 == mb1, "DDP leaves identical gradients on both ranks"
    ref0, mb0 = torch.tensor(ref0), torch.tensor(mb0)
    assert float(ref0.abs().max()) > 0 and torch.allclose(mb0, ref0, rtol=1e-5, atol=1e-6)


def _gdp_worker(rank, world, port, q):
    """fdist.GeneratorDataParallel beside DistributedDataParallel on the same module and data: plain step, the large-tensor path
    (collective started from the parameter's hook), and local accumulation under no_sync()"""
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    fdist.init_from_env(backend="gloo")
    torch.manual_seed(100 + rank)                     # DIFFERENT initial weights per rank: the wrapper must broadcast rank 0's
    net = torch.nn.Sequential(torch.nn.Linear(6, 64), torch.nn.Tanh(), torch.nn.Linear(64, 3))
    gdp = fdist.GeneratorDataParallel(net, async_numel=6 * 64, check_ranks=True)          # the first weight (384 elements) takes the early, in-place path
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    ddp = DDP(copy.deepcopy(net), find_unused_parameters=True)       # its own copy of the (broadcast) module: no hooks shared between the two
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(10 + rank))
    res = {}
    for name, wrapper in (("ddp", ddp), ("gdp", gdp)):
        flat = lambda: torch.cat([p.grad.reshape(-1) for p in wrapper.module.parameters()]).clone()
        wrapper.zero_grad(set_to_none=True)
        wrapper(x).square().sum().backward()
        res[name] = flat()
        wrapper.zero_grad(set_to_none=True)
        for split in range(4):
            with fdist.micro_batch_sync(wrapper, split, 4):
                wrapper(x[2 * split:2 * split + 2]).square().sum().backward()
        res[name + "_mb"] = flat()
    stats = dict(gdp.last_sync)
    # a second backward on gradients that are views of the previous flat buffer (zero_grad(set_to_none=False)): accumulate in place, reduce again
    net.zero_grad(set_to_none=False)
    gdp(x).square().sum().backward()
    res["gdp_again"] = flat()
    # round 6: only gradients produced since the last synchronisation are reduced.  A parameter skipped in this pass keeps its
    # stale .grad (zero_grad(set_to_none=False) semantics) untouched -- rank-dependent values are NOT averaged a second time
    net.zero_grad(set_to_none=False)
    net[2].bias.grad.fill_(float(rank + 1))
    net[2].bias.requires_grad_(False)
    gdp(x).square().sum().backward()
    net[2].bias.requires_grad_(True)
    res["stale_bias"] = net[2].bias.grad.clone()
    skipped_stats = dict(gdp.last_sync)
    # ranks that disagree on which parameters took part: check_ranks raises on every rank instead of hanging in a mismatched collective
    net.zero_grad(set_to_none=True)
    if rank == 1:
        net[2].bias.requires_grad_(False)
    try:
        gdp(x).square().sum().backward()
        res["mismatch"] = torch.zeros(1)
    except RuntimeError as e:
        res["mismatch"] = torch.ones(1) if "ranks disagree" in str(e) else torch.full((1,), 2.0)
    net[2].bias.requires_grad_(True)
    stats["skipped_flat_tensors"] = skipped_stats["flat_tensors"]
    q.put((rank, w0.tolist(), {k: v.tolist() for k, v in res.items()}, stats))
    dist.barrier()
    dist.destroy_process_group()


def test_generator_data_parallel_equals_ddp_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gdp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, r0, st0), (_, w1, r1, st1) = res
    assert w0 == w1, "parameters broadcast from rank 0 at construction"
    assert r0.pop("stale_bias") == [1.0] * 3 and r1.pop("stale_bias") == [2.0] * 3, "a gradient no backward produced is not reduced again"
    assert r0.pop("mismatch") == [1.0] and r1.pop("mismatch") == [1.0], "check_ranks: both ranks raise on a parameter-set mismatch"
    assert st0.pop("skipped_flat_tensors") == st1.pop("skipped_flat_tensors") == 2
    for k in r0:
        assert r0[k] == r1[k], f"{k}: identical gradients on both ranks"
    # two ranks: the mean of two fp32 numbers has one possible rounding, whatever the collective's order
    np.testing.assert_allclose(r0["gdp"], r0["ddp"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(r0["gdp_mb"], r0["ddp_mb"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(r0["gdp_again"], r0["gdp"], rtol=0, atol=1e-7)
    # 4 parameters: one reduced in place from its hook, three through the flat buffer = 2 collectives (DDP: per-parameter copies + buckets)
    assert st0 == st1 == {"collectives": 2, "bytes": 4 * (6 * 64 + 64 + 64 * 3 + 3), "flat_tensors": 3}


def test_generator_data_parallel_needs_a_process_group():
    with pytest.raises(RuntimeError, match="process group"):
        fdist.GeneratorDataParallel(torch.nn.Linear(2, 2))
