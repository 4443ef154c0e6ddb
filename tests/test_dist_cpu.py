"""world_size-2 gloo test (CPU) of the N>1 path: image sharding, max-over-ranks timing agreement, gather."""
import os
import socket

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
