"""Why does one optimizer step of four micro-batches through the reference's DDP wrapper take 2 x four single steps on some runs?
Times the pattern with allocator statistics around it.  usage: python tools/exp/ddp_mb4_probe.py [n_warm_steps]"""
import os
import sys
import time

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                         # noqa: E402
from fenerf_amd import procedural as proc            # noqa: E402

n_warm = int(sys.argv[1]) if len(sys.argv) > 1 else 12
SYNC_EACH = (sys.argv[2] if len(sys.argv) > 2 else "sync") == "sync"       # synchronize after every micro-batch (per-micro-batch times) or only per optimizer step
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
gen, cur, curriculums = bench.curriculum_generator(spec, sd, dev, "f16x3")
md = {**curriculums.extract_metadata(cur, 60000), "img_size": 128, "num_steps": 24, "nerf_noise": 0.5}
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{bench._free_port()}", rank=0, world_size=1, device_id=dev)
zg, za = torch.randn(1, 256, device=dev), torch.randn(1, 256, device=dev)
w = torch.randn((1, 21, 128, 128), device=dev) / (128 * 128)
params = [p for p in gen.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=cur[50000]["gen_lr"], betas=tuple(float(v) for v in cur["betas"]), weight_decay=cur["weight_decay"])
ddp = DDP(gen, device_ids=[0], find_unused_parameters=True)


def loss():
    px, _ = ddp(zg, za, **md)
    return (px * w).sum()


def stats():
    s = torch.cuda.memory_stats()
    return {k: s[k] for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "reserved_bytes.all.current", "allocated_bytes.all.peak")}


for _ in range(n_warm):
    opt.zero_grad(set_to_none=True)
    loss().backward()
    opt.step()
torch.cuda.synchronize()
for tag, sync_once in (("reference pattern", False), ("one all-reduce", True), ("reference pattern again", False)):
    for rep in range(3):
        s0 = stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        per = []
        opt.zero_grad(set_to_none=True)
        for s_ in range(4):
            t1 = time.perf_counter()
            if sync_once and s_ < 3:
                with ddp.no_sync():
                    loss().backward()
            else:
                loss().backward()
            if SYNC_EACH:
                torch.cuda.synchronize()
            per.append((time.perf_counter() - t1) * 1e3)
        opt.step()
        torch.cuda.synchronize()
        s1 = stats()
        print(f"{tag} rep {rep}: {(time.perf_counter() - t0) * 1e3:.1f} ms; micro-batches {[round(p, 1) for p in per]}; "
              f"device allocs +{s1['num_device_alloc'] - s0['num_device_alloc']} frees +{s1['num_device_free'] - s0['num_device_free']} "
              f"retries +{s1['num_alloc_retries'] - s0['num_alloc_retries']} reserved {s1['reserved_bytes.all.current'] / 2**30:.1f} GB", flush=True)
# bench.py's own bracket: one warm optimizer step, then TWO optimizer steps with no synchronisation in between
def one(sync_once):
    opt.zero_grad(set_to_none=True)
    for s_ in range(4):
        if sync_once and s_ < 3:
            with ddp.no_sync():
                loss().backward()
        else:
            loss().backward()
    opt.step()


for tag, sync_once in (("reference pattern, 2 steps back to back", False), ("one all-reduce, 2 steps back to back", True)):
    one(sync_once)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one(sync_once); one(sync_once)
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / 2 * 1e3:.1f} ms per optimizer step", flush=True)
dist.destroy_process_group()
