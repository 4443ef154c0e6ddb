"""A 40 - 70 ms stall shows up in one backward out of ~30 when the generator steps through DistributedDataParallel inside bench.py.
Is it Python's garbage collector?  Runs the DDP legs' pattern with gc.callbacks timing every collection.
usage: python tools/exp/ddp_stall_probe.py [gc|nogc]"""
import gc
import os
import sys
import time

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                         # noqa: E402
from fenerf_amd import procedural as proc            # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "gc"
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
opt = torch.optim.Adam(params, lr=1e-5, betas=(0.0, 0.9))
ddp = DDP(gen, device_ids=[0], find_unused_parameters=True)

events = []
t_gc = [0.0]


def on_gc(phase, info):
    if phase == "start":
        t_gc[0] = time.perf_counter()
    else:
        events.append((info["generation"], (time.perf_counter() - t_gc[0]) * 1e3, info.get("collected", 0)))


gc.callbacks.append(on_gc)
if mode == "nogc":
    gc.collect()
    gc.disable()
slow = []
for it in range(120):
    if it % 4 == 0:
        opt.zero_grad(set_to_none=True)
    n0 = len(events)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    px, _ = ddp(zg, za, **md)
    loss = (px * w).sum()
    loss.item()
    loss.backward()
    if it % 4 == 3:
        opt.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    if ms > 20 and it > 8:
        slow.append((it, round(ms, 1), [(g, round(d, 1), c) for g, d, c in events[n0:]]))
print(f"mode {mode}: {len(slow)} slow steps of 120: {slow}")
print(f"collections: gen0 {sum(1 for e in events if e[0] == 0)}, gen1 {sum(1 for e in events if e[0] == 1)}, gen2 {sum(1 for e in events if e[0] == 2)}; "
      f"longest {max((e[1] for e in events), default=0):.1f} ms")
dist.destroy_process_group()
