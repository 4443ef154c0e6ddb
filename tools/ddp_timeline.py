"""A few generator steps through DistributedDataParallel at world 1 (one-rank RCCL group) for a kernel-trace timeline: where does the
all-reduce of the 124 MB of gradients sit relative to the weight-gradient kernels?
    rocprofv3 --kernel-trace ... -- python tools/ddp_timeline.py [reference|tuned|bare|gdp]      then tools/gstep_timeline.py <dir>
(bare: the same steps on the module itself, no wrapper: bench.py's gstep_z; gdp: fenerf_amd.dist.GeneratorDataParallel)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                         # noqa: E402
from fenerf_amd import dist as fdist, procedural as proc     # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "tuned"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
gen, cur, curriculums = bench.curriculum_generator(spec, sd, dev, "f16x3")
md = {**curriculums.extract_metadata(cur, 60000), "img_size": 128, "num_steps": 24, "nerf_noise": 0.5}
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{bench._free_port()}", rank=0, world_size=1, device_id=dev)
kw = fdist.prepare_for_ddp(gen) if mode in ("tuned", "gdp") else dict(find_unused_parameters=True)
ddp = gen if mode == "bare" else fdist.GeneratorDataParallel(gen) if mode == "gdp" else DDP(gen, device_ids=[0], **kw)
zg, za = torch.randn(1, 256, device=dev), torch.randn(1, 256, device=dev)
w = torch.randn((1, 21, 128, 128), device=dev) / (128 * 128)
for _ in range(int(os.environ.get("STEPS", "6"))):
    for p in gen.parameters():
        p.grad = None
    px, _ = ddp(zg, za, **md)
    (px * w).sum().backward()
torch.cuda.synchronize()
dist.destroy_process_group()
