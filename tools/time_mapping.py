"""CustomMappingNetwork forward + backward at small batch: native launches (fenerf_mapping.hip) vs the PyTorch ops, both networks of a
DoubleImplicitGenerator3d (256 -> 256 x 4 -> 4096 and -> 1536).  python tools/time_mapping.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenerf_amd.siren import siren as S     # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = "cuda:0"
torch.manual_seed(0)
nets = [S.CustomMappingNetwork(256, 256, 4096).to(dev), S.CustomMappingNetwork(256, 256, 1536).to(dev)]
z = torch.randn(B, 256, device=dev)
ws = [torch.randn(B, n.network[-1].out_features, device=dev) for n in nets]


def step(native_route):
    for n, w in zip(nets, ws):
        for p in n.parameters():
            p.grad = None
        out = torch.cat(n(z), -1) if native_route else n.network(z)
        (out * w).sum().backward()


for name, route in (("native", True), ("torch ops", False), ("native", True), ("torch ops", False)):
    for _ in range(5):
        step(route)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step(route)
    torch.cuda.synchronize()
    print(f"{name:10s}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per forward + backward of both networks, batch {B}")
