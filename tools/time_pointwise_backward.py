"""Step time and peak memory of the per-point-modulated SIREN under autograd (SPATIALSIRENGRID): the native route
(siren.autograd.PointwiseSirenFunction, round 6) beside the PyTorch-ROCm route of rounds 3-5, same module, same inputs.
    python tools/time_pointwise_backward.py [points] [H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fenerf_amd import native
from fenerf_amd.siren import siren as S

P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
mod = S.SPATIALSIRENGRID(input_dim=3, z_dim=256, hidden_dim=H, output_dim=4).to(dev).train()
mod.device = dev
pts = (torch.rand((1, P, 3), device=dev) * 2 - 1)
dirs = torch.nn.functional.normalize(torch.randn((1, P, 3), device=dev), dim=-1)
z = torch.randn((1, 256), device=dev)
w = torch.randn((1, P, 4), device=dev)
for route in ("native", "torch"):
    mod.NATIVE_POINTWISE_BACKWARD = route == "native"
    def step():
        mod.zero_grad(set_to_none=True)
        out = mod(pts, z, dirs)                      # latent grid -> per-point latents -> per-point mapping network -> FiLM-SIREN
        (out * w).sum().backward()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    with native.phase_timing() as t:
        for _ in range(n):
            step()
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    lib = {k: round(v / n, 3) for k, v in t.ms.items() if v > 0}
    print(f"{route}: {ms:.2f} ms per forward + backward of {P} points (H = {H}), peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GB; "
          f"library kernels per step (ms): {lib}")
