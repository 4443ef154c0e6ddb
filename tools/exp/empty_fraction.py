"""How many samples of the bench's render take (and give) exactly zero gradient?  Under the relu clamp a sample with sigma <= 0 has
alpha = 0: its colour / label rows are multiplied by the weight 0 and relu' = 0 stops the density gradient (volumetric_rendering.py:
36-47) -- d_out of that point is an all-zero row, its whole backward contributes zeros.  Counted on the bench model (procedural weights,
sigma_gain 2000), coarse pass of 128 x 128 x 24, per point and per 16-point wave tile / 128-point workgroup tile of the chain kernel.
    python tools/exp/empty_fraction.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fenerf_amd import native, procedural as proc
from fenerf_amd.generators import volumetric_rendering as VR

DEV = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
for gain in (2000.0, 30.0):
    sd = proc.make_state_dict(spec, seed=0, sigma_gain=gain, with_mapping=False)
    nat = native.NativeModel(sd, spec, DEV, "f16x3")
    film = {k: torch.tensor(v, device=DEV) for k, v in proc.film_params(spec, 1, seed=1).items()}
    torch.manual_seed(0)
    o, d, z, _, _ = VR.sample_rays(1, 24, DEV, 12, (128, 128), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    out = nat.siren_forward_rays(o, d, z.reshape(1, 128 * 128, 24), film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])
    empty = (out[..., -1] <= 0).reshape(-1)
    n = empty.numel()
    print(f"sigma_gain {gain:g}: {n} coarse points, sigma <= 0 on {float(empty.float().mean()):.3f}; all-empty 16-point wave tiles "
          f"{float(empty.reshape(-1, 16).all(1).float().mean()):.3f}, all-empty 128-point tile groups {float(empty.reshape(-1, 128).all(1).float().mean()):.4f}; "
          f"longest run of samples with the same sign along a ray, median {int(np.median([max(len(s) for s in ''.join('1' if v else '0' for v in row).replace('01', '0 1').replace('10', '1 0').split()) for row in empty.reshape(-1, 24)[:2048].cpu().numpy()]))}")
