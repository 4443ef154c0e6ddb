"""An inversion-shaped step (256 x 256 rays, 24 samples, no importance resampling, FiLM gradients only -- inverse_render_double_semantic.py:
225-247) on the bench's procedural density field (FiLM parameters of procedural.film_params, 11 % of the coarse samples carry density):
dense node against the exact-sparsity node.   python tools/exp/one_pass_step_timing.py [film|all]"""
import functools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from fenerf_amd import procedural as proc
from fenerf_amd.generators import generators as G, autograd as GA
from fenerf_amd.siren import siren as S_

film_only = (sys.argv[1] if len(sys.argv) > 1 else "film") == "film"
dev = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
mod = S_.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE(hidden_dim=256, z_geo_dim=256, z_app_dim=256, output_dim=22)
tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
mod.spatial_embeddings = torch.nn.Parameter(tsd["spatial_embeddings"].clone())
mod.load_state_dict(tsd, strict=False)
mod.precision = "f16x3"
gen = G.DoubleImplicitGenerator3d(functools.partial(S_.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=256), 256, 256, 22)
gen.siren = mod
gen = gen.to(dev); gen.device = dev; gen.siren.device = dev
for p in mod.parameters():
    p.requires_grad_(not film_only)
film = {k: torch.tensor(v, device=dev).requires_grad_(True) for k, v in proc.film_params(spec, 1, seed=5).items()}
kw = dict(img_size=256, fov=12, ray_start=0.88, ray_end=1.12, num_steps=24, h_stddev=0, v_stddev=0, h_mean=np.pi / 2, v_mean=np.pi / 2,
          hierarchical_sample=False, sample_dist=None, clamp_mode="relu", nerf_noise=0, fill_mode="eval_seg_padding_background")
w = torch.randn((1, 21, 256, 256), device=dev)

def step():
    for t in film.values(): t.grad = None
    for p in mod.parameters(): p.grad = None
    torch.manual_seed(3)                       # the render's draws (depth jitter): the same in every step, so that the two nodes can be compared
    px, _ = gen.forward_with_frequencies(film["freq_geo"], film["freq_app"], film["phase_geo"], film["phase_app"], **kw)
    (px * w).sum().backward()
    return px

for mode in (False, True):
    mod.sparse_backward = mode
    for _ in range(3): px = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    GA.SparseHierarchicalRenderFunction.verify()
    kept = GA.SparseHierarchicalRenderFunction.last_kept if mode else None
    g = film["freq_geo"].grad.clone()
    print(f"{'FiLM gradients only' if film_only else 'all gradients'}, sparse_backward = {mode!s:5}: {ms:6.2f} ms per step of 1 x 256 x 256 x 24 samples"
          + (f"; {int(kept[0])} of {kept[1]} samples kept ({100 * int(kept[0]) / kept[1]:.1f} %)" if kept else "") + f"; peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GB", flush=True)
    if mode:
        print(f"  pixels bit-identical: {bool(torch.equal(px, px0))}; d freq_geo relative difference {float((g - g0).abs().max() / g0.abs().max()):.1e}")
    else:
        px0, g0 = px.clone(), g
    torch.cuda.reset_peak_memory_stats()
