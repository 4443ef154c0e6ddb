"""Soak of the exact-sparsity generator step: N steps at 1 x 128 x 128 x 24+24 with fresh latents-like FiLM parameters and draws every step
(different kept sets, different buffer lengths); the deferred overflow check after every step, device / host memory and step time at the
start and at the end.   python tools/exp/sparse_soak.py [steps]"""
import functools, os, resource, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from fenerf_amd import procedural as proc
from fenerf_amd.generators import generators as G, autograd as GA
from fenerf_amd.siren import siren as S_

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
mod = S_.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE(hidden_dim=256, z_geo_dim=256, z_app_dim=256, output_dim=22)
tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
mod.spatial_embeddings = torch.nn.Parameter(tsd["spatial_embeddings"].clone())
mod.load_state_dict(tsd, strict=False)
mod.precision = "f16x3"
mod.sparse_backward = True
gen = G.DoubleImplicitGenerator3d(functools.partial(S_.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=256), 256, 256, 22)
gen.siren = mod
gen = gen.to(dev); gen.device = dev; gen.siren.device = dev
films = [{k: torch.tensor(v, device=dev).requires_grad_(True) for k, v in proc.film_params(spec, 1, seed=s).items()} for s in range(16)]
kw = dict(img_size=128, fov=12, ray_start=0.88, ray_end=1.12, num_steps=24, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2, v_mean=np.pi / 2,
          hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.2, last_back=False)
w = torch.randn((1, 21, 128, 128), device=dev)
opt = torch.optim.SGD([p for n_, p in mod.named_parameters() if "mapping_network" not in n_], lr=1e-9)
times, kept, caps = [], [], set()
for i in range(n):
    film = films[i % len(films)]
    opt.zero_grad(set_to_none=True)
    t0 = time.perf_counter()
    px, _ = gen.forward_with_frequencies(film["freq_geo"], film["freq_app"], film["phase_geo"], film["phase_app"], **kw)
    (px * w).sum().backward()
    opt.step()
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
    GA.SparseHierarchicalRenderFunction.verify()
    kept.append(int(GA.SparseHierarchicalRenderFunction.last_kept[0]))
    caps.add(GA.SparseHierarchicalRenderFunction.last_groups[0][1])
    if i in (20, n - 1):
        print(f"step {i}: median of the last 20 steps {sorted(times[-20:])[10]:.2f} ms, device allocated {torch.cuda.memory_allocated() / 2**30:.2f} GB "
              f"(reserved {torch.cuda.memory_reserved() / 2**30:.2f}), host max RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:.2f} GB", flush=True)
print(f"{n} steps: kept samples {min(kept)} .. {max(kept)} of 786432, {len(caps)} different buffer lengths, no overflow flag, finite pixels: {bool(torch.isfinite(px).all())}")
