"""How much the host side adds around the fused render: generator.forward_with_frequencies (draws, ray setup, epilogue) vs the
bare fenerf_render_forward call on the same rays.  python tools/time_call_overhead.py"""
import functools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fenerf_amd import _lib, procedural as proc
from fenerf_amd.generators import generators as G, volumetric_rendering as VR
from fenerf_amd.siren import siren as S

DEV = "cuda:0"
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96, z_dim=8)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
gen = G.DoubleImplicitGenerator3d(functools.partial(S.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=256), 8, 8, 22)
tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
gen.siren.spatial_embeddings = torch.nn.Parameter(tsd["spatial_embeddings"].clone())
gen.siren.load_state_dict(tsd, strict=False)
gen = gen.to(DEV).eval()
gen.device = torch.device(DEV); gen.siren.device = gen.device
B, S_, N = 1, 128, 24
film = {k: torch.tensor(v, device=DEV) for k, v in proc.film_params(spec, B, seed=1).items()}
kw = dict(img_size=S_, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2, v_mean=np.pi / 2,
          hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.0)
nat = gen.siren.native(DEV)
o, d, z, _, _ = VR.sample_rays(B, N, gen.device, 12, (S_, S_), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
u = torch.rand((B * S_ * S_, N), device=DEV)
opts = _lib.composite_opts("relu", 0.0)
tf = (film["freq_geo"], film["phase_geo"], film["freq_app"], film["phase_app"])

def timed(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3

with torch.no_grad():
    a = timed(lambda: nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True))
    b = timed(lambda: gen.forward_with_frequencies(film["freq_geo"], film["freq_app"], film["phase_geo"], film["phase_app"], **kw))
    zz = torch.randn(B, 8, device=DEV)
    c = timed(lambda: gen(zz, zz, **kw))
print(f"bare fused render {a:.3f} ms | generator.forward_with_frequencies {b:.3f} ms (+{b - a:.3f}) | generator.forward(z) {c:.3f} ms (+{c - a:.3f})")
