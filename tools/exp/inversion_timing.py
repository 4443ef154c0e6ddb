"""Time per iteration of callers.inverse_render at the reference's inversion shape (256 x 256, 24 coarse samples, no importance resampling,
FiLM gradients only; inverse_render_double_semantic.py:225-247) with the dense and the exact-sparsity backward.
    python tools/exp/inversion_timing.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fenerf_amd import callers, procedural as proc
from fenerf_amd.generators import autograd as GA

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
gen, cur, _ = bench.curriculum_generator(spec, sd, dev, "f16x3")
gen.eval()
for p in gen.parameters():
    p.requires_grad_(False)
opts = callers.inversion_options(image_size=256, device=dev)
torch.manual_seed(0)
gt_img, gt_seg = torch.rand(1, 3, 256, 256, device=dev) * 2 - 1, torch.rand(1, 18, 256, 256, device=dev) * 2 - 1
res = {}
for mode in (False, True, "auto"):
    gen.siren.sparse_backward = mode
    torch.manual_seed(1)
    callers.inverse_render(gen, gt_img, gt_seg, opts, n_iterations=5, n_mean_latents=1000)          # warm-up
    torch.cuda.synchronize()
    torch.manual_seed(1)
    t0 = time.perf_counter()
    r = callers.inverse_render(gen, gt_img, gt_seg, opts, n_iterations=n, n_mean_latents=1000)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    GA.SparseHierarchicalRenderFunction.verify()
    kept = GA.SparseHierarchicalRenderFunction.last_kept if mode else None
    res[mode] = r["losses"]
    print(f"sparse_backward = {mode!s:5}: {ms:6.2f} ms per iteration ({n} iterations incl. the mean-latent pass), final loss {r['losses'][-1]:.6f}"
          + (f", {int(kept[0])} of {kept[1]} samples kept in the last backward" if kept else ""), flush=True)
d = max(abs(a - b) for a, b in zip(res[False], res[True]))
print(f"largest loss difference dense vs sparse over the trajectory: {d:.2e}")
