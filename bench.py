#!/usr/bin/env python
"""Benchmark of the forward-render hot path (BASELINE.json metric: rays/s per GPU).

A step = one fenerf_render_forward pass (coarse SIREN -> composite -> resample -> fine SIREN -> merge ->
composite) over one batch of synthetic rays already resident in HBM.  Workload = BASELINE.json configs[1]:
TextureEmbeddingPiGAN256SEMANTICDISENTANGLE_DIM_96 (H=256 FiLM-SIREN + 32x96^3 grid, 22 channels), 128x128 rays,
24+24 hierarchical samples, batch 1 per GPU, procedural (random-init-range) weights, synthetic latents.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU, weak scaling)

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (siren_kernel, MFMA-bound): algorithmic
FLOPs per launch (1,603,584 FLOP/point x points) / its hipEvent-timed average duration, against the
fp32-matrix peak.  `cpu_baseline` times the numpy oracle (a port of the reference CPU path) on a bounded sample.
`gstep` (N=1 only, outside the timed region) is the other half of BASELINE.json's metric: one generator step (forward +
backward + device re-pack) on the same workload shape through the native differentiable path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_POINT = 1_603_584          # SURVEY.md §8(d) / BASELINE.md §4 (dense layers, 2 FLOP per MAC)
PEAK_FP32_MATRIX_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
# HBM bytes per siren launch at this workload from rocprofv3 PMC passes (profiles/r01_pmc_*.txt): FETCH_SIZE 173 MiB-equivalent
# KiB counter (x1024, uncorrected; dominated by the scattered grid gather) + WRITE_SIZE 34.6 MB.  MFMA-bound kernel: context only.
TRAFFIC_BYTES_PER_LAUNCH = int(177.07e6 + 34.60e6)
PEAK_F16_MATRIX_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense fp16 MFMA (v_mfma_f32_32x32x16_f16)


def cpu_baseline(spec, sd, film, seed):
    """Oracle (numpy port of the reference CPU path) on one image of the same workload: 128x128 rays, 24+24 samples, same
    model -- ~10 s of CPU work on the GPU box's host."""
    from fenerf_amd import procedural as proc
    from oracle import fenerf_oracle as O
    S, N, B = 128, 24, 1
    R = S * S
    rng = np.random.default_rng(seed)
    rand = dict(u_jitter=rng.random((B, R, N, 1), dtype=np.float32), theta=np.full((B, 1), np.pi / 2 + 0.1, np.float32),
                phi=np.full((B, 1), np.pi / 2 - 0.05, np.float32), noise_coarse=None, u_fine=rng.random((B * R, N), dtype=np.float32),
                noise_fine=None)
    t0 = time.perf_counter()
    O.render_forward(sd, spec, film, S, 12, 0.88, 1.12, N, rand, hierarchical_sample=True, clamp_mode="relu")
    dt = time.perf_counter() - t0
    return dict(value=R / dt, unit="rays/s", cores=os.cpu_count(), kind="port",
                sample=f"numpy oracle render_forward, {S}x{S} rays, {N}+{N} samples, H=256+96^3 grid, 1 run of {dt:.1f}s "
                       f"(BLAS GEMMs use all {os.cpu_count()} host cores, elementwise ops 1)")


def gstep_leg(spec, sd, dev, B, S, N, precision, iters=5):
    """BASELINE.json's metric also names the generator step: forward + backward (+ the device-side re-pack an optimizer step
    forces) through DoubleImplicitGenerator3d.forward_with_frequencies on the same workload shape, native differentiable path
    (DESIGN.md 4.5).  Reported beside the headline value; never part of the timed region."""
    import functools
    from fenerf_amd.generators import generators as G
    from fenerf_amd.siren import siren as S_
    from fenerf_amd import procedural as proc
    H = spec["hidden_dim"]
    z_dim = spec.get("z_dim", 256)
    mod = S_.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE(hidden_dim=H, z_geo_dim=z_dim, z_app_dim=z_dim, output_dim=spec["output_dim"])
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    mod.spatial_embeddings = torch.nn.Parameter(tsd["spatial_embeddings"].clone())
    mod.load_state_dict(tsd, strict=False)
    mod.precision = precision
    gen = G.DoubleImplicitGenerator3d(functools.partial(S_.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE, hidden_dim=H), z_dim, z_dim, spec["output_dim"])
    gen.siren = mod
    gen = gen.to(dev)
    gen.device = dev
    gen.siren.device = dev
    film = {k: torch.tensor(v, device=dev).requires_grad_(True) for k, v in proc.film_params(spec, B, seed=5).items()}
    kw = dict(img_size=S, fov=12, ray_start=0.88, ray_end=1.12, num_steps=N, h_stddev=0.3, v_stddev=0.155, h_mean=np.pi / 2,
              v_mean=np.pi / 2, hierarchical_sample=True, sample_dist="gaussian", clamp_mode="relu", nerf_noise=0.2, last_back=False)
    w = torch.randn((B, spec["output_dim"] - 1, S, S), device=dev)
    params = [p for n, p in mod.named_parameters() if "mapping_network" not in n]

    def step():
        for p in params:
            p.grad = None
        with torch.no_grad():
            params[0].add_(0)            # version bump like optimizer.step(): the packed streams are rebuilt on the device
        px, _ = gen.forward_with_frequencies(film["freq_geo"], film["freq_app"], film["phase_geo"], film["phase_app"], **kw)
        (px * w).sum().backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    return {"ms": ms, "what": f"forward + backward + device re-pack of one generator step, batch {B} x {S}x{S} rays x {N}+{N} samples "
                              f"({B * S * S * 2 * N} points), native differentiable path, precision {precision}",
            "rays_per_s": B * S * S / (ms * 1e-3), "peak_GB": torch.cuda.max_memory_allocated() / 2**30}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--img-size", type=int, default=128)
    ap.add_argument("--num-steps", type=int, default=24)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gstep", action="store_true", help="skip the generator-step (forward + backward) leg")
    ap.add_argument("--precision", choices=["f32", "f16x3"], default="f16x3",
                    help="arithmetic of the dense layers: exact fp32 MFMA, or error-compensated fp16 MFMA (fp32-class accuracy)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU with torch.distributed.run"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    from fenerf_amd import dist as fdist
    if world > 1:
        fdist.init_from_env(backend="nccl", device=dev)   # RCCL over xGMI; used only for the timing barrier / max-reduce

    from fenerf_amd import _lib, native, procedural as proc
    from fenerf_amd.generators import volumetric_rendering as VR

    spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
    sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
    nat = native.NativeModel(sd, spec, dev, args.precision)
    B, S, N = args.batch, args.img_size, args.num_steps
    R = S * S
    # every rank renders its own images (shard by image, no data-path collective): different latents / poses per rank
    film = proc.film_params(spec, B, seed=1000 + rank)
    tf = tuple(torch.as_tensor(film[k], device=dev) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
    torch.manual_seed(1234 + rank)
    o, d, z, _, _ = VR.sample_rays(B, N, dev, 12, (S, S), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
    u = torch.rand((B * R, N), device=dev)
    opts = _lib.composite_opts("relu", 0.0, fill_mode="seg_padding_background", fill_color="black")

    def step():
        return nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    dt = fdist.max_over_ranks(dt, device=dev)
    rays_total = world * B * R * args.steps
    value = rays_total / dt

    out = None
    if rank == 0:
        # dominant kernel alone (coarse-pass shape == fine-pass shape): hipEvents on the launch stream
        pts = B * R * N
        k_ms = nat.time_siren_rays(o, d, z, *tf, iters=max(5, args.steps // 2))
        achieved = pts * FLOP_PER_POINT / (k_ms * 1e-3) / 1e12
        if args.precision == "f32":
            peak, dtype = PEAK_FP32_MATRIX_TFLOPS, "f32"
            kname, mfma = "siren_kernel<256,true>", "v_mfma_f32_32x32x2_f32 (exact fp32)"
            extra = {}
        else:
            # every algorithmic product is evaluated as 3 fp16 MFMAs (wh*xh + wh*xl + wl*xh, fp32 accumulate): the attainable
            # ceiling of this algorithm on the fp16 pipe is peak/3; `frac` is quoted against the full dense fp16 peak.
            peak, dtype = PEAK_F16_MATRIX_TFLOPS, "f16x3 (error-compensated fp16 MFMA, fp32 accumulate; fp32-class accuracy)"
            kname, mfma = "siren16s_kernel<256,true,false>", "v_mfma_f32_32x32x16_f16, 3 per product"
            extra = {"frac_of_f16x3_ceiling": achieved / (peak / 3)}
        out = {
            "metric": f"rays/s/GPU forward render ({S}x{S}, {N}+{N} samples, H=256 FiLM-SIREN + 32x96^3 grid)",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"configs[1]: CelebA_double_semantic_texture_embedding_256_dim_96 generator, {S}x{S}, "
                                   f"{N}+{N} hierarchical samples, batch {B}/GPU, forward-only render, procedural weights",
                       "img_size": S, "num_steps": N, "batch_per_gpu": B, "sharding": "by image, no collective"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": TRAFFIC_BYTES_PER_LAUNCH,
                         "kernel": kname, "kernel_ms": k_ms, "points_per_launch": pts,
                         "flop_per_point_algorithmic": FLOP_PER_POINT, "mfma": mfma, **extra},
            "rays_per_s_per_gpu": value / world,
        }
        if world == 1 and not args.no_gstep:
            try:
                out["gstep"] = gstep_leg(spec, sd, dev, B, S, N, args.precision)
            except Exception as e:          # the extra leg must never take the headline metric down with it
                out["gstep"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline and world == 1:
            film1 = proc.film_params(spec, 1, seed=1000)
            out["cpu_baseline"] = cpu_baseline(spec, sd, film1, 7)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
