#!/usr/bin/env python
"""Benchmark of the forward-render hot path (BASELINE.json metric: rays/s per GPU).

A step = one fenerf_render_forward pass (coarse SIREN -> composite -> resample -> fine SIREN -> merge ->
composite) over one batch of synthetic rays already resident in HBM.  Workload = BASELINE.json configs[1]:
TextureEmbeddingPiGAN256SEMANTICDISENTANGLE_DIM_96 (H=256 FiLM-SIREN + 32x96^3 grid, 22 channels), 128x128 rays,
24+24 hierarchical samples, batch 1 per GPU, procedural (random-init-range) weights, synthetic latents.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher environment (WORLD_SIZE unset) re-executes itself under `python -m torch.distributed.run
--nproc-per-node N --master-addr 127.0.0.1`; launched BY torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE from
the environment.  One rank per GPU, every rank renders its own images (weak scaling, no data-path collective); RCCL carries
the timing barrier, the max-over-ranks reduce and an all-reduce of ones (`n_ranks_seen`).

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the SIREN kernel, MFMA-bound): algorithmic
FLOPs per launch (1,603,584 FLOP/point x points) / its hipEvent-timed average duration.  Extra objects on the
same line (DESIGN.md §5 has the table): `cpu_baseline` (numpy oracle = port of the reference CPU path, 1 warm-up + 3 timed runs per
shape, N=1); `f32` (the same step at exact-fp32 precision), `f16x3c2` / `f16x2` (the opt-in reduced-precision forward modes, each with
its pixel difference from the headline's image), `sweep64` (north_star's 64x64, 24+24 scaling batch), `render_one_launch`; `gstep`
(one generator step, default precision tier, with its HBM byte model per kernel), `gstep_tape16` / `gstep_amp` / `gstep_amp16` (the
other tiers), `gstep_b6` (configs[2]'s 6-image micro-batch), `gstep_z` (generator(z) + backward on the bare module), `gstep_ddp` /
`gstep_ddp_b6` (the same step through DistributedDataParallel as the reference wraps it, with the recommended arguments, and through
fenerf_amd.dist.GeneratorDataParallel; at every N).
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
PEAK_F16_MATRIX_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense fp16 MFMA
PEAK_HBM_TBPS = 8.0                 # MI355X_MICROARCH.md: HBM3E
SPEC_CLOCK_GHZ = 2.4                # the clock the matrix peaks above are quoted at


COMPACT_LINE_LIMIT = 4096           # bytes; the driver reads the LAST stdout line (round 5's 20.6 KB line came back `parsed: null`)
DETAIL_FILE = "bench_detail.json"


def _sig(v, n=6):
    """floats to n significant digits (the compact line is read by people and by a parser with a tail limit)"""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    return float(f"{v:.{n}g}")


def compact_line(out):
    """The ONE line the driver parses, from the full result object `out` (which goes to bench_detail.json unchanged): the contract's
    keys, `roofline` and `cpu_baseline` as flat objects of scalars, and one scalar (or a small flat object) per extra leg.  No `what`
    strings, no per-kernel tables, no tier legs.  Always < COMPACT_LINE_LIMIT bytes (asserted here and in tests/test_host_cpu.py)."""
    def pick(d, keys):
        return {k: _sig(d[k]) for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], (dict, list))}

    def leg_ms(name, keys=("ms",)):
        d = out.get(name)
        if not isinstance(d, dict):
            return None
        if "error" in d:
            return {"error": str(d["error"])[:120]}
        return pick(d, keys)

    line = {k: _sig(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                      "vs_baseline", "dtype", "data") if k in out}
    line["config"] = {k: v for k, v in out.get("config", {}).items() if not isinstance(v, (dict, list))}
    line["roofline"] = pick(out.get("roofline", {}), ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "points_per_launch",
                                                      "frac_of_f16x3_ceiling", "frac_executed", "clock_ghz_effective", "frac_at_spec_clock"))
    line["roofline"].setdefault("traffic", None)
    if "cpu_baseline" in out:
        line["cpu_baseline"] = pick(out["cpu_baseline"], ("value", "unit", "cores", "threads", "kind", "sample"))
    line["n_ranks_seen"] = out.get("n_ranks_seen")
    line["launches_per_step"] = out.get("launches_per_step")
    for name, keys in (("f32", ("value", "ms_per_step")), ("sweep64", ("value", "ms_per_step", "roofline_frac")),
                       ("gstep", ("ms", "rays_per_s", "peak_GB")), ("gstep_sparse", ("ms", "peak_GB", "kept_frac")),
                       ("gstep_b6", ("ms", "ms_per_image", "peak_GB")), ("gstep_sparse_b6", ("ms", "ms_per_image", "peak_GB", "kept_frac")),
                       ("gstep_ddp", ("ms", "ms_no_ddp", "ms_tuned", "ms_generator_data_parallel", "allreduce_ms_exposed", "allreduce_bytes",
                                      "n_ranks_seen", "dist_backend")),
                       ("gstep_ddp_b6", ("ms", "ms_no_ddp", "ms_generator_data_parallel", "allreduce_ms_exposed"))):
        v = leg_ms(name, keys)
        if v is not None:
            line[name] = v
    if isinstance(out.get("f32"), dict) and isinstance(out["f32"].get("roofline"), dict):
        line["f32"]["roofline_frac"] = _sig(out["f32"]["roofline"].get("frac"))
    g = out.get("gstep")
    if isinstance(g, dict) and isinstance(g.get("roofline"), dict):          # the step's roofs both ways (bytes / 8 TB/s and FLOP / 2,500 TFLOP/s)
        line["gstep"].update(pick(g["roofline"], ("frac", "frac_flop")))
        line["gstep"]["kernels_ms"] = {k["name"]: _sig(k["ms"], 4) for k in g["roofline"].get("per_kernel", [])}
    line["detail"] = DETAIL_FILE
    text = json.dumps(line, separators=(", ", ": "))
    assert len(text.encode()) < COMPACT_LINE_LIMIT and "\n" not in text, len(text)
    return text


def write_detail(out):
    """the full result object (every leg, `what` strings, per-kernel tables) next to the script and, on a gpurun box, under gpurun_out/"""
    text = json.dumps(out, indent=1)
    paths = [os.path.join(ROOT, DETAIL_FILE)]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", DETAIL_FILE))
    for pth in paths:
        try:
            with open(pth, "w") as f:
                f.write(text + "\n")
        except OSError as e:
            print(f"bench.py: could not write {pth}: {e}", file=sys.stderr)


def pmc_traffic():
    """HBM-side bytes per SIREN launch at the default workload (393,216 points) from the newest committed rocprofv3 PMC summary
    (separate --pmc FETCH_SIZE / WRITE_SIZE passes over this file's default command, tools/gpu_r3.sh pmc -> tools/pmc_summary.py):
    a counter pass cannot run inside the bench process, so the file is parsed at run time -- nothing is typed in."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_siren16w_f16x3.txt")))
    if not files:
        return None
    vals = {}
    for line in open(files[-1]):
        parts = line.strip().split(",")
        if len(parts) == 3 and parts[0] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[parts[0]] = float(parts[1]) * 1024.0                 # both counters are in KiB
    if len(vals) != 2:
        return None
    return {
        "source": os.path.relpath(files[-1], ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; parsed at run time)",
        "fetch_raw_bytes": vals["FETCH_SIZE"],
        "fetch_x2_bytes": 2 * vals["FETCH_SIZE"],     # MI355X_MICROARCH.md HBM: gfx950 reports 1/2 of wide coalesced reads; upper bound here
        "write_bytes": vals["WRITE_SIZE"],            # = 393,216 points x 22 channels x 4 B
        # compulsory bytes of one launch: outputs 34.6 MB + z 1.57 MB + rays 0.39 MB + weight stream 2.75 MB
        "algorithmic_bytes": 393216 * 22 * 4 + 1.57e6 + 0.39e6 + 2.75e6,
        "note": "excess over algorithmic = the 8 x 128-B trilinear corner fetches per point (113 MB grid, TCC hit ~96 %); "
                "MFMA-bound kernel, 0.15-0.28 TB/s: context, not the limiter",
    }


def quiesce_gc():
    """Before a timed region: collect, then move every live object to the permanent generation.  This process holds ~10^6 Python objects
    (state dicts, modules, ctypes mirrors); a full collection of them takes 70 ms and is triggered by allocation counts, i.e. at a fixed
    but arbitrary step -- measured in round 5 as one 85-ms generator step in ~100 (tools/exp/ddp_stall_probe.py: the stall IS a generation-2
    collection; none with the collector frozen).  A 56-ms headline region cannot absorb that.  Training loops over these classes want the
    same two lines after building their models."""
    import gc
    gc.collect()
    gc.freeze()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch_command(n, argv, script=None, port=None):
    """The command `python bench.py --gpus N` turns itself into when no launcher set WORLD_SIZE: one rank per GPU under
    torch.distributed.run on 127.0.0.1 (reference analogue: mp.spawn over GPUs, train_double_latent_semantic.py:584)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or _free_port()), script or os.path.abspath(__file__)] + list(argv)


def _oracle_inputs(S, N, seed):
    B, R = 1, S * S
    rng = np.random.default_rng(seed)
    return dict(u_jitter=rng.random((B, R, N, 1), dtype=np.float32), theta=np.full((B, 1), np.pi / 2 + 0.1, np.float32),
                phi=np.full((B, 1), np.pi / 2 - 0.05, np.float32), noise_coarse=None,
                u_fine=rng.random((B * R, N), dtype=np.float32), noise_fine=None)


def _oracle_run(spec, sd, film, S, N, hier, seed):
    """one image through the numpy oracle (BLAS GEMMs threaded, every elementwise pass on one core): detail file only since round 6"""
    from oracle import fenerf_oracle as O
    rand = _oracle_inputs(S, N, seed)
    t0 = time.perf_counter()
    O.render_forward(sd, spec, film, S, 12, 0.88, 1.12, N, rand, hierarchical_sample=hier, clamp_mode="relu")
    return time.perf_counter() - t0


def _oracle_torch_run(spec, tsd, film, S, N, hier, seed, max_batch_size):
    """one image through the torch-CPU oracle: the reference's ATen statements (F.grid_sample, F.linear, torch.sin, cumprod, searchsorted,
    sort / gather), all of them on torch's intra-op thread pool"""
    from oracle import fenerf_oracle_torch as OT
    rand = _oracle_inputs(S, N, seed)
    t0 = time.perf_counter()
    OT.render_forward(tsd, spec, film, S, 12, 0.88, 1.12, N, rand, hierarchical_sample=hier, clamp_mode="relu", max_batch_size=max_batch_size)
    return time.perf_counter() - t0


from fenerf_amd.host import effective_host_cores          # noqa: E402  (cgroup quota / affinity aware; shared with the command-line front ends)


CPU_BASELINE_BUDGET_S = 40.0        # bound on the whole leg (the contract: ~10-30 s of CPU work, the default bench run within minutes)


def cpu_baseline(spec, sd, film, seed, full=True, budget_s=CPU_BASELINE_BUDGET_S):
    """the reference's CPU path on the GPU box's host cores = oracle/fenerf_oracle_torch.py (bit-identical to the imported
    reference on the committed fixtures, tests/test_oracle_golden.py) under torch.set_num_threads(all host cores): 1 warm-up + 3 timed runs
    per shape, median; shapes = configs[1] (128x128, 24+24: the headline workload, `value`), configs[0] (64x64, 12 coarse) and the 64x64,
    24+24 scaling batch; point chunks of 2,400,000 as the reference's render scripts pass (render_multiview_images_double_semantic.py:36).
    The leg is bounded: when a slow host would take it past `budget_s`, later shapes run 1 timed run or are skipped (recorded in `runs`).
    `full=False` (N > 1 lines): headline shape only, 1 + 1 runs.  The numpy oracle (the parity checker; single-threaded outside BLAS) is
    timed once on the headline shape for the detail file."""
    from oracle import fenerf_oracle_torch as OT
    cores = effective_host_cores()          # all the host cores this process is allowed (cgroup quota honoured), one thread on each
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    threads = torch.get_num_threads()
    t_leg = time.perf_counter()
    try:
        tsd = OT.state_to_torch(sd)
        shapes = [("configs[1]: 128x128 rays, 24+24 samples", 128, 24, True)]
        if full:
            shapes += [("configs[0]: 64x64 rays, 12 coarse samples, no resampling", 64, 12, False),
                       ("scaling batch: 64x64 rays, 24+24 samples", 64, 24, True)]
        res = []
        for name, S, N, hier in shapes:
            left = budget_s - (time.perf_counter() - t_leg)
            if res and left <= 0:
                res.append({"shape": name, "skipped": "leg budget spent"})
                continue
            warm = _oracle_torch_run(spec, tsd, film, S, N, hier, seed, 2400000)
            left = budget_s - (time.perf_counter() - t_leg)
            n = 3 if (full and 3 * warm <= max(left, 0.0)) else 1
            ts = [_oracle_torch_run(spec, tsd, film, S, N, hier, seed + 1 + i, 2400000) for i in range(n)]
            res.append({"shape": name, "rays": S * S, "warmup_s": round(warm, 3), "seconds": [round(t, 3) for t in ts],
                        "rays_per_s": S * S / float(np.median(ts))})
        head = res[0]
        out = dict(value=head["rays_per_s"], unit="rays/s", cores=cores, threads=threads, kind="port",
                   sample=f"torch-CPU oracle (the reference's ATen statements, every pass threaded) on {threads} threads = the {cores} host cores this "
                          f"process may use ({os.cpu_count()} logical CPUs, cgroup quota / affinity honoured), one 128x128 x 24+24 image (configs[1]), "
                          f"1 warm-up + {len(head['seconds'])} timed run(s), median",
                   logical_cpus=os.cpu_count(), runs=res)
        t_head = float(np.median(head["seconds"]))
        if full:
            try:        # detail only, while the budget lasts: the reference's default chunk (staged_forward's max_batch_size=50000) ...
                if budget_s - (time.perf_counter() - t_leg) > 1.5 * t_head:
                    out["chunk_50000"] = {"rays_per_s": 128 * 128 / _oracle_torch_run(spec, tsd, film, 128, 24, True, seed + 9, 50000)}
                if budget_s - (time.perf_counter() - t_leg) > 5 * t_head:      # ... and the numpy oracle (3-5 x the torch edition's time)
                    t_np = _oracle_run(spec, sd, film, 128, 24, True, seed + 9)
                    out["numpy_oracle"] = {"rays_per_s": 128 * 128 / t_np, "seconds": round(t_np, 3),
                                           "note": "the parity checker; BLAS GEMMs threaded, elementwise passes on one core (round 5's cpu_baseline)"}
            except Exception as e:
                out["detail_error"] = f"{type(e).__name__}: {e}"
        out["leg_seconds"] = round(time.perf_counter() - t_leg, 2)
        return out
    finally:
        torch.set_num_threads(prev)


def gstep_leg(spec, sd, dev, B, S, N, precision, iters=8, breakdown=True, grad_precision="f32", per_step_median=False, sparse=False):
    """BASELINE.json's metric also names the generator step: forward + backward (+ the device-side re-pack an optimizer step
    forces) through DoubleImplicitGenerator3d.forward_with_frequencies on the same workload shape, native differentiable path
    (DESIGN.md 4.5).  Reported beside the headline value; never part of the timed region.

    `roofline`: the step is HBM-bound -- the forward-save passes write the tape (pre-FiLM accumulators of all L FiLM layers), the
    backward chain reads it and writes the d(theta) dump, the weight-gradient kernels read the dump and the tape again.  algorithmic
    bytes = those streams plus the per-point inputs / outputs of each kernel (formulas below, sizes from the library:
    fenerf_siren_tape_floats / fenerf_siren_dtheta_floats); `achieved` = bytes / step time against 8 TB/s; `per_kernel` = the same per
    launch group with device times from hipEvent pairs around every launch (fenerf_phase_timing), measured in separate instrumented
    steps."""
    import functools
    from fenerf_amd.generators import generators as G
    from fenerf_amd.siren import siren as S_
    from fenerf_amd import native, procedural as proc
    H = spec["hidden_dim"]
    z_dim = spec.get("z_dim", 256)
    mod = S_.TextureEmbeddingPiGAN128SEMANTICDISENTANGLE(hidden_dim=H, z_geo_dim=z_dim, z_app_dim=z_dim, output_dim=spec["output_dim"])
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    mod.spatial_embeddings = torch.nn.Parameter(tsd["spatial_embeddings"].clone())
    mod.load_state_dict(tsd, strict=False)
    mod.precision = precision
    mod.grad_precision = grad_precision
    mod.sparse_backward = bool(sparse)       # opt-in: backward over the samples with a non-zero gradient row only (generators/autograd.py; exact)
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

    bump = min(params, key=lambda t: t.numel())

    def step():
        for p in params:
            p.grad = None
        with torch.no_grad():
            bump.add_(0)                 # version bump like optimizer.step() (on the smallest tensor: the optimizer itself is not timed):
                                         # the packed streams are rebuilt on the device
        px, _ = gen.forward_with_frequencies(film["freq_geo"], film["freq_app"], film["phase_geo"], film["phase_app"], **kw)
        (px * w).sum().backward()

    torch.cuda.reset_peak_memory_stats()
    quiesce_gc()
    for _ in range(3):           # the first steps after the allocator's first 10 GB run 3-10 % slow (kernel trace: 16.6, 15.4, 15.0, 14.9 ...)
        step()
    torch.cuda.synchronize()
    if per_step_median:      # long steps (the 6-image micro-batch, ~70 ms): median of individually timed steps -- one allocator hiccup in a
        ts = []              # mean of three moved the leg by 7 % in a round-6 run (75.2 against 68.6-70.4 ms in every other run)
        for _ in range(iters):
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ms = sorted(ts)[len(ts) // 2]
    else:
        t0 = time.perf_counter()
        for _ in range(iters):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
    out = {"ms": ms, "what": f"generator.forward_with_frequencies + backward + device re-pack, batch {B} x {S}x{S} rays x {N}+{N} samples "
                             f"({B * S * S * 2 * N} points): the render part of a generator step -- the two mapping networks, their backward "
                             f"and the optimizer are NOT in it (the reference's own step, generator_ddp(z) + backward [+ Adam], is "
                             f"gstep_ddp.ms_no_ddp / .ms / .ms_with_optimizer); native differentiable path, precision {precision}, weight-gradient operands "
                             + {"f32": "fp32 class (default)", "amp": "bf16, one MFMA per product: AMP class, opt-in (siren.grad_precision = 'amp')",
                                "amp16": "bf16 weight-gradient operands AND the 16-bit tape: AMP class throughout (FiLM frequency gradients included), "
                                         "opt-in (siren.grad_precision = 'amp16')",
                                "tape16": "fp32 class, with the 16-bit tape between forward and backward (frac(theta) as fixed point, 2 instead of 4 bytes per "
                                          "(point, layer-feature)): gradients within ~1.2e-4 of fp64 autograd instead of ~4e-5 -- a tier between the default "
                                          "and AMP, opt-in (siren.grad_precision = 'tape16')"}[grad_precision],
           "rays_per_s": B * S * S / (ms * 1e-3), "peak_GB": torch.cuda.max_memory_allocated() / 2**30}
    if sparse:
        from fenerf_amd.generators import autograd as GA
        GA.SparseHierarchicalRenderFunction.verify()
        kept, total = GA.SparseHierarchicalRenderFunction.last_kept
        kept = int(kept)
        out["what"] += ("; SPARSE backward (opt-in, siren.sparse_backward): the forward is the no-grad render, the backward runs forward-save / chain / "
                        "weight gradients only over the samples whose upstream gradient row is not all zero -- exact: under the relu clamp a sample "
                        "with sigma + noise <= 0 has weight 0 and relu' = 0, the reference's autograd multiplies those zeros through the network; "
                        "data-dependent (this workload's procedural density field)")
        out["kept_samples"], out["samples"], out["kept_frac"] = kept, total, kept / total
        groups = GA.SparseHierarchicalRenderFunction.last_groups
        out["launch_groups"] = [{"images": len(g), "slots_per_image": c} for g, c in groups]       # images padded to their group's fullest
        out["slots_frac"] = sum(len(g) * c for g, c in groups) / total
        return out
    if not breakdown:
        return out
    # ---- HBM roofline of the step: algorithmic bytes per launch group (all sizes per sample point, x points of the step)
    nat = mod.native_differentiable(dev)
    L, C, G_ = spec["n_geo"] + spec["n_color"], spec["output_dim"], spec["grid_ch"]
    pts = B * ((S * S * N + 31) // 32 * 32) * 2                         # coarse + fine pass, whole 32-point tiles per image
    fmt = mod.tape_format(nat, film_only=False)                          # include/fenerf.h FENERF_TAPE_*: fp32 accumulators, or 16-bit phases
    tape_b = nat.tape_floats(pts, fmt) * 4.0 / pts                       # the tape of L layers, bytes per point
    dump = nat.dump_bytes_per_point(tape_format=fmt)                     # what the chain writes / the weight gradients read, per layer-feature
    row, xyz = 4.0 * C, 12.0
    model = {
        "forward_save": tape_b + (128.0 if G_ else 0.0) + row + 2 * xyz,                        # tape + grid features + outputs written; points, dirs read
        "chain": tape_b + 2 * row + xyz + dump["chain_write"] * L * H,                           # tape, out, d_out read; dump written
        "wgrad_square": dump["square_read"] * (L - 1) * H,                                       # d(theta)_l and x_{l-1} (or tape_{l-1}) of the L-1 square layers
        "wgrad_thin": dump["thin_read"] * H + (128.0 if G_ else 0.0) + 2 * xyz + 3 * row,        # four thin jobs: two dumps, two tape layers, points / dirs / features, rows
    }
    # the same launch groups by FLOP (round 6: the counters say these kernels are issue-bound, not byte-bound -- both roofs are reported):
    # algorithmic dense-layer FLOP per point, 2 per MAC.  forward_save = the forward; the chain multiplies d(theta) by the same
    # matrices transposed (same MACs); the weight gradients are the outer products of the same shapes: L-1 square H x H layers, and the thin
    # ones (first layer 3 x H, colour layer 0's dirs + grid columns, sigma / rgb / folded label rows)
    thin_cols = 3 + (3 + G_ if spec["kind"] != "spatial" else 3) + (C - 1) + 1
    flop_model = {"forward_save": float(FLOP_PER_POINT) if (H, L, C, G_) == (256, 11, 22, 32) else None,
                  "chain": float(FLOP_PER_POINT) if (H, L, C, G_) == (256, 11, 22, 32) else None,
                  "wgrad_square": 2.0 * (L - 1) * H * H, "wgrad_thin": 2.0 * thin_cols * H}
    steps_b = 3
    with native.phase_timing() as t:
        for _ in range(steps_b):
            step()
    per_kernel, accounted = [], 0.0
    for name, b_pt in model.items():
        k_ms = t.ms.get(name, 0.0) / steps_b
        nbytes = b_pt * pts
        accounted += k_ms
        flop = flop_model[name] * pts if flop_model.get(name) else None
        per_kernel.append({"name": name, "ms": k_ms, "launches": t.calls.get(name, 0) // steps_b, "algorithmic_bytes": nbytes,
                           "achieved_TBps": nbytes / (k_ms * 1e-3) / 1e12 if k_ms else None,
                           "frac": nbytes / (k_ms * 1e-3) / 1e12 / PEAK_HBM_TBPS if k_ms else None,
                           "algorithmic_flop": flop, "achieved_TFLOPs": flop / (k_ms * 1e-3) / 1e12 if (k_ms and flop) else None,
                           "frac_flop": flop / (k_ms * 1e-3) / 1e12 / PEAK_F16_MATRIX_TFLOPS if (k_ms and flop) else None})
    rest = {k: v / steps_b for k, v in t.ms.items() if k not in model}
    total_bytes = sum(k["algorithmic_bytes"] for k in per_kernel)
    total_flop = sum(k["algorithmic_flop"] or 0.0 for k in per_kernel)
    out["roofline"] = {"bound": "hbm", "algorithmic_bytes": total_bytes, "achieved": total_bytes / (ms * 1e-3) / 1e12, "achieved_TBps": total_bytes / (ms * 1e-3) / 1e12,
                       "peak": PEAK_HBM_TBPS, "unit": "TB/s", "frac": total_bytes / (ms * 1e-3) / 1e12 / PEAK_HBM_TBPS,
                       "algorithmic_flop": total_flop, "achieved_TFLOPs": total_flop / (ms * 1e-3) / 1e12, "peak_flop": PEAK_F16_MATRIX_TFLOPS,
                       "frac_flop": total_flop / (ms * 1e-3) / 1e12 / PEAK_F16_MATRIX_TFLOPS,
                       "per_kernel": per_kernel, "other_library_launches_ms": rest,
                       "ms_not_in_library_kernels": ms - accounted - sum(rest.values()),
                       "launch_groups_per_step": sum(t.calls.values()) // steps_b,
                       "bytes_per_point": {"tape": tape_b, "tape_format": {0: "f32", 1: "u16", 2: "f32_w (fp32 tape; FiLM frequency gradients from the weight-gradient sums)"}[int(fmt)], **dump},     # tape per point; the others per (point x layer-feature)
                       "note": "algorithmic bytes = tape written once and read by the chain, the chain's dump written once and read once, the "
                               "tape layers the weight-gradient kernels re-read, per-point rows; `frac` = bytes / whole step time / 8 TB/s; "
                               "per_kernel times from hipEvent pairs in instrumented steps (their sum + torch glue = ms)"}
    return out


def curriculum_generator(spec, sd, dev, precision, seed=11):
    """The generator BASELINE.json names -- DoubleImplicitGenerator3d over TextureEmbeddingPiGAN256SEMANTICDISENTANGLE_DIM_96 with both
    mapping networks -- built like train_double_latent_semantic.py:142 builds it; render weights = the bench's procedural ones, mapping
    networks from a fixed torch seed (identical on every rank; DDP broadcasts rank 0's anyway)."""
    from fenerf_amd import curriculums
    from fenerf_amd.generators import generators as G
    from fenerf_amd.siren import siren as S_
    cur = curriculums.CelebA_double_semantic_texture_embedding_256_dim_96
    torch.manual_seed(seed)
    gen = G.DoubleImplicitGenerator3d(getattr(S_, cur["model"]), cur["latent_geo_dim"], cur["latent_app_dim"], cur["output_dim"])
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    gen.siren.load_state_dict(tsd, strict=False)
    gen = gen.to(dev)
    gen.set_device(dev)
    gen.siren.precision = precision
    gen.train()
    return gen, cur, curriculums


GSTEP_DDP_KEYS = ("what", "ms", "ms_no_ddp", "ms_no_ddp_runs", "allreduce_ms_exposed", "allreduce_bytes", "allreduce_bytes_largest_tensor", "ddp_bucket_cap_mb",
                  "allreduce_per_micro_batch", "ms_with_optimizer", "ms_optimizer_step_4_micro_batches", "ms_optimizer_step_4_micro_batches_one_allreduce",
                  "ms_tuned", "allreduce_ms_exposed_tuned", "tuned_config", "ms_generator_data_parallel",
                  "allreduce_ms_exposed_generator_data_parallel", "collectives_per_step_generator_data_parallel", "rays_per_s",
                  "n_ranks", "n_ranks_seen", "batch_per_rank", "dist_backend", "peak_GB")
# `ms_tuned`: what fenerf_amd recommends instead of the reference's wrapper -- fenerf_amd.dist.prepare_for_ddp(generator) (two-node backward:
# the grid gradient reaches DDP before the weight-gradient kernels run, so its all-reduce overlaps them) + RECOMMENDED_DDP_KWARGS
TUNED_DDP = dict(find_unused_parameters=False, gradient_as_bucket_view=True, bucket_cap_mb=128)       # == fenerf_amd.dist.RECOMMENDED_DDP_KWARGS


def ddp_timed_leg(model, params, loss_of, opt, dev, world, barrier, max_over_ranks, iters, rays_per_rank, what, batch_per_rank, tuned_prepare=None,
                  agree=lambda ok: ok):
    """Times `loss_of(m).backward()` on the bare module and on DistributedDataParallel(module, find_unused_parameters=True) with the
    headline's bracket (barrier on both sides, max over ranks) -> the `gstep_ddp` object (GSTEP_DDP_KEYS).  Backend-agnostic: the GPU
    bench hands it the generator over RCCL; `--dist-check` (CPU, gloo, world 2) hands it a small stand-in so that the wrapper, the
    rank census, the byte count and the schema of the N > 1 line are covered without a GPU (tests/test_dist_cpu.py).
    `agree(ok)` = "every rank says ok" (an all-reduce at N > 1): the bare-module run -- which contains no collective of its own -- is made
    symmetric with it, so that a rank that fails there (out of memory, say) takes every rank out of the leg together instead of leaving
    the others waiting in DistributedDataParallel's all-reduce until the launcher's timeout."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    bump = min(params, key=lambda t: t.numel())

    def run(m, n, optimizer, guarded=False):
        def step():
            opt.zero_grad(set_to_none=True)
            if not optimizer:
                with torch.no_grad():
                    bump.add_(0)             # the version bump an optimizer step causes: packed weight streams are rebuilt on the device
            loss_of(m).backward()
            if optimizer:
                opt.step()

        def steps(k):                        # guarded (no collective inside the steps): an exception is held until every rank has passed the barriers
            try:
                for _ in range(k):
                    step()
            except Exception as e:
                if not guarded:
                    raise
                return e
            return None
        err = steps(3)           # as gstep_leg: the first steps after the allocator's first 10 GB run 3-10 % slow
        barrier()
        t0 = time.perf_counter()
        err = err or steps(n)
        barrier()
        ms_ = max_over_ranks(time.perf_counter() - t0) / n * 1e3
        if guarded and not agree(err is None):
            raise RuntimeError(f"the generator step failed on a rank before any collective (this rank: {type(err).__name__ if err else 'ok'}: {err})")
        return ms_

    cuda = dev.type == "cuda"
    if cuda:
        torch.cuda.reset_peak_memory_stats()
    quiesce_gc()
    ms_bare = run(model, iters, False, guarded=True)
    created = False
    if not dist.is_initialized():            # N = 1 without a launcher: a one-rank group, so that the leg exists at every N
        kw = {"device_id": dev} if cuda else {}
        dist.init_process_group("nccl" if cuda else "gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, **kw)
        created = True
    try:
        ddp = DDP(model, device_ids=[dev.index] if cuda else None, find_unused_parameters=True)      # the reference's wrapper, train...py:148
        ms = run(ddp, iters, False)
        ms_opt = run(ddp, max(2, iters // 2), True)
        # One OPTIMIZER step of the reference's loop = `batch_split` (4) micro-batches (train_double_latent_semantic.py:407-446).  The reference
        # wraps no no_sync() around them: four all-reduces of the full gradient set per optimizer step.  fenerf_amd.dist.micro_batch_sync
        # accumulates locally and reduces in the last micro-batch's backward: the same gradients (fp32 summation order), one all-reduce.
        # Both patterns through the reference's wrapper, so that a multi-GPU run shows them side by side.
        from fenerf_amd import dist as fdist
        import contextlib

        def opt_step_ms(sync_once, n=5, MB=4):
            def one():
                opt.zero_grad(set_to_none=True)
                for s_ in range(MB):
                    with (fdist.micro_batch_sync(ddp, s_, MB) if sync_once else contextlib.nullcontext()):
                        loss = loss_of(ddp)
                        loss.item()              # the reference reads every micro-batch's loss on the host (generator_losses.append(g_loss.item()), :444)
                        loss.backward()
                opt.step()
            one()
            per_step = []
            for _ in range(n):                   # MEDIAN of n optimizer steps, each in the headline's bracket: DistributedDataParallel has one-off
                barrier()                        # events at fixed iteration counts (a 70-ms stall in one backward of ~30 was traced to the wrapper,
                t0 = time.perf_counter()         # not to a kernel: every other micro-batch of the same pattern takes 13.4 ms) that a mean of 2-3 steps
                one()                            # turns into +25 - 50 %
                barrier()
                per_step.append(max_over_ranks(time.perf_counter() - t0) * 1e3)
            return sorted(per_step)[len(per_step) // 2]
        ms_mb4, ms_mb4_once = opt_step_ms(False), opt_step_ms(True)
        del ddp
        opt.zero_grad(set_to_none=True)
        if tuned_prepare is not None:
            tuned_prepare(True)
        try:
            ddp = DDP(model, device_ids=[dev.index] if cuda else None, **TUNED_DDP)
            ms_tuned = run(ddp, iters, False)
            del ddp
            opt.zero_grad(set_to_none=True)
            # round 5: fenerf_amd.dist.GeneratorDataParallel -- the same averaged gradients without DDP's per-parameter launches and bucket
            # bookkeeping: the grid's gradient reduced in place from its hook, the other 36 tensors as one flat buffer
            ms_gdp = gdp_collectives = ddp = None
            try:                              # its own guard: a failure here must not take the DistributedDataParallel figures above with it
                ddp = fdist.GeneratorDataParallel(model)
                ms_gdp = run(ddp, iters, False)
                gdp_collectives = ddp.last_sync["collectives"]
            except Exception as e:
                gdp_collectives = f"{type(e).__name__}: {e}"
            finally:
                try:
                    ddp.detach_hooks()
                except Exception:
                    pass
            del ddp
            opt.zero_grad(set_to_none=True)
        finally:
            if tuned_prepare is not None:
                tuned_prepare(False)
        # the bare module once more, last: the first legs of a fresh generator run up to 0.3 ms slow (allocator growth, clocks) and every
        # `exposed` figure is a difference against this number -- the lower of the two runs is the baseline, both are reported
        ms_bare_runs = [ms_bare, run(model, iters, False, guarded=True)]
        ms_bare = min(ms_bare_runs)
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)
        out = {"what": what, "ms": ms, "ms_no_ddp": ms_bare, "ms_no_ddp_runs": ms_bare_runs, "allreduce_ms_exposed": ms - ms_bare,
               "allreduce_bytes": sum(p.numel() * 4 for p in params), "allreduce_bytes_largest_tensor": max(p.numel() for p in params) * 4,
               "ddp_bucket_cap_mb": 25, "allreduce_per_micro_batch": True, "ms_with_optimizer": ms_opt,
               "ms_optimizer_step_4_micro_batches": ms_mb4, "ms_optimizer_step_4_micro_batches_one_allreduce": ms_mb4_once,
               "ms_tuned": ms_tuned, "allreduce_ms_exposed_tuned": ms_tuned - ms_bare,
               "tuned_config": ("fenerf_amd.dist.prepare_for_ddp(generator) [grid gradient delivered before the weight-gradient kernels], " if tuned_prepare else "")
                               + ", ".join(f"{k}={v}" for k, v in TUNED_DDP.items()),
               "ms_generator_data_parallel": ms_gdp, "allreduce_ms_exposed_generator_data_parallel": (ms_gdp - ms_bare) if ms_gdp is not None else None,
               "collectives_per_step_generator_data_parallel": gdp_collectives,
               "rays_per_s": world * rays_per_rank / (ms * 1e-3), "n_ranks": world, "n_ranks_seen": int(seen.item()),
               "batch_per_rank": batch_per_rank, "dist_backend": dist.get_backend(),
               "peak_GB": torch.cuda.max_memory_allocated() / 2**30 if cuda else None}
        opt.zero_grad(set_to_none=True)
    finally:
        if created:
            dist.destroy_process_group()
    assert tuple(out) == GSTEP_DDP_KEYS
    return out


def gstep_ddp_leg(spec, sd, dev, rank, world, B, S, N, precision, barrier, max_over_ranks, iters=6, agree=lambda ok: ok):
    """The reference's generator step as its training loop runs it (train_double_latent_semantic.py:148-150, 402-446):
    `generator_ddp = DDP(generator, find_unused_parameters=True)`; per micro-batch `gen_imgs, _ = generator_ddp(z_geo, z_app, **metadata)`
    (both mapping networks inside), `loss.backward()` -- DDP's bucketed all-reduce of EVERY generator gradient over RCCL/xGMI fires in
    that backward; there is no no_sync() around the `batch_split` micro-batches, so every micro-batch all-reduces (kept: reference
    behaviour, `allreduce_per_micro_batch`) -- then Adam.  Every rank renders its own latents.  Reported: `ms` (DDP step), `ms_no_ddp`
    (same rank, same step on the bare module: no collective; at N = 1 this is the reference's G step `generator(z)` + backward),
    `allreduce_ms_exposed` = their difference, `allreduce_bytes` = fp32 bytes of all gradients DDP reduces (113 MB of them the 96^3
    grid), `ms_with_optimizer` (+ torch.optim.Adam.step() on all generator parameters)."""
    err = None
    try:
        gen, cur, curriculums = curriculum_generator(spec, sd, dev, precision)
    except Exception as e:
        err = e
    if not agree(err is None):               # every rank leaves together (no rank waits for one that could not build its model)
        raise RuntimeError(f"building the generator failed on a rank (this rank: {type(err).__name__ if err else 'ok'}: {err})")
    md = {**curriculums.extract_metadata(cur, 60000), "img_size": S, "num_steps": N, "nerf_noise": 0.5}    # the 128 x 128 stage; noise as mid-fade (train...py:276)
    torch.manual_seed(4242 + rank)
    zg, za = torch.randn(B, cur["latent_geo_dim"], device=dev), torch.randn(B, cur["latent_app_dim"], device=dev)
    w = torch.randn((B, cur["output_dim"] - 1, S, S), device=dev) / (B * S * S)
    params = [p for p in gen.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=cur[50000]["gen_lr"], betas=tuple(float(v) for v in cur["betas"]), weight_decay=cur["weight_decay"])

    def loss_of(m):
        px, _ = m(zg, za, **md)
        return (px * w).sum()

    what = (f"generator_ddp(z_geo, z_app, **metadata) + backward per rank: batch {B} x {S}x{S} rays x {N}+{N} samples, both mapping networks, "
            "DDP(find_unused_parameters=True) bucketed all-reduce of all generator gradients inside backward (every micro-batch all-reduces: the "
            "reference has no no_sync()), device re-pack of the changed weights; precision " + precision)
    from fenerf_amd import dist as fdist
    assert fdist.RECOMMENDED_DDP_KWARGS == TUNED_DDP
    return ddp_timed_leg(gen, params, loss_of, opt, dev, world, barrier, max_over_ranks, iters, B * S * S, what, B,
                         tuned_prepare=lambda on: fdist.prepare_for_ddp(gen, on), agree=agree)


def dist_check(args):
    """Rendezvous only (CPU-capable, backend gloo): proves the self-launch produced `--gpus` ranks that see each other."""
    import torch.distributed as dist
    from fenerf_amd import dist as fdist
    rank, local_rank, world = fdist.init_from_env(backend=args.dist_backend)
    ones = torch.ones(1)
    if world > 1:
        dist.all_reduce(ones)
    # the DDP generator-step leg of the N > 1 line on a stand-in module (no GPU here): wrapper, bracket, rank census, byte count, schema
    torch.manual_seed(3)
    stand_in = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 4))
    params = list(stand_in.parameters())
    x = torch.randn(8, 16, generator=torch.Generator().manual_seed(100 + rank))
    dev = torch.device("cpu")
    def agree(ok):
        t = torch.tensor([1.0 if ok else 0.0])
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)
    leg = ddp_timed_leg(stand_in, params, lambda m: m(x).square().sum(), torch.optim.Adam(params, lr=1e-3), dev, world,
                        (dist.barrier if world > 1 else (lambda: None)), lambda v: fdist.max_over_ranks(v), 3, 8, "stand-in module (dist-check)", 8,
                        agree=agree)
    # after DDP steps every rank holds the same parameters (all-reduced gradients, same optimizer): checksum agreement
    csum = torch.tensor([float(sum(p.detach().double().sum() for p in params))], dtype=torch.float64)
    sums = [torch.zeros_like(csum) for _ in range(world)]
    if world > 1:
        dist.all_gather(sums, csum)
    else:
        sums = [csum]
    if rank == 0:
        print(json.dumps({"dist_check": True, "n_gpus": args.gpus, "world_size": world, "n_ranks_seen": int(ones.item()),
                          "backend": args.dist_backend, "gstep_ddp": leg,
                          "params_identical_across_ranks": all(float(t) == float(sums[0]) for t in sums)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--img-size", type=int, default=128)
    ap.add_argument("--num-steps", type=int, default=24)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick-cpu-baseline", action="store_true", help="headline shape only, 1 warm-up + 1 timed run")
    ap.add_argument("--no-gstep", action="store_true", help="skip the generator-step (forward + backward) leg")
    ap.add_argument("--no-gstep-b6", action="store_true", help="skip the 6-image generator micro-batch (configs[2]) legs")
    ap.add_argument("--gstep-sparse-b6", action="store_true",
                    help="also time the 6-image micro-batch with the opt-in exact-sparsity backward (not in the default command: its 6-image no-grad "
                         "launches of the headline kernel would mix into that kernel's rocprofv3 average)")
    ap.add_argument("--no-gstep-ddp", action="store_true", help="skip the DistributedDataParallel generator-step legs (run at every N)")
    ap.add_argument("--no-f32", action="store_true", help="skip the exact-fp32 leg")
    ap.add_argument("--no-sweep64", action="store_true", help="skip the 64x64, 24+24 scaling-batch leg")
    ap.add_argument("--dead-ends", action="store_true", help="also run the documented dead ends (f16x2 forward, the one-launch render): detail file only")
    ap.add_argument("--precision", choices=["f32", "f16x3"], default="f16x3",
                    help="arithmetic of the dense layers: exact fp32 MFMA, or error-compensated fp16 MFMA (fp32-class accuracy)")
    ap.add_argument("--dist-check", action="store_true", help="rendezvous + all-reduce of ones only (no GPU needed)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) on GPUs; gloo for the CPU dry run of --dist-check and for --one-device")
    ap.add_argument("--one-device", action="store_true",
                    help="every rank on cuda:0 (with --dist-backend gloo: RCCL refuses two ranks on one device): the N > 1 code path on a one-GPU box")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's N=1 command with N changed: become the launcher (one rank per GPU over RCCL)
        import subprocess
        sys.exit(subprocess.call(self_launch_command(args.gpus, argv)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; use --nproc-per-node {args.gpus}")
    if args.dist_check:
        return dist_check(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = 0 if args.one_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    from fenerf_amd import dist as fdist
    n_ranks_seen = 1
    # FENERF_BENCH_FORCE_DIST=1: take the N > 1 branch (RCCL process group, barrier, max-reduce, all-gather) with a single rank too,
    # so that a one-GPU box exercises the code the driver's 2/4/8-GPU runs depend on (tests/test_gpu_parity.py)
    use_dist = world > 1 or bool(os.environ.get("FENERF_BENCH_FORCE_DIST"))
    if use_dist:
        if args.one_device and world > 1 and args.dist_backend == "nccl":
            sys.exit("bench.py: --one-device with N > 1 needs --dist-backend gloo (RCCL refuses two ranks on one device)")
        fdist.init_from_env(backend=args.dist_backend, device=dev, force=True)   # RCCL over xGMI; timing barrier / max-reduce / rank census only
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        n_ranks_seen = int(ones.item())

    from fenerf_amd import _lib, native, procedural as proc
    from fenerf_amd.generators import volumetric_rendering as VR

    spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
    sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
    if os.environ.get("FENERF_BENCH_ZERO_WEIGHTS"):    # DVFS probe only (data-dependent power): same instruction stream, all-zero operands
        sd = {k: np.zeros_like(v) for k, v in sd.items()}
    opts = _lib.composite_opts("relu", 0.0, fill_mode="seg_padding_background", fill_color="black")

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_render(nat, B, S, N, steps, warmup, seed):
        """W untimed + K timed render steps bracketed by barrier + synchronize; returns (max-over-ranks seconds, own seconds,
        the resident inputs)."""
        R = S * S
        # every rank renders its own images (shard by image, no data-path collective): different latents / poses per rank
        film = proc.film_params(spec, B, seed=seed + rank)
        tf = tuple(torch.as_tensor(film[k], device=dev) for k in ("freq_geo", "phase_geo", "freq_app", "phase_app"))
        torch.manual_seed(seed + 234 + rank)
        o, d, z, _, _ = VR.sample_rays(B, N, dev, 12, (S, S), 0.88, 1.12, 0.3, 0.155, np.pi / 2, np.pi / 2, "gaussian")
        u = torch.rand((B * R, N), device=dev)
        quiesce_gc()
        for _ in range(warmup):
            nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            nat.render(o, d, z, u, None, None, *tf, opts, hierarchical=True)
        barrier()
        own = time.perf_counter() - t0
        return fdist.max_over_ranks(own, device=dev), own, (o, d, z, tf)

    def gather_floats(v):
        if not use_dist:
            return [float(v)]
        t = torch.tensor([float(v)], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")     # gloo gathers host memory
        outs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return [float(x.item()) for x in outs]

    def roofline_of(nat, inputs, pts, precision, iters):
        o, d, z, tf = inputs
        k_ms = nat.time_siren_rays(o, d, z, *tf, iters=iters)       # dominant kernel alone, hipEvents on the launch stream
        achieved = pts * FLOP_PER_POINT / (k_ms * 1e-3) / 1e12
        if precision == "f32":
            peak, kname, mfma, extra = PEAK_FP32_MATRIX_TFLOPS, "siren_kernel<256, true, false>", "v_mfma_f32_32x32x2_f32 (exact fp32)", {}
        else:
            # every algorithmic product is evaluated as 3 fp16 MFMAs (wh*xh + wh*xl + wl*xh, fp32 accumulate): the attainable
            # ceiling of this algorithm on the fp16 pipe is peak/3; `frac` is quoted against the full dense fp16 peak.
            peak, kname, mfma = PEAK_F16_MATRIX_TFLOPS, native.forward_kernel_name(nat), "fp16 MFMA, 3 per product (hi/lo error compensation)"
            extra = {"frac_of_f16x3_ceiling": achieved / (peak / 3)}
        # what the matrix pipe really executes (static MFMA count of the kernel: 3 MFMAs per product at f16x3, label head folded)
        ex_flop = nat.executed_flop_per_point()
        extra.update({"flop_per_point_executed": ex_flop, "mfma_executed_tflops": pts * ex_flop / (k_ms * 1e-3) / 1e12,
                      "frac_executed": pts * ex_flop / (k_ms * 1e-3) / 1e12 / peak})
        # the power wall from inside the kernel: every workgroup stamps s_memtime / s_memrealtime at its first and
        # last instruction -> shader cycles per launch and the clock granted while it ran; the same cycles at the 2.4 GHz the peak is
        # quoted at give frac_at_spec_clock
        try:
            cp = nat.clock_probe(o, d, z, *tf, iters=iters)
            t_spec = cp["cycles_per_launch"] / (SPEC_CLOCK_GHZ * 1e9)
            extra.update({"cycles_per_launch": cp["cycles_per_launch"], "clock_ghz_effective": cp["clock_ghz"],
                          "kernel_ms_with_clock_stamps": cp["kernel_ms"], "spec_clock_ghz": SPEC_CLOCK_GHZ,
                          "kernel_ms_at_spec_clock": t_spec * 1e3, "frac_at_spec_clock": pts * FLOP_PER_POINT / t_spec / 1e12 / peak,
                          "wall_clock_khz": cp["wall_clock_khz"]})
        except Exception as e:
            extra["clock_probe_error"] = f"{type(e).__name__}: {e}"
        return {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "kernel": kname, "kernel_ms": k_ms, "points_per_launch": pts, "flop_per_point_algorithmic": FLOP_PER_POINT,
                "mfma": mfma, **extra}

    nat = native.NativeModel(sd, spec, dev, args.precision)
    B, S, N = args.batch, args.img_size, args.num_steps
    R = S * S
    dt, own, inputs = timed_render(nat, B, S, N, args.steps, args.warmup, 1000)
    value = world * B * R * args.steps / dt
    o_, d_, z_, tf_ = inputs
    with native.phase_timing() as one:          # launch groups of one render (every kernel launch of the library is one)
        nat.render(o_, d_, z_, torch.rand((B * R, N), device=dev), None, None, *tf_, opts, hierarchical=True)
    render_launches = sum(one.calls.values())
    per_rank = [B * R * args.steps / t for t in gather_floats(own)]
    roof = roofline_of(nat, inputs, B * R * N, args.precision, max(5, args.steps // 2))
    roof_frac_ranks = gather_floats(roof["frac"])
    sweep = None
    if not args.no_sweep64:
        # north_star: "rays/s on synthetic 64x64x24-sample batches at 1/2/4/8 GPUs ... as fraction of roofline"
        sB, sS, sN = 4, 64, 24
        sdt, _, sin = timed_render(nat, sB, sS, sN, args.steps, args.warmup, 2000)
        sroof = roofline_of(nat, sin, sB * sS * sS * sN, args.precision, max(5, args.steps // 2))
        sweep = {"workload": f"{sB} images/GPU x {sS}x{sS} rays x {sN}+{sN} samples", "value": world * sB * sS * sS * args.steps / sdt,
                 "unit": "rays/s", "ms_per_step": sdt / args.steps * 1e3, "roofline_frac": sroof["frac"],
                 "roofline_frac_of_f16x3_ceiling": sroof.get("frac_of_f16x3_ceiling"), "kernel_ms": sroof["kernel_ms"]}

    # the generator step with its DDP all-reduce: every rank takes part (north_star: "RCCL all-reduce of G/D grads over xGMI")
    ddp_legs = {}
    if not args.no_gstep_ddp:
        mor = lambda v: fdist.max_over_ranks(v, device=dev)

        def agree(ok):                           # "every rank says ok"
            if not use_dist:
                return bool(ok)
            t = torch.tensor([1.0 if ok else 0.0], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item() > 0.5)
        for key, b_, it, skip in (("gstep_ddp", args.batch, 12, False), ("gstep_ddp_b6", 6, 3, args.no_gstep_b6 or (B, S, N) != (1, 128, 24))):
            if skip:
                continue
            try:
                torch.cuda.empty_cache()
                ddp_legs[key] = gstep_ddp_leg(spec, sd, dev, rank, world, b_, S, N, args.precision, barrier, mor, iters=it, agree=agree)
            except Exception as e:               # an extra leg must never take the headline metric down with it
                ddp_legs[key] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

    if rank == 0:
        dtype = "f32" if args.precision == "f32" else "f16x3 (error-compensated fp16 MFMA, fp32 accumulate; fp32-class accuracy)"
        tr = pmc_traffic() if (args.precision == "f16x3" and (B, S, N) == (1, 128, 24)) else None
        out = {
            "metric": f"rays/s/GPU forward render ({S}x{S}, {N}+{N} samples, H=256 FiLM-SIREN + 32x96^3 grid)",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"configs[1]: CelebA_double_semantic_texture_embedding_256_dim_96 generator, {S}x{S}, "
                                   f"{N}+{N} hierarchical samples, batch {B}/GPU, forward-only render, procedural weights",
                       "img_size": S, "num_steps": N, "batch_per_gpu": B, "sharding": "by image, no collective"},
            "roofline": {**roof, "traffic": (tr["fetch_x2_bytes"] + tr["write_bytes"]) if tr else None, "traffic_detail": tr,
                         "frac_per_rank": roof_frac_ranks},
            "rays_per_s_per_gpu": value / world, "rays_per_s_per_rank": per_rank, "n_ranks_seen": n_ranks_seen,
            "timed_region": "fenerf_render_forward on rays resident in HBM (coarse SIREN, composite, resample, fine SIREN, merge + composite: "
                            f"{render_launches} kernel launches per step); the mapping networks, ray setup and the NCHW epilogue of a full "
                            "generator call (0.05-0.18 ms, tools/time_call_overhead.py) are outside it",
            "launches_per_step": render_launches,
            "launcher": ((f"torch.distributed.run, one rank per GPU, backend {'nccl (RCCL)' if args.dist_backend == 'nccl' else args.dist_backend}"
                          if not args.one_device else f"torch.distributed.run, {world} ranks on ONE device (cuda:0), backend {args.dist_backend}")
                         if "TORCHELASTIC_RUN_ID" in os.environ or world > 1
                         else "single process") + (" [process group forced at world 1]" if use_dist and world == 1 else ""),
            "dist_backend": dist.get_backend() if use_dist else None,
        }
        if sweep:
            out["sweep64"] = sweep
        out.update(ddp_legs)
        if "ms_no_ddp" in ddp_legs.get("gstep_ddp", {}):     # the reference's G step on one GPU without the wrapper: generator(z_geo, z_app) + backward
            out["gstep_z"] = {"ms": ddp_legs["gstep_ddp"]["ms_no_ddp"], "what": "generator(z_geo, z_app, **metadata) + backward on the bare module "
                              "(both mapping networks inside; = gstep_ddp.ms_no_ddp)"}
        if world == 1 and not args.no_f32 and args.precision != "f32":
            try:
                nat32 = native.NativeModel(sd, spec, dev, "f32")
                fdt, _, fin = timed_render(nat32, B, S, N, args.steps, args.warmup, 1000)
                out["f32"] = {"value": B * R * args.steps / fdt, "unit": "rays/s", "ms_per_step": fdt / args.steps * 1e3,
                              "dtype": "f32 (exact fp32 MFMA, v_mfma_f32_32x32x2_f32)",
                              "roofline": roofline_of(nat32, fin, B * R * N, "f32", max(5, args.steps // 2))}
                del nat32
            except Exception as e:
                out["f32"] = {"error": f"{type(e).__name__}: {e}"}
            # Round 5: the opt-in reduced-precision forwards (include/fenerf.h fenerf_model_set_forward_mode) beside the headline: the same
            # timed render, the same roofline fields (cycles per launch, clock granted) and what they cost in accuracy against the
            # headline's own pixels on the same rays.  Never the headline: they do not meet its asserted bounds (profiles/r05_*).
            for key, what in ((("f16x2", "two fp16 MFMAs per product everywhere (weights as ONE fp16 value: wl*xh dropped, the lo halves neither fetched nor read)"),) if args.dead_ends else ()) + (
                              ("f16x3c2", "three fp16 MFMAs per product through the geometry trunk and the label / sigma head (sigma and labels bit-identical to the headline's), two in the colour layers and the rgb head"),):
                try:
                    natx = native.NativeModel(sd, spec, dev, key)
                    xdt, _, xin = timed_render(natx, B, S, N, args.steps, args.warmup, 1000)
                    xroof = roofline_of(natx, xin, B * R * N, "f16x3", max(5, args.steps // 2))
                    xroof["mfma"] = what
                    xroof.pop("frac_of_f16x3_ceiling", None)
                    o2, d2, z2, tf2 = xin
                    u2 = torch.rand((B * R, N), device=dev)
                    ref_px, ref_dp, _, _ = nat.render(o2, d2, z2, u2, None, None, *tf2, opts, hierarchical=True)
                    got_px, got_dp, _, _ = natx.render(o2, d2, z2, u2, None, None, *tf2, opts, hierarchical=True)
                    err = (got_px - ref_px).abs().amax(-1)
                    out[key] = {"value": B * R * args.steps / xdt, "unit": "rays/s", "ms_per_step": xdt / args.steps * 1e3, "dtype": what,
                                "vs_headline": (B * R * args.steps / xdt) / value * world, "roofline": xroof,
                                "pixels_vs_headline": {"max_abs": float(err.max()), "mean_abs": float(err.mean()),
                                                       "rays_beyond_1e-3": int((err > 1e-3).sum()), "rays": int(err.numel()),
                                                       "depth_max_abs": float((got_dp - ref_dp).abs().max())}}
                    del natx
                except Exception as e:
                    out[key] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.precision == "f16x3" and args.dead_ends:
            # the same render as ONE launch (include/fenerf.h fenerf_set_render_fusion): reported beside the headline,
            # which takes the faster four-launch route
            try:
                with native.render_fusion("force"):
                    odt, _, _ = timed_render(nat, B, S, N, args.steps, args.warmup, 1000)
                    with native.phase_timing() as one1:
                        rgb1 = nat.render(o_, d_, z_, torch.ones((B * R, N), device=dev) * 0.5, None, None, *tf_, opts, hierarchical=True)[0]
                rgb4 = nat.render(o_, d_, z_, torch.ones((B * R, N), device=dev) * 0.5, None, None, *tf_, opts, hierarchical=True)[0]
                out["render_one_launch"] = {"value": B * R * args.steps / odt, "unit": "rays/s", "ms_per_step": odt / args.steps * 1e3,
                                            "launches_per_step": sum(one1.calls.values()), "launch_groups": dict(one1.calls),
                                            "bit_identical_to_the_four_launch_render": bool(torch.equal(rgb1, rgb4)),
                                            "vs_headline": (B * R * args.steps / odt) / value * world}
            except Exception as e:
                out["render_one_launch"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_gstep:
            try:
                out["gstep"] = gstep_leg(spec, sd, dev, B, S, N, args.precision)
            except Exception as e:          # the extra leg must never take the headline metric down with it
                out["gstep"] = {"error": f"{type(e).__name__}: {e}"}
            if args.precision == "f16x3":
                try:   # opt-in AMP-class weight-gradient operands (the class of the reference's own autocast training): reported beside, never instead
                    out["gstep_amp"] = gstep_leg(spec, sd, dev, B, S, N, args.precision, grad_precision="amp")
                except Exception as e:
                    out["gstep_amp"] = {"error": f"{type(e).__name__}: {e}"}
                for key in ("tape16", "amp16"):   # opt-in 16-bit tape (round 5): the tier between the default and AMP, and AMP on top of it; beside, never instead
                    try:
                        out["gstep_" + key] = gstep_leg(spec, sd, dev, B, S, N, args.precision, grad_precision=key)
                    except Exception as e:
                        out["gstep_" + key] = {"error": f"{type(e).__name__}: {e}"}
            try:   # opt-in exact sparsity of the backward (round 6): beside the dense default, never instead
                out["gstep_sparse"] = gstep_leg(spec, sd, dev, B, S, N, args.precision, breakdown=False, sparse=True)
            except Exception as e:
                out["gstep_sparse"] = {"error": f"{type(e).__name__}: {e}"}
            if not args.no_gstep_b6 and (B, S, N) == (1, 128, 24):
                try:   # BASELINE.json configs[2]: the reference's generator micro-batch (batch 24 split 4 -> 6 images of 128x128 x 24+24 per GPU)
                    torch.cuda.empty_cache()
                    out["gstep_b6"] = gstep_leg(spec, sd, dev, 6, S, N, args.precision, iters=5, breakdown=False, per_step_median=True)
                    out["gstep_b6"]["ms_per_image"] = out["gstep_b6"]["ms"] / 6
                    torch.cuda.empty_cache()
                except Exception as e:
                    out["gstep_b6"] = {"error": f"{type(e).__name__}: {e}"}
                if args.gstep_sparse_b6:
                    try:   # the same micro-batch with the opt-in exact-sparsity backward (opt-in leg: --gstep-sparse-b6)
                        out["gstep_sparse_b6"] = gstep_leg(spec, sd, dev, 6, S, N, args.precision, iters=5, breakdown=False, per_step_median=True,
                                                           sparse=True)
                        out["gstep_sparse_b6"]["ms_per_image"] = out["gstep_sparse_b6"]["ms"] / 6
                        torch.cuda.empty_cache()
                    except Exception as e:
                        out["gstep_sparse_b6"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline:
            # N = 1: BASELINE.md's plan (1 warm-up + 3 timed runs x 3 shapes); N > 1: the headline shape once, so that every line of
            # the driver's scaling sweep is self-contained while the other ranks wait at the final barrier for ~20 s, not a minute
            film1 = proc.film_params(spec, 1, seed=1000)
            out["cpu_baseline"] = cpu_baseline(spec, sd, film1, 7, full=(world == 1 and not args.quick_cpu_baseline))
        write_detail(out)
        print(compact_line(out), flush=True)        # the LAST stdout line, < 4 KB
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
