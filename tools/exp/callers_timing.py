"""End-to-end time of the inference callers against the time of the SIREN launches in them (torch.profiler): where a caller loses time
outside the kernels.  python tools/exp/callers_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from fenerf_amd import callers, procedural as proc

dev = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
gen, cur, curriculums = bench.curriculum_generator(spec, sd, dev, "f16x3")
gen.eval()


def timed(name, fn, reps=3):
    for _ in range(3): fn()          # (the first ~10 images of a process run 1.5 x slow: tools/exp/staged_segments.py)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type.name == "CUDA" and e.device_time_total > 0]
    siren = sum(e.device_time_total for e in ev if "siren" in e.name) / 1e3
    dev_all = sum(e.device_time_total for e in ev) / 1e3
    cpu = sorted([e for e in prof.events() if e.device_type.name == "CPU"], key=lambda e: -e.self_cpu_time_total)[:5]
    print(f"{name}: {min(ts):.1f} .. {max(ts):.1f} ms per call; device {dev_all:.1f} ms of which SIREN launches {siren:.1f} ms; "
          f"top self-CPU: {[(e.name[:32], round(e.self_cpu_time_total / 1e3, 1)) for e in cpu]}", flush=True)


timed("render_multiview 5 views 256x256x48+48", lambda: callers.render_multiview(gen, cur, 0, dev))
opts = callers.video_kwargs(cur, image_size=256, num_frames=8)
traj = callers.camera_trajectory("front", 8, 12)
timed("render_double_latent_video 8 frames 256x256x48+48", lambda: callers.render_double_latent_video(gen, 0, opts, traj, latent_type="geo", psi=0.5, device=dev))
z = torch.randn(1, 256, device=dev)
timed("sample_generator 128^3", lambda: callers.sample_generator(gen, z, voxel_resolution=128))
timed("sample_generator 256^3", lambda: callers.sample_generator(gen, z, voxel_resolution=256), reps=2)
