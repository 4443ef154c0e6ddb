"""The FID image dump (callers.output_images_double, fid_evaluation.py:126-150) end to end: ms per image incl. the JPEG writes, against the
device time in it.   python tools/exp/dump_timing.py [images]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from fenerf_amd import callers, procedural as proc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
spec = proc.model_spec("texture", hidden_dim=256, grid_size=96)
sd = proc.make_state_dict(spec, seed=0, sigma_gain=2000.0, with_mapping=False)
gen, cur, curriculums = bench.curriculum_generator(spec, sd, dev, "f16x3")
md = {**curriculums.extract_metadata(cur, 60000), "nerf_noise": 0}      # (the training loop adds nerf_noise before it calls the dump)
out = tempfile.mkdtemp()
callers.output_images_double(gen, md, 0, 1, out, num_imgs=8)
torch.cuda.synchronize()
t0 = time.perf_counter()
callers.output_images_double(gen, md, 0, 1, out, num_imgs=n)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    callers.output_images_double(gen, md, 0, 1, out, num_imgs=16)
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type.name == "CUDA" and e.device_time_total > 0]
devms = sum(e.device_time_total for e in ev) / 1e3
cpu = sorted([e for e in prof.events() if e.device_type.name == "CPU"], key=lambda e: -e.self_cpu_time_total)[:6]
tot = {}
for e in prof.events():
    if e.device_type.name == "CPU":
        a = tot.setdefault(e.name[:40], [0.0, 0]); a[0] += e.cpu_time_total / 1e3; a[1] += 1
print("CPU time by op (total, calls) over 16 images:", sorted(((k, round(v[0], 1), v[1]) for k, v in tot.items()), key=lambda t: -t[1])[:14])
ev_sorted = sorted(ev, key=lambda e: e.time_range.start)
gaps = []
for a, b in zip(ev_sorted, ev_sorted[1:]):
    g = b.time_range.start - a.time_range.end
    if g > 300: gaps.append((round(g / 1e3, 2), b.name[:50]))
print("device idle gaps > 0.3 ms:", gaps[:16])
print(f"output_images_double, {n} images of 128 x 128 x 24+24 in batches of 4: {ms / n:.2f} ms per image end to end; device {devms / 16:.2f} ms per image; "
      f"top self-CPU of 16 images: {[(e.name[:30], round(e.self_cpu_time_total / 1e3, 1)) for e in cpu]}")
t0 = time.perf_counter()
img = torch.rand(3, 128, 128) * 2 - 1
from fenerf_amd import imageio_lite
for i in range(32): imageio_lite.save_image(img, os.path.join(out, "x.jpg"), normalize=True, value_range=(-1, 1))
print(f"imageio_lite.save_image of one 128 x 128 image: {(time.perf_counter() - t0) / 32 * 1e3:.2f} ms")
shutil.rmtree(out)
