"""CPU tests of the inference front ends' host pieces: the argparse surfaces mirror the reference scripts, the
torchvision-free grid / PNG / AVI writers, the options bags and camera trajectories, and a torch_ema-style pickle loading
through the import aliases."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from fenerf_amd import callers, curriculums, imageio_lite  # noqa: E402


def test_make_grid_layout_and_normalisation_like_torchvision():
    imgs = torch.arange(3 * 3 * 2 * 2, dtype=torch.float32).reshape(3, 3, 2, 2)
    g = imageio_lite.make_grid(imgs, nrow=2, padding=2, pad_value=0.0)
    assert tuple(g.shape) == (3, 2 * 4 + 2, 2 * 4 + 2)                      # ymaps * (H + pad) + pad, xmaps * (W + pad) + pad
    assert torch.equal(g[:, 2:4, 2:4], imgs[0]) and torch.equal(g[:, 2:4, 6:8], imgs[1]) and torch.equal(g[:, 6:8, 2:4], imgs[2])
    assert (g[:, 6:8, 6:8] == 0).all() and (g[:, :2] == 0).all()             # empty cell and border are pad_value
    # normalize with a value range: clamp, then (x - low) / (high - low); without: min / max of the whole batch
    x = torch.tensor([[[[-2.0, -1.0], [0.0, 1.0]]]])
    n = imageio_lite.make_grid(x, normalize=True, value_range=(-1, 1))
    assert tuple(n.shape) == (3, 2, 2)                                       # single image: no padding, 1 -> 3 channels
    np.testing.assert_allclose(n[0].numpy(), [[0.0, 0.0], [0.5, 1.0]], atol=1e-6)
    n2 = imageio_lite.make_grid(x, normalize=True)
    np.testing.assert_allclose(n2[0].numpy(), [[0.0, 1 / 3], [2 / 3, 1.0]], atol=1e-6)
    assert imageio_lite.to_uint8_hwc(torch.full((3, 1, 1), 0.5)).tolist() == [[[128, 128, 128]]]     # x * 255 + 0.5, truncated


def test_save_image_png_round_trip(tmp_path):
    from PIL import Image
    imgs = torch.rand(5, 3, 8, 8) * 2 - 1
    arr = imageio_lite.save_image(imgs, str(tmp_path / "g.png"), normalize=True, value_range=(-1, 1))
    back = np.asarray(Image.open(tmp_path / "g.png"))
    assert back.shape == (8 + 4, 5 * 10 + 2, 3) and np.array_equal(back, arr)
    expect = ((imgs[3].clamp(-1, 1) + 1) / 2 * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    assert np.array_equal(back[2:10, 3 * 10 + 2:3 * 10 + 10], expect)


def test_avi_writer_container(tmp_path):
    w = imageio_lite.AviWriter(str(tmp_path / "v.avi"), fps=25)
    frames = [np.random.default_rng(i).integers(0, 255, (6, 10, 3), dtype=np.uint8) for i in range(4)]
    for f in frames:
        w.write(f)
    w.release()
    raw = open(tmp_path / "v.avi", "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"AVI " and struct.unpack("<I", raw[4:8])[0] == len(raw) - 8
    assert raw.count(b"00db") == 2 * 4                                      # four frame chunks + four index entries
    p = raw.index(b"movi") + 4
    assert raw[p:p + 4] == b"00db"
    size = struct.unpack("<I", raw[p + 4:p + 8])[0]
    assert size == 6 * 32                                                    # rows padded to 4 bytes: 10 * 3 = 30 -> 32
    first = np.frombuffer(raw[p + 8:p + 8 + size], np.uint8).reshape(6, 32)[:, :30].reshape(6, 10, 3)
    assert np.array_equal(first[::-1, :, ::-1], frames[0])                   # bottom-up BGR


def test_argparse_surfaces_mirror_the_reference_scripts():
    import render_multiview
    import render_video_interpolation
    o = render_multiview.build_parser().parse_args(["ckpt/generator.pth", "--seeds", "0", "7", "--output_dir", "o", "--max_batch_size", "99",
                                                    "--lock_view_dependence", "--image_size", "128", "--ray_step_multiplier", "3",
                                                    "--curriculum", "CelebA_double_semantic"])
    assert (o.path, o.seeds, o.output_dir, o.max_batch_size, o.lock_view_dependence, o.image_size, o.ray_step_multiplier, o.curriculum) == \
        ("ckpt/generator.pth", ["0", "7"], "o", 99, True, 128, 3, "CelebA_double_semantic")
    d = render_multiview.build_parser().parse_args(["g.pth"])
    assert (d.seeds, d.output_dir, d.max_batch_size, d.image_size, d.ray_step_multiplier, d.curriculum) == ([0], "imgs", 2400000, 256, 2, "CelebA")
    v = render_video_interpolation.build_parser().parse_args(["g.pth"])
    assert (v.interpolation_type, v.latent_type, v.seeds, v.output_dir, v.batch_size, v.max_batch_size, v.depth_map, v.lock_view_dependence,
            v.image_size, v.ray_step_multiplier, v.num_frames, v.curriculum, v.trajectory, v.psi, v.fill_color, v.fov, v.save_with_video,
            v.save_with_latent, v.seed_mode) == \
        ("video_double_latent_interpolation", "geo", [0], "vids", 1, 2400000, False, False, 256, 2, 36, "CelebA", "front", 0.5, "black", 12,
         False, False, "single")
    cur = render_multiview.resolve_curriculum("CelebA_double_semantic_texture_embedding_256_dim_96")
    assert cur[0]["num_steps"] == curriculums.CelebA_double_semantic_texture_embedding_256_dim_96[0]["num_steps"]


def test_options_bags_and_trajectories():
    cur = curriculums.CelebA_double_semantic_texture_embedding_256_dim_96
    mv = callers.multiview_kwargs(cur, image_size=256, ray_step_multiplier=2)
    assert mv["num_steps"] == 2 * cur[0]["num_steps"] and mv["psi"] == 0.7 and mv["h_stddev"] == 0 and mv["last_back"] is False
    assert all(type(k) is str for k in mv)
    vk = callers.video_kwargs(cur, image_size=64, ray_step_multiplier=1, psi=0.5, num_frames=6, fov=12, fill_color="white")
    assert vk["fill_mode"] == "eval_seg_padding_background" and vk["fill_color"] == "white" and vk["num_frames"] == 6
    assert vk["last_back"] == cur.get("eval_last_back", False)
    tr = callers.camera_trajectory("front", 5, 12)
    assert len(tr) == 5 and tr[0][0] == 0 and tr[-1][0] == 1
    np.testing.assert_allclose(tr[0][1:], (0.2 + np.pi / 2, np.pi / 2, 17.0), atol=1e-12)        # pitch, yaw, fov at t = 0
    orbit = callers.camera_trajectory("orbit", 3, 12)
    np.testing.assert_allclose([y for _, _, y, _ in orbit], [0, np.pi / 2, np.pi], atol=1e-12)
    assert [round(f, 6) for *_, f in callers.camera_trajectory("zoom", 5, 12)] == [12, 17, 12, 7, 12]
    with pytest.raises(ValueError):
        callers.camera_trajectory("nope", 3, 12)
    # the single-latent script's own trajectories (render_video_interpolation_semantic.py:197-262)
    so = callers.camera_trajectory_single("orbit", 5, 12)
    np.testing.assert_allclose([y for _, _, y, _ in so], np.linspace(0, 2 * np.pi, 5), atol=1e-12)
    np.testing.assert_allclose(so[0][1], 0.2 + np.pi / 4, atol=1e-12)
    rh = callers.camera_trajectory_single("rotation_horizontal", 7, 12)          # two sweeps of num_frames // 2
    assert len(rh) == 6 and [round(t, 6) for t, *_ in rh] == [-1, 0, 1, 1, 0, -1]
    ra = callers.camera_trajectory_single("rotation_angles", 99, 12)
    np.testing.assert_allclose([y - np.pi / 2 for _, _, y, _ in ra], [-0.5, -0.25, 0, 0.25, 0.5], atol=1e-12)
    rp = callers.camera_trajectory_single("rotation_pi", 3, 12)
    np.testing.assert_allclose([y for _, _, y, _ in rp], [0, np.pi / 2, np.pi], atol=1e-12)
    assert callers.camera_trajectory_single("front", 4, 12) == callers.camera_trajectory("front", 4, 12)
    with pytest.raises(ValueError):
        callers.camera_trajectory_single("zoom", 3, 12)


def test_torch_ema_style_pickle_loads_through_the_aliases(tmp_path):
    """A pickle whose class path is torch_ema.ema.ExponentialMovingAverage with torch_ema 0.2's attribute layout (decay,
    num_updates, shadow_params, collected_params -- what `torch.save(ema, 'ema.pth')` wrote in the reference's environment,
    train_double_latent_semantic.py:525) unpickles, in a process that has no torch_ema, into fenerf_amd.ema and copy_to works."""
    code = r'''
import sys, types, torch
sys.path.insert(0, %r)
m = types.ModuleType("torch_ema"); e = types.ModuleType("torch_ema.ema")
class ExponentialMovingAverage:                 # the writer's class: attribute layout of torch_ema 0.2
    def __init__(self, parameters, decay):
        self.decay, self.num_updates = decay, 3
        self.shadow_params = [p.clone().detach() + 1.0 for p in parameters if p.requires_grad]
        self.collected_params = []
ExponentialMovingAverage.__module__ = "torch_ema.ema"
e.ExponentialMovingAverage = ExponentialMovingAverage; m.ema = e; m.ExponentialMovingAverage = ExponentialMovingAverage
sys.modules["torch_ema"], sys.modules["torch_ema.ema"] = m, e
lin = torch.nn.Linear(3, 2)
torch.save(ExponentialMovingAverage(lin.parameters(), 0.999), %r)
torch.save(lin.state_dict(), %r)
''' % (ROOT, str(tmp_path / "ema.pth"), str(tmp_path / "lin.pth"))
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).returncode == 0
    code2 = r'''
import sys, torch
sys.path.insert(0, %r)
from fenerf_amd import compat
compat.install_aliases()
ema = torch.load(%r, weights_only=False)
assert type(ema).__module__ == "fenerf_amd.ema", type(ema)
lin = torch.nn.Linear(3, 2); lin.load_state_dict(torch.load(%r))
before = [p.clone() for p in lin.parameters()]
ema.copy_to(lin.parameters())
assert all(torch.equal(p, b + 1.0) for p, b in zip(lin.parameters(), before))
assert ema.decay == 0.999 and ema.num_updates == 3
ema.store(lin.parameters()); ema.update(lin.parameters()); ema.restore(lin.parameters())
print("ok")
''' % (ROOT, str(tmp_path / "ema.pth"), str(tmp_path / "lin.pth"))
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-1500:]


def test_inversion_and_shape_front_ends_mirror_the_reference_scripts():
    """tools/inverse_render.py / tools/extract_shapes.py: the argparse surfaces of inverse_render_double_semantic.py:132-169 and
    extract_double_semantic_shapes.py:90-97, option for option and default for default."""
    import extract_shapes
    import inverse_render
    d = inverse_render.build_parser().parse_args(["debug", "ckpt/7000_generator.pth"])
    assert (d.name, d.generator_path, d.image_path, d.seg_path, d.save_dir, d.load_checkpoint, d.seeds, d.init_seed, d.image_size, d.fov,
            d.num_frames, d.max_batch_size, d.lock_view_dependence, d.iteration, d.background_mask, d.white_background_mask, d.inverse_type,
            d.img_loss, d.seg_loss, d.lambda_img, d.lambda_seg, d.lambda_percept, d.lambda_norm, d.latent_normalize, d.latent_type, d.psi,
            d.init_psi, d.trajectory, d.depth_map, d.save_with_video, d.recon, d.fill_color, d.no_center_crop, d.checkpoint_path) == \
        ("debug", "ckpt/7000_generator.pth", None, None, None, False, [0], 0, 256, 12, 100, 2400000, False, 1000, False, False, "semantic",
         "mse", "mse", 0.0, 0.0, 0.0, 1.0, False, "app", 0, 0, "front", False, False, False, "black", False, "")
    o = inverse_render.build_parser().parse_args(["n", "g.pth", "--image_path", "a.jpg", "--seg_path", "a.png", "--save_dir", "o", "--iteration", "7",
                                                  "--lambda_seg", "1", "--lambda_img", "0.5", "--latent_normalize", "--recon", "--trajectory",
                                                  "rotation_linear", "--fill_color", "white", "--no_center_crop", "--white_background_mask"])
    assert (o.iteration, o.lambda_seg, o.lambda_img, o.latent_normalize, o.recon, o.trajectory, o.fill_color, o.no_center_crop,
            o.white_background_mask) == (7, 1.0, 0.5, True, True, "rotation_linear", "white", True, True)
    assert inverse_render.PREVIEW_ANGLES == (-0.5, -0.4, -0.3, -0.2, -0.1, 0, 0.1, 0.2, 0.3, 0.4, 0.5)
    s = extract_shapes.build_parser().parse_args(["g.pth"])
    assert (s.path, s.seeds, s.cube_size, s.voxel_resolution, s.output_dir, s.latent_path) == ("g.pth", [3, 4, 5], 0.3, 256, "shapes", None)
    s = extract_shapes.build_parser().parse_args(["g.pth", "--seeds", "9", "--cube_size", "0.5", "--voxel_resolution", "64", "--output_dir", "x",
                                                  "--latent_path", "m.pth"])
    assert (s.seeds, s.cube_size, s.voxel_resolution, s.output_dir, s.latent_path) == (["9"], 0.5, 64, "x", "m.pth")


def test_mrc_writer_round_trip_and_header(tmp_path):
    """imageio_lite.write_mrc: the mode-2 MRC2014 map the reference's shape scripts write through mrcfile (extract_double_semantic_shapes.py:
    120-121): header words at their MRC2014 byte offsets, data in [nz][ny][nx] order."""
    rng = np.random.default_rng(3)
    vol = rng.normal(size=(3, 4, 5)).astype(np.float32)
    p = str(tmp_path / "v.mrc")
    imageio_lite.write_mrc(p, vol)
    raw = open(p, "rb").read()
    assert len(raw) == 1024 + vol.size * 4
    assert struct.unpack_from("<4i", raw, 0) == (5, 4, 3, 2)                              # nx, ny, nz, mode
    assert struct.unpack_from("<3i", raw, 28) == (5, 4, 3) and struct.unpack_from("<3i", raw, 64) == (1, 2, 3)
    assert struct.unpack_from("<3f", raw, 52) == (90.0, 90.0, 90.0) and struct.unpack_from("<i", raw, 88)[0] == 1
    assert raw[208:212] == b"MAP " and raw[212:214] == b"\x44\x44" and struct.unpack_from("<i", raw, 108)[0] == 20140
    back, h = imageio_lite.read_mrc(p)
    assert np.array_equal(back, vol) and back.dtype == np.float32
    np.testing.assert_allclose([h["dmin"], h["dmax"], h["dmean"], h["rms"]], [vol.min(), vol.max(), vol.mean(), vol.std()], rtol=1e-6)
    np.testing.assert_array_equal(np.frombuffer(raw, "<f4", 5, 1024), vol[0, 0])           # x runs fastest
    with pytest.raises(ValueError):
        imageio_lite.write_mrc(p, np.zeros((4, 4), np.float32))


def test_inversion_targets_options_and_trajectories():
    """callers.inversion_targets restates the script's torchvision pipelines on PIL (inverse_render_double_semantic.py:178-222, :290-327);
    inversion_options / inversion_render_options are its two kwargs bags (:225-265); inversion_trajectory its set_trajectory (:504-570)."""
    from PIL import Image
    # a 400 x 300 photo of a constant colour and a label map of four constant quadrants: Resize(320) -> 426 x 320 (int(320 * 400 / 300)),
    # CenterCrop(256) at left = round(170 / 2) = 85, top = 32, then NEAREST to S x S
    photo = Image.fromarray(np.full((300, 400, 3), (255, 128, 0), np.uint8))
    lab = np.zeros((300, 400), np.uint8)
    lab[:150, :200], lab[:150, 200:], lab[150:, :200], lab[150:, 200:] = 0, 1, 5, 18
    seg = Image.fromarray(lab, "L")
    img, s18, s19 = callers.inversion_targets(photo, seg, image_size=16)
    assert tuple(img.shape) == (1, 3, 16, 16) and tuple(s18.shape) == (1, 18, 16, 16) and tuple(s19.shape) == (1, 19, 256, 256)
    np.testing.assert_allclose(img[0, :, 3, 3].numpy(), [1.0, 128 / 255 * 2 - 1, -1.0], atol=1e-6)      # Normalize(0.5, 0.5)
    assert set(np.unique(s18.numpy())) == {-1.0, 1.0} and set(np.unique(s19.numpy())) == {0.0, 1.0}
    # quadrant interiors (the bilinear Resize(320) only blends at the quadrant borders): label 1 -> channel 0 of 18 / channel 1 of 19
    assert s18[0, 0, 2, 12] == 1 and s18[0, 4, 12, 2] == 1 and s18[0, 17, 12, 12] == 1 and (s18[0, :, 2, 2] == -1).all()      # background: no channel
    assert s19[0, 0, 20, 20] == 1 and s19[0, 1, 20, 200] == 1 and s19[0, 5, 200, 20] == 1 and s19[0, 18, 200, 200] == 1
    assert (s19.sum(1)[0, 20:100, 20:100] == 1).all()
    # --no_center_crop: the last resize only; --white_background_mask paints the photo 1 where the label map is 0
    img2, s18b, _ = callers.inversion_targets(photo, seg, image_size=8, no_center_crop=True, white_background_mask=True)
    assert tuple(img2.shape) == (1, 3, 8, 8) and (img2[0, :, 1, 1] == 1.0).all() and img2[0, 2, 6, 6] == -1.0
    assert s18b[0, 4, 6, 1] == 1
    img3, _, _ = callers.inversion_targets(photo, seg, image_size=8, no_center_crop=True, background_mask=True)
    assert (img3[0, :, 1, 1] == -1.0).all()
    # helpers
    m = np.array([[0, 1], [18, 5]])
    l18, l19 = callers.mask2labels(m, 18), callers.mask2labels(m, 19)
    assert l18.shape == (18, 2, 2) and l18[0, 0, 1] == 1 and l18[17, 1, 0] == 1 and l18[:, 0, 0].sum() == 0
    assert l19[0, 0, 0] == 1 and l19[18, 1, 0] == 1 and l19.sum() == 4
    a = torch.tensor(l19[None], dtype=torch.float)
    np.testing.assert_allclose(callers.mIOU(a, a).item(), 4 / 19, atol=1e-6)               # four classes present with IoU 1, fifteen empty with 0
    # options bags
    o = callers.inversion_options(64, 12)
    assert (o["img_size"], o["num_steps"], o["hierarchical_sample"], o["fill_mode"], o["nerf_noise"], o["h_stddev"]) == \
        (64, 24, False, "eval_seg_padding_background", 0, 0) and abs(float(o["h_mean"]) - np.pi / 2) < 1e-6
    r = callers.inversion_render_options(12, "white")
    assert (r["img_size"], r["num_steps"], r["hierarchical_sample"], r["last_back"], r["fill_color"], r["fill_mode"]) == \
        (256, 48, True, False, "white", "eval_seg_padding_background") and "h_mean" not in r
    # trajectories
    t = callers.inversion_trajectory("inverse_sphere", 5, 12)
    np.testing.assert_allclose(t[0][1:], (np.pi / 2, np.pi / 2, 12), atol=1e-12)
    np.testing.assert_allclose(t[2][1], 0.4 + np.pi / 2, atol=1e-12)                       # t = 0.5: 0.2 * (1 - cos(pi))
    z = callers.inversion_trajectory("zoom", 7, 12)
    assert len(z) == 50 and abs(z[0][3] - 17) < 1e-9                                        # linspace(-1, 1) has 50 points whatever num_frames says
    rl = callers.inversion_trajectory("rotation_linear", 3, 12)
    np.testing.assert_allclose([y for _, _, y, _ in rl], [np.pi / 2 - 0.4, np.pi / 2, np.pi / 2 + 0.4], atol=1e-12)
    assert callers.inversion_trajectory("front", 4, 12) == callers.camera_trajectory("front", 4, 12)
    with pytest.raises(ValueError):
        callers.inversion_trajectory("nope", 3, 12)
    # film_from_inversion: mean + offset in forward_with_frequencies' argument order
    meta = {k: torch.full((1, 4), float(i)) for i, k in enumerate(
        ("w_geo_frequencies", "w_geo_phase_shifts", "w_app_frequencies", "w_app_phase_shifts", "w_geo_frequency_offsets",
         "w_geo_phase_shift_offsets", "w_app_frequency_offsets", "w_app_phase_shift_offsets"))}
    fg, fa, pg, pa = callers.film_from_inversion(meta)
    assert (float(fg[0, 0]), float(fa[0, 0]), float(pg[0, 0]), float(pa[0, 0])) == (0 + 4, 2 + 6, 1 + 5, 3 + 7)


# ---- the FID image dump (fid_evaluation.py:96-150): how the reference runs its forward path on several GPUs ---------------------------
class _StandInDouble:
    """a two-latent generator's surface as the dump loops use it (device, z_geo_dim / z_app_dim, eval, staged_forward)"""
    z_geo_dim, z_app_dim = 5, 7

    def __init__(self):
        self.device, self.training, self.calls = torch.device("cpu"), True, []

    def eval(self):
        self.training = False
        return self

    def staged_forward(self, z_geo, z_app, **md):
        self.calls.append((z_geo.clone(), z_app.clone(), dict(md)))
        B, S = z_geo.shape[0], 6
        base = (z_geo.sum(1) + 2 * z_app.sum(1)).reshape(B, 1, 1, 1)
        return torch.tanh(base + torch.arange(21 * S * S, dtype=torch.float32).reshape(1, 21, S, S) / (21 * S * S)), torch.zeros(B, S, S)


class _StandInSingle(_StandInDouble):
    z_dim = 4

    def staged_forward(self, z, **md):
        self.calls.append((z.clone(), dict(md)))
        B, S = z.shape[0], 6
        return torch.tanh(z.sum(1).reshape(B, 1, 1, 1) + torch.zeros(B, 3, S, S)), torch.zeros(B, S, S), torch.zeros(B, S, S)


class _DDPLike:
    def __init__(self, module):
        self.module = module

    def eval(self):
        self.module.eval()
        return self


_DUMP_MD = dict(img_size=32, batch_size=24, h_stddev=0.3, v_stddev=0.155, h_stddev_eval=0.2, sample_dist='gaussian', sample_dist_eval='uniform',
                psi=0.7, num_steps=24, fov=12, clamp_mode='relu', nerf_noise=0.0)


def _run_dump(fn, gen, rank, world, num_imgs, tmp, seed=100):
    torch.manual_seed(seed + rank)
    got = []
    paths = fn(_DDPLike(gen), _DUMP_MD, rank, world, str(tmp), num_imgs=num_imgs, save=lambda img, path: got.append((os.path.basename(path), img.clone())))
    assert [os.path.basename(p) for p in paths] == [n for n, _ in got]
    return got


@pytest.mark.parametrize("world", [1, 2, 3])
def test_fid_image_dump_shards_by_image_id(tmp_path, world):
    """callers.output_images_double / output_images: every rank writes the ids rank, rank + world, ... (whole batches of 4, so the last
    batch may run past num_imgs, as in the reference); the option bag is the reference's (128 x 128, batch 4, *_eval overrides, psi 1);
    draws are z_geo then z_app per batch; rgb = the last three channels; the caller's metadata is not modified."""
    num = 10
    ids = set()
    for rank in range(world):
        gen = _StandInDouble()
        got = _run_dump(callers.output_images_double, gen, rank, world, num, tmp_path)
        mine = [int(n.split(".")[0]) for n, _ in got]
        assert mine == list(range(rank, rank + world * len(mine), world)) and all(n.endswith(".jpg") and len(n) == 9 for n, _ in got)
        assert len(mine) % 4 == 0 and mine[-4] < num <= mine[-1] + world      # the last batch started below num_imgs and ended at or past it
        ids |= set(mine)
        assert not gen.training
        torch.manual_seed(100 + rank)
        for k, (zg, za, md) in enumerate(gen.calls):
            assert torch.equal(zg, torch.randn(4, 5)) and torch.equal(za, torch.randn(4, 7))
            assert md["img_size"] == 128 and md["batch_size"] == 4 and md["psi"] == 1 and md["h_stddev"] == 0.2 and md["v_stddev"] == 0.155
            assert md["sample_dist"] == "uniform" and md["num_steps"] == 24
            ref = gen.staged_forward(zg, za)[0]
            gen.calls.pop()
            for j in range(4):
                assert got[4 * k + j][1].shape == (3, 6, 6) and torch.equal(got[4 * k + j][1], ref[j, -3:])
    assert set(range(num)) <= ids
    assert _DUMP_MD["img_size"] == 32 and _DUMP_MD["psi"] == 0.7
    one = _StandInSingle()
    got = _run_dump(callers.output_images, one, 0, 1, 5, tmp_path)
    assert [n for n, _ in got] == [f"{i:05d}.jpg" for i in range(8)] and len(one.calls) == 2 and one.calls[0][0].shape == (4, 4)
    # the default writer: a JPEG through Pillow, normalised from [-1, 1]
    from PIL import Image
    paths = callers.output_images(_StandInSingle(), _DUMP_MD, 0, 1, str(tmp_path / "jpg"), num_imgs=1)
    assert len(paths) == 4 and np.asarray(Image.open(paths[0])).shape == (6, 6, 3)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference only exists in the build container")
@pytest.mark.parametrize("world", [1, 2])
def test_fid_image_dump_equals_the_references_own_loops(tmp_path, world):
    """fid_evaluation.output_images / output_images_double themselves (AST-extracted from the reference's file -- importing it needs
    torchvision and pytorch_fid --, executed with a recording save_image) on the same stand-in generator and seeds: same file names, same
    tensors, same calls."""
    import ast
    import copy
    src = open("/root/reference/fid_evaluation.py").read()
    fns = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in ("output_images", "output_images_double")]
    assert len(fns) == 2
    rec = []

    class _Bar:
        def __init__(self, *a, **k): pass
        def update(self, n): pass
        def close(self): pass

    def save_image(img, path, normalize=False, range=None):
        assert normalize is True and tuple(range) == (-1, 1)
        rec.append((os.path.basename(path), img.clone()))
    ns = dict(copy=copy, os=os, torch=torch, tqdm=_Bar, save_image=save_image)
    exec(compile(ast.Module(body=fns, type_ignores=[]), "fid_evaluation.py", "exec"), ns)
    for name, stand_in in (("output_images_double", _StandInDouble), ("output_images", _StandInSingle)):
        for rank in range(world):
            gen_ref, gen_own = stand_in(), stand_in()
            rec.clear()
            torch.manual_seed(7 + rank)
            ns[name](_DDPLike(gen_ref), _DUMP_MD, rank, world, str(tmp_path), num_imgs=9)
            theirs = list(rec)
            ours = _run_dump(getattr(callers, name), gen_own, rank, world, 9, tmp_path, seed=7)
            assert [n for n, _ in ours] == [n for n, _ in theirs] and all(torch.equal(a[1], b[1]) for a, b in zip(ours, theirs))
            assert len(gen_ref.calls) == len(gen_own.calls)
            for a, b in zip(gen_ref.calls, gen_own.calls):
                assert all(torch.equal(x, y) for x, y in zip(a[:-1], b[:-1])) and a[-1] == b[-1]


def test_dump_images_front_end_arguments():
    import dump_images
    opt = dump_images.build_parser().parse_args(["ckpt/generator.pth", "--curriculum", "CelebA", "--num_imgs", "64", "--output_dir", "o", "--one_device",
                                                 "--dist_backend", "gloo", "--seed", "3"])
    assert opt.path == "ckpt/generator.pth" and opt.num_imgs == 64 and opt.one_device and opt.dist_backend == "gloo" and opt.step == 100000 and opt.seed == 3


def test_eval_metrics_image_loop_options_and_draws(tmp_path):
    """callers.eval_metrics_images = the loop of the reference's eval_metrics.py:41-52: the stage of generator.step, 128 x 128, psi 1,
    last_back = eval_last_back, nerf_noise 0, one randn [1, latent_dim] per image, max_batch_size handed through."""
    class Gen(_StandInSingle):
        step = 25000
    gen = Gen()
    cur = {0: dict(img_size=32, num_steps=12), 20000: dict(img_size=64, num_steps=24), 'latent_dim': 4, 'eval_last_back': True, 'last_back': False,
           'fov': 12, 'psi': 0.5, 'nerf_noise': 1.0, 'clamp_mode': 'relu'}
    got = []
    torch.manual_seed(3)
    paths = callers.eval_metrics_images(gen, cur, str(tmp_path / "e"), num_images=3, max_batch_size=777, save=lambda img, path: got.append((os.path.basename(path), img.clone())))
    assert [os.path.basename(p) for p in paths] == ["00000.jpg", "00001.jpg", "00002.jpg"] and len(gen.calls) == 3 and not gen.training
    torch.manual_seed(3)
    for z, md in gen.calls:
        assert torch.equal(z, torch.randn(1, 4))
        assert md["img_size"] == 128 and md["num_steps"] == 24 and md["psi"] == 1 and md["last_back"] is True and md["nerf_noise"] == 0 and md["max_batch_size"] == 777
    assert got[0][1].shape == (1, 3, 6, 6) and cur["psi"] == 0.5 and cur[20000]["img_size"] == 64
