"""Thin callers of the generator API -- the library-level counterparts of the reference's inference scripts
(SURVEY §8f.2 / f.3).  They add nothing to the hot path: every render goes through generator.staged_forward* /
siren.forward_with_frequencies_phase_shifts, i.e. the HIP kernels.

reference: mask2color + COLOR_MAP  train_double_latent_semantic.py:35-72
           multi-view loop         render_multiview_images_double_semantic.py:24-85
           voxel-grid evaluation   extract_double_semantic_shapes.py:13-90
           inversion loop          inverse_render_double_semantic.py:306-410
           latent interpolation    render_video_interpolation_semantic.py:131-187
"""
import numpy as np
import torch

from . import native

# label id -> RGB (train_double_latent_semantic.py:35-55); configuration data of the dataset's 19 classes
COLOR_MAP = {
    0: [0, 0, 0], 1: [204, 0, 0], 2: [76, 153, 0], 3: [204, 204, 0], 4: [51, 51, 255], 5: [204, 0, 204], 6: [0, 255, 255],
    7: [255, 204, 204], 8: [102, 51, 0], 9: [255, 0, 0], 10: [102, 204, 0], 11: [255, 255, 0], 12: [0, 0, 153],
    13: [0, 0, 204], 14: [255, 51, 153], 15: [0, 204, 204], 16: [0, 51, 0], 17: [255, 153, 51], 18: [0, 204, 0],
}
_COLOR_LUT = np.zeros((max(COLOR_MAP) + 1, 3), dtype=np.float32)
for _k, _v in COLOR_MAP.items():
    _COLOR_LUT[_k] = _v


def mask2color(masks, device=None, scale=None):
    """[B,19,H,W] logits -> [B,3,H,W] float colours (0..255) on the CPU, like the reference: argmax over the label
    channels (first index wins ties) then the LUT (train_double_latent_semantic.py:66-72)."""
    # the label channels of an image that is already on the host (staged_forward's result) go through numpy: single-threaded, 4 ms per
    # 256 x 256 image wherever it runs -- torch's CPU argmax over a non-last dimension takes 9 ms with a sane thread count and 50-70 ms in
    # a container whose CPU quota is far below its logical CPU count (the GPU boxes of this project: 256 logical CPUs, 16 granted), which
    # made it the largest item of a 256 x 256 multi-view render (tools/exp/callers_timing.py).  numpy's argmax also returns the first maximum.
    # `device`: a GPU the caller has at hand -- the host image's label channels take the argmax there (0.5 ms round trip through pinned memory
    # instead of 4 ms; the device's argmax also returns the first maximum, NaNs included: tools/exp/argmax_ties_probe.py)
    if not masks.is_cuda and device is not None and torch.device(device).type == "cuda":
        masks = masks.to(device, non_blocking=True)
    if masks.is_cuda:
        idx = native.to_host(torch.argmax(masks, dim=1).to(torch.uint8)).numpy()
    else:
        idx = np.argmax(masks.detach().numpy(), axis=1)
    out = np.ascontiguousarray(_COLOR_LUT[idx].transpose(0, 3, 1, 2))
    if scale is not None:          # `mask2color(...) / 255.` of the reference's scripts, as one fp32 division per element here (numpy) instead of
        out = out / np.float32(scale)    # a torch CPU operation per view (5-15 ms each where the CPU quota is far below the CPU count)
    return torch.from_numpy(out)


def multiview_kwargs(curriculum, image_size=256, ray_step_multiplier=2, lock_view_dependence=False):
    """The kwargs bag render_multiview_images_double_semantic.py:43-54 builds from a curriculum."""
    c = dict(curriculum)
    c["num_steps"] = curriculum[0]["num_steps"] * ray_step_multiplier
    c["img_size"] = image_size
    c["psi"] = 0.7
    c["v_stddev"] = 0
    c["h_stddev"] = 0
    c["lock_view_dependence"] = lock_view_dependence
    c["last_back"] = False
    c["nerf_noise"] = 0
    return {k: v for k, v in c.items() if type(k) is str}


def load_generator(path, device, use_ema=True, reset_render_options=True):
    """What every inference script of the reference does first (render_multiview_images_double_semantic.py:58-66,
    render_video_interpolation_semantic.py:317-324): unpickle the generator module saved by the training loop
    (`torch.save(generator_ddp.module, 'generator.pth')`, train...py:524), unpickle the torch_ema object next to it
    (`<prefix>ema.pth`, the prefix being everything before 'generator' in the path) and copy the averaged weights in, then
    set_device + eval.  Pickles written by the reference resolve through compat.install_aliases(): its module paths
    (generators.generators.*, siren.siren.*) map to this package and torch_ema.ema.ExponentialMovingAverage to fenerf_amd.ema
    when torch_ema is not installed (same attribute layout: decay, num_updates, shadow_params, collected_params).
    `reset_render_options`: the multi-view and inversion scripts overwrite three attributes of the pickled module
    (softmax_label = False, neural_renderer_img / _seg = None; render_multiview...:60-62, inverse_render...), the video script keeps
    what the pickle says (render_video_interpolation_semantic.py:317-324) -- its front end passes False.  `use_ema=False` (the tools'
    --no_ema) renders the raw generator weights; the reference has no such switch and fails without the ema file."""
    import os
    from . import compat
    compat.install_aliases()
    device = torch.device(device)
    generator = torch.load(path, map_location=device, weights_only=False)
    if reset_render_options:
        generator.softmax_label = False
        generator.neural_renderer_img = None
        generator.neural_renderer_seg = None
    ema_file = path.split("generator")[0] + "ema.pth"
    if use_ema:
        if not os.path.exists(ema_file):
            raise FileNotFoundError(f"{ema_file}: the reference loads the EMA weights next to the generator pickle")
        ema = torch.load(ema_file, map_location=device, weights_only=False)
        ema.copy_to(generator.parameters())
    generator.set_device(device)
    generator.eval()
    return generator


def _latent_dims(generator, default=256):
    """(z_geo_dim, z_app_dim): the reference hard-codes 256 (render_multiview...:79-80); read from the module when it says otherwise."""
    return int(getattr(generator, "z_geo_dim", default)), int(getattr(generator, "z_app_dim", default))


def render_multiview(generator, curriculum, seed, device, face_angles=(-0.5, -0.25, 0.0, 0.25, 0.5), image_size=256,
                     ray_step_multiplier=2, lock_view_dependence=False, z_dim=None, latents=None):
    """Five yaw angles of one identity (render_multiview_images_double_semantic.py:66-85).
    -> (images [V,3,S,S] in [-1,1], segmaps [V,3,S,S] in [0,1]) on the CPU.
    latents: optional (z_geo, z_app) instead of the seeded draws (tests teacher-force the reference's CPU-generator values)."""
    kw = multiview_kwargs(curriculum, image_size, ray_step_multiplier, lock_view_dependence)
    h_mean = kw["h_mean"]
    zg_dim, za_dim = (z_dim, z_dim) if z_dim is not None else _latent_dims(generator)
    images, segmaps = [], []
    for a in face_angles:
        kw["h_mean"] = a + h_mean
        torch.manual_seed(seed)
        z_geo = torch.randn((1, zg_dim), device=device)
        z_app = torch.randn((1, za_dim), device=device)
        if latents is not None:
            z_geo, z_app = (torch.as_tensor(t, dtype=torch.float32, device=device) for t in latents)
        with torch.no_grad():
            img, _ = generator.staged_forward(z_geo, z_app, **kw)
        images.append(img[:, -3:])
        segmaps.append(mask2color(img[:, :-3], device, scale=255.0))
    return torch.cat(images), torch.cat(segmaps)


def create_samples(N=256, voxel_origin=(0, 0, 0), cube_length=2.0, device=None):
    """Voxel-centre coordinates [1, N^3, 3] exactly as the reference builds them (extract_double_semantic_shapes.py:13-35):
    note the y / x indices are (i / N) % N and (i / N / N) % N in FLOAT arithmetic (not floor-divided) -- kept as is.
    device: build them there (the same statements: int64 -> float conversion, IEEE division (by a tensor divisor), fmod, one multiply and one add per coordinate
    -- bit for bit the host's values, a GPU test compares) instead of on the host followed by a 200-MB pageable copy at N = 256."""
    voxel_origin = np.array(voxel_origin) - cube_length / 2
    voxel_size = cube_length / (N - 1)
    overall_index = torch.arange(0, N ** 3, 1, dtype=torch.long, device=device)
    samples = torch.zeros(N ** 3, 3, device=device)
    # (on a GPU torch divides by a Python scalar as a multiplication by its reciprocal -- not the host's correctly rounded quotient unless N
    # is a power of two; a tensor divisor takes the IEEE division: tools/exp/div_probe.py)
    Nf = N if samples.device.type == "cpu" else torch.tensor(float(N), device=samples.device)
    samples[:, 2] = overall_index % N
    samples[:, 1] = (overall_index.float() / Nf) % Nf
    samples[:, 0] = ((overall_index.float() / Nf) / Nf) % Nf
    samples[:, 0] = (samples[:, 0] * voxel_size) + voxel_origin[2]
    samples[:, 1] = (samples[:, 1] * voxel_size) + voxel_origin[1]
    samples[:, 2] = (samples[:, 2] * voxel_size) + voxel_origin[0]
    return samples.unsqueeze(0), voxel_origin, voxel_size


def sample_generator(generator, z_geo, z_app=None, max_batch=None, voxel_resolution=256, voxel_origin=(0, 0, 0), cube_length=2.0,
                     psi=0.5):
    """Density volume [N,N,N] (numpy) of one identity for marching cubes (extract_double_semantic_shapes.py:38-62):
    truncated FiLM parameters, view direction locked to (0,0,-1).  One fused kernel launch evaluates all N^3 points;
    `max_batch` is accepted for signature compatibility and ignored.  (The reference passes the same z to both mapping
    networks; pass z_app to differ.)"""
    z_app = z_geo if z_app is None else z_app
    samples, _, _ = create_samples(voxel_resolution, voxel_origin, cube_length, device=z_geo.device)
    avg_fg, avg_pg, avg_fa, avg_pa = generator.generate_avg_frequencies()
    with torch.no_grad():
        raw_fg, raw_pg = generator.siren.geo_mapping_network(z_geo)
        raw_fa, raw_pa = generator.siren.app_mapping_network(z_app)
        fg, pg = avg_fg + psi * (raw_fg - avg_fg), avg_pg + psi * (raw_pg - avg_pg)
        fa, pa = avg_fa + psi * (raw_fa - avg_fa), avg_pa + psi * (raw_pa - avg_pa)
        out = generator.siren.native(samples.device).siren_forward(samples, None, fg, pg, fa, pa)   # None = locked view dir
    return native.to_host(out[..., -1].reshape(voxel_resolution, voxel_resolution, voxel_resolution).contiguous()).numpy()


def film_from_inversion(meta, device=None):
    """(freq_geo, freq_app, phase_geo, phase_app) = mean + offset of an inversion checkpoint (the dict inverse_render returns / the
    reference's `freq_phase_offset_<name>.pth`), in forward_with_frequencies' argument order (extract_double_semantic_shapes.py:127-133,
    inverse_render_double_semantic.py:465-467)."""
    t = (lambda v: v.detach().to(device)) if device is not None else (lambda v: v.detach())
    return (t(meta["w_geo_frequencies"]) + t(meta["w_geo_frequency_offsets"]), t(meta["w_app_frequencies"]) + t(meta["w_app_frequency_offsets"]),
            t(meta["w_geo_phase_shifts"]) + t(meta["w_geo_phase_shift_offsets"]), t(meta["w_app_phase_shifts"]) + t(meta["w_app_phase_shift_offsets"]))


def sample_generator_wth_frequencies_phase_shifts(generator, meta, max_batch=None, voxel_resolution=256, voxel_origin=(0, 0, 0),
                                                  cube_length=2.0, psi=0.5):
    """Density volume [N,N,N] of an INVERTED identity (extract_double_semantic_shapes.py:65-87, the reference's spelling of the name):
    meta carries 'truncated_frequencies_geo' / '_app' and 'truncated_phase_shifts_geo' / '_app' (the script fills them with mean +
    offset of an inversion checkpoint, :127-133 -- film_from_inversion above); view direction locked to (0, 0, -1); `max_batch` and `psi`
    are accepted and ignored (the reference ignores psi here too)."""
    samples, _, _ = create_samples(voxel_resolution, voxel_origin, cube_length, device=generator.device)
    with torch.no_grad():
        out = generator.siren.native(samples.device).siren_forward(samples, None, meta["truncated_frequencies_geo"], meta["truncated_phase_shifts_geo"],
                                                                   meta["truncated_frequencies_app"], meta["truncated_phase_shifts_app"])
    return native.to_host(out[..., -1].reshape(voxel_resolution, voxel_resolution, voxel_resolution).contiguous()).numpy()


# ---- the inversion script's host pieces (inverse_render_double_semantic.py) ------------------------------------------------------------
COLOR_MAP_COMPLETE_KEYS = 19      # labels 0 (background) .. 18 of the script's COLOR_MAP_COMPLETE (:50-69); COLOR_MAP has 1 .. 18


def mask2labels(mask_np, n_labels=18):
    """One-hot [n_labels, H, W] float64 of a label image (:82-92): 19 labels -> channel i is label i; 18 -> channel i is label i + 1
    (background dropped)."""
    labels = np.zeros((n_labels, mask_np.shape[0], mask_np.shape[1]))
    for i in range(n_labels):
        labels[i][mask_np == (i if n_labels == 19 else i + 1)] = 1.0
    return labels


def mIOU(source, target):
    """(:122-126) mean over classes of |s * t| / (|s + t > 0| + 1e-6), per image"""
    return torch.mean(torch.div(torch.sum(source * target, dim=[2, 3]).float(), torch.sum((source + target) > 0, dim=[2, 3]).float() + 1e-6), dim=1)


def _pil_resize_shorter(img, size, resample):
    """torchvision.transforms.Resize(int) on a PIL image: the shorter side becomes `size`, the other int(size * long / short)"""
    w, h = img.size
    if (w <= h and w == size) or (h <= w and h == size):
        return img
    if w < h:
        return img.resize((size, int(size * h / w)), resample)
    return img.resize((int(size * w / h), size), resample)


def _pil_center_crop(img, size):
    """torchvision.transforms.CenterCrop(size) for an image at least that large: left/top = round((dim - size) / 2)"""
    w, h = img.size
    left, top = int(round((w - size) / 2.0)), int(round((h - size) / 2.0))
    return img.crop((left, top, left + size, top + size))


def _pil_to_tensor(img):
    """transforms.ToTensor(): [C, H, W] float in [0, 1] ('L' -> one channel)"""
    a = np.asarray(img, dtype=np.uint8)
    a = a[:, :, None] if a.ndim == 2 else a
    return torch.from_numpy(np.array(a.transpose(2, 0, 1))).float().div(255)


def inversion_targets(gt_image, gt_seg, image_size=256, no_center_crop=False, background_mask=False, white_background_mask=False):
    """The targets of one inversion from two PIL images -- an RGB photo and its 'L' label map (values 0 .. 18) -- as
    inverse_render_double_semantic.py:178-222, :290-327 builds them with torchvision transforms (restated on PIL directly: Resize(320,
    bilinear) -> CenterCrop(256) -> Resize((S, S), NEAREST) -> ToTensor [-> Normalize(0.5, 0.5)]; `no_center_crop`: the last resize only):
    -> gt_image [1,3,S,S] in [-1,1], gt_seg_18 [1,18,S,S] in {-1,1}, gt_seg_19 [1,19,256,256] in {0,1} (the mIoU target, always 256^2).
    background_mask / white_background_mask paint the photo 0 / 1 where the label map is 0 (:295-310)."""
    from PIL import Image
    gt_image, gt_seg = gt_image.convert("RGB"), gt_seg.convert("L")
    if background_mask or white_background_mask:
        i = _pil_to_tensor(gt_image)
        lab = _pil_to_tensor(gt_seg.resize(gt_image.size, resample=Image.NEAREST)) * 255.0
        i[lab.expand_as(i) == 0] = 0.0 if background_mask else 1.0
        gt_image = Image.fromarray(i.mul(255).byte().permute(1, 2, 0).numpy())     # transforms.ToPILImage() of a float tensor: mul(255).byte()

    def pipeline(img, size):
        if not no_center_crop:
            img = _pil_center_crop(_pil_resize_shorter(img, 320, Image.BILINEAR), 256)
        return _pil_to_tensor(img.resize((size, size), Image.NEAREST))

    image = (pipeline(gt_image, image_size) - 0.5) / 0.5
    seg18 = (torch.tensor(mask2labels((pipeline(gt_seg, image_size) * 255.0)[0].numpy(), 18), dtype=torch.float) - 0.5) / 0.5
    seg19 = torch.tensor(mask2labels((pipeline(gt_seg, 256) * 255.0)[0].numpy(), 19), dtype=torch.float)
    return image[None], seg18[None], seg19[None]


def inversion_options(image_size=256, fov=12, device=None):
    """The kwargs bag of the optimisation renders (:225-247): 24 coarse samples only, frontal camera, eval fill mode"""
    import math
    hv = torch.tensor(math.pi / 2)
    hv = hv.to(device) if device is not None else hv
    return {'img_size': image_size, 'fov': fov, 'ray_start': 0.88, 'ray_end': 1.12, 'num_steps': 24, 'h_stddev': 0, 'v_stddev': 0,
            'h_mean': hv, 'v_mean': hv.clone(), 'hierarchical_sample': False, 'sample_dist': None, 'clamp_mode': 'relu', 'nerf_noise': 0,
            'fade_steps': 10000, 'z_app_lambda': 0, 'z_geo_lambda': 0, 'pos_lambda': 0, 'tok_interval': 2000, 'tok_v': 0.6, 'betas': (0, 0.9),
            'fill_mode': 'eval_seg_padding_background'}


def inversion_render_options(fov=12, fill_color='black', img_size=256, num_steps=48):
    """The kwargs bag of the preview / reconstruction renders (:249-265): 256^2, 48 + 48 samples"""
    import math
    return {'img_size': img_size, 'fov': fov, 'ray_start': 0.88, 'ray_end': 1.12, 'num_steps': num_steps, 'h_stddev': 0, 'v_stddev': 0,
            'v_mean': math.pi / 2, 'hierarchical_sample': True, 'sample_dist': None, 'clamp_mode': 'relu', 'nerf_noise': 0, 'last_back': False,
            'fill_mode': 'eval_seg_padding_background', 'fill_color': fill_color}


def inversion_trajectory(name, num_frames, fov):
    """[(t, pitch, yaw, fov)] of the inversion script's set_trajectory (:504-570): the video script's trajectories plus 'inverse_sphere'
    and 'rotation_linear'; its 'zoom' runs t over linspace(-1, 1) (50 frames whatever num_frames says) with fov + 5 + 5 sin(2 pi t)."""
    pi = np.pi
    if name in ("front", "orbit", "non_rotation", "sphere", "rotation_horizontal"):
        return camera_trajectory(name, num_frames, fov)
    if name == "inverse_sphere":
        return [(t, 0.2 * (1 - np.cos(t * 2 * pi)) + pi / 2, 0.4 * np.sin(t * 2 * pi) + pi / 2, fov) for t in np.linspace(0, 1, num_frames)]
    if name == "zoom":
        return [(t, pi / 2, pi / 2, fov + 5 + np.sin(t * 2 * pi) * 5) for t in np.linspace(-1, 1)]
    if name == "rotation_linear":
        return [(t, pi / 2, pi / 2 + t, fov) for t in np.linspace(-0.4, 0.4, num_frames)]
    raise ValueError(f"unknown trajectory {name!r} (front | orbit | non_rotation | sphere | inverse_sphere | rotation_horizontal | zoom | rotation_linear)")


def render_inversion_views(generator, meta, render_options, angles=(0.0,), max_batch_size=2400000, lock_view_dependence=False):
    """staged_forward_with_frequencies of an inversion state at yaws pi/2 + angle (:419-431, :441-446) -> [(angle, frame [1,22,S,S])]"""
    import math
    film = film_from_inversion(meta)
    out = []
    with torch.no_grad():
        for angle in angles:
            img, _, _ = generator.staged_forward_with_frequencies(*film, h_mean=math.pi / 2 + angle, max_batch_size=max_batch_size,
                                                                  lock_view_dependence=lock_view_dependence, **render_options)
            out.append((angle, img))
    return out


def render_inversion_recon(generator, meta, render_options, trajectory, max_batch_size=2400000, lock_view_dependence=False):
    """The frames of run_render_recon_video (:462-501): per trajectory entry one staged render of the inverted identity at (pitch, yaw) --
    the trajectory's fov is NOT applied (the reference sets only h_mean / v_mean) -- as uint8 [S, 3 S, 3] RGB panels
    [image | label colours | 0.5 / 0.5 blend] (the reference hands them to cv2 as BGR)."""
    from . import imageio_lite
    film = film_from_inversion(meta)
    kw = dict(render_options)
    frames = []
    to_u8 = lambda t: imageio_lite.to_uint8_hwc(imageio_lite.make_grid(t, normalize=True))     # tensor_to_PIL (:114-119)
    with torch.no_grad():
        for _, pitch, yaw, _ in trajectory:
            kw.update(h_mean=float(yaw), v_mean=float(pitch))
            frame, _, _ = generator.staged_forward_with_frequencies(*film, max_batch_size=max_batch_size, lock_view_dependence=lock_view_dependence, **kw)
            image, sem = to_u8(frame[:, -3:].cpu()), to_u8(mask2color(frame[:, :-3], generator.device))
            blend = image * 0.5 + sem * 0.5
            frames.append(np.concatenate([image, sem, blend], axis=1).astype("uint8"))
    return frames


def inverse_render(generator, gt_image, gt_seg, options, n_iterations=700, init_psi=0.0, lambda_seg=1.0, lambda_img=1.0,
                   lambda_percept=0.0, lambda_norm=0.0, percept=None, z_dim=256, lr=1e-2, on_step=None, latent_noise=0.03,
                   n_mean_latents=10000, record_offsets=False):
    """GAN inversion in FiLM space (inverse_render_double_semantic.py:306-410): optimise additive offsets on the geometry /
    appearance frequencies and phase shifts with Adam (lr 1e-2, weight_decay 1e-4, StepLR(100, 0.75)) under annealed
    latent noise so that generator.forward_with_frequencies reproduces gt_image [1,3,S,S] and gt_seg [1,18,S,S] (both in
    [-1,1]).  Every iteration is a native differentiable render; when the generator's parameters do not require grad
    only the FiLM gradients are computed.  `percept` (e.g. an LPIPS module) is optional -- none is shipped.
    Every draw (the mean-latent blocks -- `n_mean_latents` = 10,000 in the script --, the random latents, the four noise tensors of an
    iteration) goes through generator.draws in the script's order: with the default source that is torch.randn / randn_like on the
    device, value for value the script's generator consumption; a test replays the draws the reference recorded.
    Returns a dict with the reference's checkpoint keys (w_*_frequencies, w_*_phase_shifts, w_*_offsets) + `losses`
    (+ `offset_history`: the four offset tensors after every iteration, on the host, if record_offsets)."""
    device = generator.device
    siren = generator.siren
    draws = generator.draws

    def init(mapping):     # :307-327
        z = draws.randn((n_mean_latents, z_dim), device)
        rand_z = draws.randn((1, z_dim), device)
        with torch.no_grad():
            freq, phase = mapping(z)
            rand_f, rand_p = mapping(rand_z)
        w_f, w_p = freq.mean(0, keepdim=True), phase.mean(0, keepdim=True)
        w_f = w_f + init_psi * (rand_f - w_f)
        w_p = w_p + init_psi * (rand_p - w_p)
        return w_f, w_p, torch.zeros_like(w_f).requires_grad_(), torch.zeros_like(w_p).requires_grad_()

    w_gf, w_gp, o_gf, o_gp = init(siren.geo_mapping_network)
    w_af, w_ap, o_af, o_ap = init(siren.app_mapping_network)
    if lambda_img == 0:        # :371-376
        opt_params = [o_gf, o_gp]
    elif lambda_seg == 0:
        opt_params = [o_af, o_ap]
    else:
        opt_params = [o_gf, o_gp, o_af, o_ap]
    optimizer = torch.optim.Adam(opt_params, lr=lr, weight_decay=1e-4)
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, 100, gamma=0.75)
    mse = torch.nn.MSELoss(reduction="mean")
    losses, history = [], []
    for i in range(n_iterations):
        k = (n_iterations - i) / n_iterations                     # annealed noise on the FiLM parameters (:381-384; 0.03 there)
        # draw order and arithmetic of the reference (:381-384): geo freq, geo phase, app freq, app phase; (0.03 * randn) * k
        n_gf, n_gp = latent_noise * draws.randn(tuple(w_gf.shape), device) * k, latent_noise * draws.randn(tuple(w_gp.shape), device) * k
        n_af, n_ap = latent_noise * draws.randn(tuple(w_af.shape), device) * k, latent_noise * draws.randn(tuple(w_ap.shape), device) * k
        frame, _ = generator.forward_with_frequencies(w_gf + n_gf + o_gf, w_af + n_af + o_af, w_gp + n_gp + o_gp, w_ap + n_ap + o_ap,
                                                      **options)
        loss = lambda_seg * mse(frame[:, :-3], gt_seg) + lambda_img * mse(frame[:, -3:], gt_image)
        if lambda_percept and percept is not None:
            loss = loss + lambda_percept * percept(frame[:, -3:], gt_image).sum()
        if lambda_norm > 0:
            loss = loss + lambda_norm * sum((o ** 2).mean() for o in (o_gf, o_gp, o_af, o_ap))
        loss.backward()
        optimizer.step()
        optimizer.zero_grad()
        scheduler.step()
        # the loss stays on the device unless a callback wants it now: reading it every iteration makes the host wait for the step it has
        # just enqueued, and the next iteration's launches then start from an idle device
        losses.append(loss.detach() if on_step is None else float(loss.detach()))
        if record_offsets:
            history.append(tuple(o.detach().cpu().clone() for o in (o_gf, o_gp, o_af, o_ap)))
        if on_step is not None:      # (i, loss, the reference's checkpoint dict at this iteration: what its in-loop preview renders use, :419-452)
            on_step(i, losses[-1], dict(w_geo_frequencies=w_gf, w_geo_phase_shifts=w_gp, w_app_frequencies=w_af, w_app_phase_shifts=w_ap,
                                        w_geo_frequency_offsets=o_gf.detach(), w_geo_phase_shift_offsets=o_gp.detach(),
                                        w_app_frequency_offsets=o_af.detach(), w_app_phase_shift_offsets=o_ap.detach()))
    if losses and on_step is None:
        losses = torch.stack(losses).cpu().tolist()
    extra = dict(offset_history=history) if record_offsets else {}
    return dict(**extra, w_geo_frequencies=w_gf, w_geo_phase_shifts=w_gp, w_app_frequencies=w_af, w_app_phase_shifts=w_ap,
                w_geo_frequency_offsets=o_gf.detach(), w_geo_phase_shift_offsets=o_gp.detach(),
                w_app_frequency_offsets=o_af.detach(), w_app_phase_shift_offsets=o_ap.detach(), losses=losses)


def render_latent_interpolation(generator, z1_geo, z2_geo, z1_app, z2_app, options, n_frames=8, latent_type="both", psi=1.0,
                                trajectory=None):
    """Frames of a walk between two identities in FiLM space (render_video_interpolation_semantic.py:131-187, :314-380):
    truncate both ends towards the mean FiLM parameters, interpolate the geometry and / or appearance side linearly
    ('geo' | 'app' | 'both' | 'non'; 'app' sweeps t over [-1, 1] like the reference), render every frame with
    staged_forward_with_frequencies.  trajectory: optional list of (pitch, yaw) per frame overriding v_mean / h_mean.
    -> (frames [F, C, S, S], depth [F, S, S]) on the CPU."""
    avg_fg, avg_pg, avg_fa, avg_pa = generator.generate_avg_frequencies()
    trunc = lambda avg, raw: avg + psi * (raw - avg)
    with torch.no_grad():
        ends = []
        for zg, za in ((z1_geo, z1_app), (z2_geo, z2_app)):
            fg, pg = generator.siren.geo_mapping_network(zg)
            fa, pa = generator.siren.app_mapping_network(za)
            ends.append((trunc(avg_fg, fg), trunc(avg_fa, fa), trunc(avg_pg, pg), trunc(avg_pa, pa)))
    lerp = lambda a, b, t: a * (1 - t) + b * t
    frames, depths = [], []
    for i, t in enumerate(np.linspace(0, 1, n_frames)):
        t = float(t)
        if latent_type == "app":
            t = (t - 0.5) * 2
        move_geo, move_app = latent_type in ("geo", "both"), latent_type in ("app", "both")
        (fg1, fa1, pg1, pa1), (fg2, fa2, pg2, pa2) = ends
        film = (lerp(fg1, fg2, t) if move_geo else fg1, lerp(fa1, fa2, t) if move_app else fa1,
                lerp(pg1, pg2, t) if move_geo else pg1, lerp(pa1, pa2, t) if move_app else pa1)
        kw = dict(options)
        if trajectory is not None:
            kw["v_mean"], kw["h_mean"] = trajectory[i]
        img, depth, _ = generator.staged_forward_with_frequencies(*film, **kw)
        frames.append(img)
        depths.append(depth)
    return torch.cat(frames), torch.cat(depths)


# ---------------------------------------------------------------------------------------------------
# render_video_interpolation_semantic.py: options bag and camera trajectories
# ---------------------------------------------------------------------------------------------------
def video_kwargs(curriculum, image_size=256, ray_step_multiplier=2, psi=0.5, lock_view_dependence=False, num_frames=36, fov=12,
                 fill_color="black"):
    """The kwargs bag render_video_interpolation_semantic.py:52-69 builds from a curriculum."""
    c = dict(curriculum)
    c["num_steps"] = curriculum[0]["num_steps"] * ray_step_multiplier
    c["img_size"] = image_size
    c["psi"] = psi
    c["v_stddev"] = 0
    c["h_stddev"] = 0
    c["lock_view_dependence"] = lock_view_dependence
    c["last_back"] = curriculum.get("eval_last_back", False)
    c["num_frames"] = num_frames
    c["nerf_noise"] = 0
    c["fov"] = fov
    c["fill_mode"] = curriculum.get("fill_mode", "weight")
    if c["fill_mode"] == "seg_padding_background":
        c["fill_mode"] = "eval_seg_padding_background"
    c["fill_color"] = fill_color
    return {k: v for k, v in c.items() if type(k) is str}


def camera_trajectory(name, num_frames, fov):
    """[(t, pitch, yaw, fov)] of run_video_double_latent_interpolation (render_video_interpolation_semantic.py:326-379)."""
    pi = np.pi
    if name == "front":
        return [(t, 0.2 * np.cos(t * 2 * pi) + pi / 2, 0.4 * np.sin(t * 2 * pi) + pi / 2, fov + 5 + np.sin(t * 2 * pi) * 5)
                for t in np.linspace(0, 1, num_frames, endpoint=True)]
    if name == "orbit":
        return [(t, pi / 2, t * 2 * pi, fov) for t in np.linspace(0, 0.5, num_frames, endpoint=True)]
    if name == "rotation_horizontal":
        return [(t, pi / 2, pi / 2 + t * 0.5, fov) for t in np.linspace(-1, 1, num_frames)]
    if name == "non_rotation":
        return [(t, pi / 2, pi / 2, fov) for t in np.linspace(0, 1, num_frames, endpoint=True)]
    if name == "sphere":
        return [(t, 0.2 * np.cos(t * 2 * pi) + pi / 2, 0.4 * np.sin(t * 2 * pi) + pi / 2, fov) for t in np.linspace(0, 1, num_frames, endpoint=True)]
    if name == "zoom":
        return [(t, pi / 2, pi / 2, fov + np.sin(t * 2 * pi) * 5) for t in np.linspace(0, 1, num_frames)]
    raise ValueError(f"unknown trajectory {name!r} (front | orbit | rotation_horizontal | non_rotation | sphere | zoom)")


def render_double_latent_video(generator, seed, options, trajectory, latent_type="geo", psi=0.5, device=None, latents=None):
    """The frame loop of run_video_double_latent_interpolation (:381-430): two identities from seeds `seed` and `seed + 1`
    (z_geo then z_app each), FiLM parameters truncated towards the mean and interpolated on the `latent_type` side
    (DoubleFrequencyInterpolator, :130-176: 'app' sweeps t over [-1, 1]), one staged_forward_with_frequencies per trajectory
    entry with v_mean / h_mean / fov from it.  -> dict(images [F,3,S,S], labels [F,3,S,S] (0..255 colours), acc [F,3,S,S],
    depth [F,S,S]) on the CPU.  latents: optional ((z_geo1, z_app1), (z_geo2, z_app2)) instead of the seeded draws."""
    device = torch.device(device) if device is not None else generator.device
    zg_dim, za_dim = _latent_dims(generator)
    if latents is None:
        torch.manual_seed(seed)
        z1 = (torch.randn(1, zg_dim, device=device), torch.randn(1, za_dim, device=device))
        torch.manual_seed(int(seed) + 1)
        z2 = (torch.randn(1, zg_dim, device=device), torch.randn(1, za_dim, device=device))
    else:
        z1, z2 = (tuple(torch.as_tensor(t, dtype=torch.float32, device=device) for t in pair) for pair in latents)
    avg_fg, avg_pg, avg_fa, avg_pa = generator.generate_avg_frequencies()
    trunc = lambda avg, raw: avg + psi * (raw - avg)
    with torch.no_grad():
        ends = []
        for zg, za in (z1, z2):
            fg, pg = generator.siren.geo_mapping_network(zg)
            fa, pa = generator.siren.app_mapping_network(za)
            ends.append((trunc(avg_fg, fg), trunc(avg_fa, fa), trunc(avg_pg, pg), trunc(avg_pa, pa)))
    lerp = lambda a, b, t: a * (1 - t) + b * t
    (fg1, fa1, pg1, pa1), (fg2, fa2, pg2, pa2) = ends
    move_geo, move_app = latent_type in ("geo", "both"), latent_type in ("app", "both")
    out = dict(images=[], labels=[], acc=[], depth=[])
    kw = {k: v for k, v in options.items() if k != "num_frames"}
    for t, pitch, yaw, fov in trajectory:
        t = float(t)
        if latent_type == "app":
            t = (t - 0.5) * 2
        film = (lerp(fg1, fg2, t) if move_geo else fg1, lerp(fa1, fa2, t) if move_app else fa1,
                lerp(pg1, pg2, t) if move_geo else pg1, lerp(pa1, pa2, t) if move_app else pa1)
        kw.update(h_mean=float(yaw), v_mean=float(pitch), fov=float(fov), h_stddev=0, v_stddev=0)
        frame, depth, weight_sum = generator.staged_forward_with_frequencies(*film, **kw)
        out["images"].append(frame[:, -3:])
        out["labels"].append(mask2color(frame[:, :-3], device))
        out["acc"].append(weight_sum[:, -3:])
        out["depth"].append(depth)
    return {k: torch.cat(v) for k, v in out.items()}


def camera_trajectory_single(name, num_frames, fov):
    """[(t, pitch, yaw, fov)] of run_video_latent_interpolation, the single-latent variant (render_video_interpolation_semantic.py:
    197-262) -- its 'orbit' and 'rotation_horizontal' differ from the double-latent function's, and it has 'rotation_angles' /
    'rotation_pi' instead of 'zoom'."""
    pi = np.pi
    lin = lambda a, b, n: np.linspace(a, b, n)
    if name == "front":
        return [(t, 0.2 * np.cos(t * 2 * pi) + pi / 2, 0.4 * np.sin(t * 2 * pi) + pi / 2, fov + 5 + np.sin(t * 2 * pi) * 5) for t in lin(0, 1, num_frames)]
    if name == "orbit":
        return [(t, 0.2 * np.cos(t * 2 * pi) + pi / 4, t * 2 * pi, fov) for t in lin(0, 1, num_frames)]
    if name == "rotation_horizontal":
        return [(t, pi / 2, pi / 2 + t * 0.5, fov) for t in list(lin(-1, 1, num_frames // 2)) + list(lin(1, -1, num_frames // 2))]
    if name == "rotation_angles":
        return [(t, pi / 2, pi / 2 + a, fov) for t, a in enumerate([-0.5, -0.25, 0.0, 0.25, 0.5])]
    if name == "rotation_pi":
        return [(t, pi / 2, pi / 2 + t * 0.5 * pi, fov) for t in lin(-1, 1, num_frames)]
    if name == "non_rotation":
        return [(t, pi / 2, pi / 2, fov) for t in lin(0, 1, num_frames)]
    if name == "sphere":
        return [(t, 0.2 * np.cos(t * 2 * pi) + pi / 2, 0.4 * np.sin(t * 2 * pi) + pi / 2, fov) for t in lin(0, 1, num_frames)]
    raise ValueError(f"unknown trajectory {name!r} (front | orbit | rotation_horizontal | rotation_angles | rotation_pi | non_rotation | sphere)")


def render_latent_video(generator, seed, options, trajectory, latent_type="geo", psi=0.5, device=None, latents=None):
    """The frame loop of run_video_latent_interpolation (:264-299) for a single-latent ImplicitGenerator3d: z_current, z_next drawn
    back to back after torch.manual_seed(seed), FiLM parameters truncated towards the mean (FrequencyInterpolator, :109-128) and
    interpolated with the trajectory's t (latent_type 'non': the first identity throughout), one
    staged_forward_with_frequencies per trajectory entry.  -> dict(images [F,C,S,S] on the device, depth [F,S,S] on the CPU)."""
    device = torch.device(device) if device is not None else generator.device
    z_dim = int(getattr(generator, "z_dim", 256))
    if latents is None:
        torch.manual_seed(seed)
        z1 = torch.randn(1, z_dim, device=device)
        z2 = torch.randn(1, z_dim, device=device)
    else:
        z1, z2 = (torch.as_tensor(t, dtype=torch.float32, device=device) for t in latents)
    avg_f, avg_p = generator.generate_avg_frequencies()
    with torch.no_grad():
        f1, p1 = generator.siren.mapping_network(z1)
        f2, p2 = generator.siren.mapping_network(z2)
    f1, p1, f2, p2 = avg_f + psi * (f1 - avg_f), avg_p + psi * (p1 - avg_p), avg_f + psi * (f2 - avg_f), avg_p + psi * (p2 - avg_p)
    out = dict(images=[], depth=[])
    kw = {k: v for k, v in options.items() if k != "num_frames"}
    for t, pitch, yaw, fov in trajectory:
        t = float(t)
        film = (f1, p1) if latent_type == "non" else (f1 * (1 - t) + f2 * t, p1 * (1 - t) + p2 * t)
        kw.update(h_mean=float(yaw), v_mean=float(pitch), fov=float(fov), h_stddev=0, v_stddev=0)
        frame, depth = generator.staged_forward_with_frequencies(*film, **kw)
        out["images"].append(frame)
        out["depth"].append(depth)
    return {k: torch.cat(v) for k, v in out.items()}



# ------------------------------------------------------------------------------------------------------------------------------------
# The image dump of the reference's FID evaluation (fid_evaluation.py:96-150), which is how its forward path runs on several GPUs:
# every rank renders batches of 4 identities with staged_forward and writes the images whose ids are rank, rank + world, ... -- no
# data-path collective (the FID statistics themselves are third-party code on the dumped files and are not part of this package).
# ------------------------------------------------------------------------------------------------------------------------------------
def fid_dump_metadata(input_metadata, img_size=128, batch_size=4):
    """The option bag both dump loops build from the training step's metadata (fid_evaluation.py:97-106, :127-135): 128 x 128, batches
    of 4, the *_eval pose spread / distribution when the curriculum has them, psi = 1."""
    import copy
    metadata = copy.deepcopy(dict(input_metadata))
    metadata['img_size'] = img_size
    metadata['batch_size'] = batch_size
    metadata['h_stddev'] = metadata.get('h_stddev_eval', metadata['h_stddev'])
    metadata['v_stddev'] = metadata.get('v_stddev_eval', metadata['v_stddev'])
    metadata['sample_dist'] = metadata.get('sample_dist_eval', metadata['sample_dist'])
    metadata['psi'] = 1
    return metadata


class _OrderedWriter:
    """save(img, path) calls on ONE worker thread, in submission order: the JPEG encoding of a batch runs while the device renders the next
    one (the reference writes between renders, fid_evaluation.py:115-121; same files, same order).  An exception of `save` is raised by
    the next submit() or by close()."""

    def __init__(self, save):
        from concurrent.futures import ThreadPoolExecutor
        self.save, self.pool, self.pending = save, ThreadPoolExecutor(max_workers=1), []

    def _reap(self, wait):
        while self.pending and (wait or self.pending[0].done()):
            self.pending.pop(0).result()

    def submit(self, img, path):
        self._reap(False)
        self.pending.append(self.pool.submit(self.save, img, path))

    def close(self):
        try:
            self._reap(True)
        finally:
            self.pool.shutdown(wait=True)


def _dump_loop(generator, metadata, rank, world_size, output_dir, num_imgs, draw, save):
    import os
    from . import imageio_lite
    module = getattr(generator, "module", generator)        # the reference passes the DistributedDataParallel wrapper
    os.makedirs(output_dir, exist_ok=True)
    save = save or (lambda img, path: imageio_lite.save_image(img, path, normalize=True, value_range=(-1, 1)))
    generator.eval()
    img_counter = rank
    written = []
    writer = _OrderedWriter(save)
    try:
        with torch.no_grad():
            while img_counter < num_imgs:
                generated_imgs = draw(module, metadata)
                for img in generated_imgs:          # (a batch is written out whole: the last one may run past num_imgs, as in the reference)
                    if img.shape[0] != 3:
                        img = img[-3:]
                    path = os.path.join(output_dir, f'{img_counter:0>5}.jpg')
                    writer.submit(img, path)
                    written.append(path)
                    img_counter += world_size
    finally:
        writer.close()
    return written


def output_images(generator, input_metadata, rank, world_size, output_dir, num_imgs=2048, save=None):
    """fid_evaluation.output_images (:96-123) for the single-latent generators: z ~ randn [4, z_dim] per batch,
    staged_forward(z, **metadata)[0], image ids rank, rank + world_size, ... as `<id:05>.jpg` normalised from [-1, 1].
    -> the paths this rank wrote.  `save(img [3,S,S], path)` replaces the writer (tests)."""
    metadata = fid_dump_metadata(input_metadata)

    def draw(module, md):
        z = torch.randn((md['batch_size'], module.z_dim), device=module.device)
        return module.staged_forward(z, **md)[0]
    return _dump_loop(generator, metadata, rank, world_size, output_dir, num_imgs, draw, save)


def output_images_double(generator, input_metadata, rank, world_size, output_dir, num_imgs=2048, save=None):
    """fid_evaluation.output_images_double (:126-150) for the two-latent generators: z_geo then z_app ~ randn [4, dim] per batch,
    staged_forward(z_geo, z_app, **metadata)[0], the last three channels (rgb) of every image, ids rank, rank + world_size, ..."""
    metadata = fid_dump_metadata(input_metadata)

    def draw(module, md):
        z_geo = torch.randn((md['batch_size'], module.z_geo_dim), device=module.device)
        z_app = torch.randn((md['batch_size'], module.z_app_dim), device=module.device)
        return module.staged_forward(z_geo, z_app, **md)[0]
    return _dump_loop(generator, metadata, rank, world_size, output_dir, num_imgs, draw, save)


def eval_metrics_images(generator, curriculum, output_dir, num_images=2048, max_batch_size=94800000, save=None):
    """The image loop of the reference's eval_metrics.py (:41-52; single-latent generators): render options = the curriculum stage of
    `generator.step` with img_size 128, psi 1, last_back = eval_last_back, nerf_noise 0; one identity per call, z ~ randn [1, latent_dim],
    staged_forward(z, max_batch_size=..., **options)[0] written as `<i:05>.jpg` normalised from [-1, 1].  (The script then hands the
    directory to torch_fidelity -- third-party, not part of this package.)  -> the paths written."""
    import os
    from . import curriculums as _cur, imageio_lite
    options = _cur.extract_metadata(curriculum, generator.step)
    options['img_size'] = 128
    options['psi'] = 1
    options['last_back'] = options.get('eval_last_back', False)
    options['nerf_noise'] = 0
    os.makedirs(output_dir, exist_ok=True)
    save = save or (lambda img, path: imageio_lite.save_image(img, path, normalize=True, value_range=(-1, 1)))
    generator.eval()
    written = []
    writer = _OrderedWriter(save)
    try:
        for img_counter in range(num_images):
            z = torch.randn(1, options['latent_dim'], device=generator.device)
            with torch.no_grad():
                img = generator.staged_forward(z, max_batch_size=max_batch_size, **options)[0].to(generator.device)
            path = os.path.join(output_dir, f'{img_counter:0>5}.jpg')
            writer.submit(img, path)
            written.append(path)
    finally:
        writer.close()
    return written


def training_snapshots(generator, ema, fixed_z_geo, fixed_z_app, metadata, output_dir, step, device=None):
    """The sample-image block of the reference's training loop (train_double_latent_semantic.py:464-523), which is how staged_forward is
    called DURING training: under autocast, 25 fixed identities at 128 x 128 with the pose spread at zero, frontal and tilted
    (h_mean + 0.5), with the live weights and again with the EMA weights swapped in (store / copy_to / eval ... restore), plus 25 fresh
    identities at psi = 0.7 under the EMA weights.  Writes `<step>_{seg,img}_{fixed,tilted,fixed_ema,tilted_ema,random}.png`
    (5 x 5 grids, normalised over the batch like torchvision's save_image(normalize=True)) -> their paths.
    `generator`: the module or its DistributedDataParallel wrapper; `ema`: a torch_ema.ExponentialMovingAverage (or fenerf_amd.ema's)."""
    import copy
    import os
    from . import imageio_lite
    module = getattr(generator, "module", generator)
    device = module.device if device is None else device
    os.makedirs(output_dir, exist_ok=True)
    written = []

    def dump(tag, z_geo, z_app, **overrides):
        with torch.no_grad():
            with torch.autocast("cuda", enabled=torch.device(device).type == "cuda"):
                copied_metadata = copy.deepcopy(metadata)
                copied_metadata['h_stddev'] = copied_metadata['v_stddev'] = 0
                copied_metadata['img_size'] = 128
                for k, v in overrides.items():
                    copied_metadata[k] = copied_metadata[k] + v if k == 'h_mean' else v
                gen_imgs = module.staged_forward(z_geo.to(device), z_app.to(device), **copied_metadata)[0]
                gen_labels = mask2color(gen_imgs[:, :-3], device)
        for kind, t in (("seg", gen_labels[:25]), ("img", gen_imgs[:25, -3:])):
            path = os.path.join(output_dir, f"{step}_{kind}_{tag}.png")
            imageio_lite.save_image(t, path, nrow=5, normalize=True)
            written.append(path)

    generator.eval()
    dump("fixed", fixed_z_geo, fixed_z_app)
    dump("tilted", fixed_z_geo, fixed_z_app, h_mean=0.5)
    ema.store(generator.parameters())
    ema.copy_to(generator.parameters())
    generator.eval()
    dump("fixed_ema", fixed_z_geo, fixed_z_app)
    dump("tilted_ema", fixed_z_geo, fixed_z_app, h_mean=0.5)
    dump("random", torch.randn_like(fixed_z_geo), torch.randn_like(fixed_z_app), psi=0.7)
    ema.restore(generator.parameters())
    return written
