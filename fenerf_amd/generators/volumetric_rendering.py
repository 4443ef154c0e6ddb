"""Ray generation / camera sampling (host side, PyTorch on the render device) and the reference-named
entry points for compositing and resampling, which run as HIP kernels through the C-ABI.

reference: generators/volumetric_rendering.py -- get_initial_rays_trig :109-131, perturb_points :133-139,
transform_sampled_points :142-168, sample_camera_positions :179-228, create_cam2world_matrix :230-248,
fancy_integration :18-106, sample_pdf :259-300.

All random draws happen HERE (or in generators.py), with torch, in the reference's order;
the kernels receive them as inputs.  `draws` lets tests teacher-force recorded reference draws.
"""
import math
import random

import numpy as np
import torch

from .. import _lib, native
from .math_utils_torch import normalize_vecs


class TorchDraws:
    """Default random source: torch.rand / torch.randn on the render device (same call order and shapes
    as the reference, so a seeded run consumes the generator identically)."""

    def rand(self, shape, device):
        return torch.rand(shape, device=device)

    def randn(self, shape, device):
        return torch.randn(shape, device=device)

    def randperm(self, n, device):
        return torch.randperm(n, device=device)

    def normal_candidates(self, shape, device):
        """truncated_normal_'s candidate block: an uninitialised tensor filled in place, like the reference (:172)"""
        return torch.empty(shape, device=device).normal_()

    def coin(self):
        """the host-RNG coin of the 'hybrid' camera mode (random.random(), :196)"""
        return random.random()


class RecordedDraws:
    """Replays a list of recorded arrays (tests): each call pops the next one and checks its shape."""

    def __init__(self, arrays):
        self.arrays = list(arrays)

    def _next(self, shape, device):
        a = self.arrays.pop(0)
        t = torch.as_tensor(np.asarray(a), dtype=torch.float32, device=device)
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t

    def randperm(self, n, device):
        a = self.arrays.pop(0)
        t = torch.as_tensor(np.asarray(a), dtype=torch.long, device=device)
        assert tuple(t.shape) == (n,), (tuple(t.shape), n)
        return t

    rand = _next
    randn = _next
    normal_candidates = _next

    def coin(self):
        return float(np.asarray(self.arrays.pop(0)).reshape(()))


_DEFAULT_DRAWS = TorchDraws()


def _camera_ray_dirs(resolution, fov, device):
    """Unit camera-space directions [R,3] of the W x H pixel grid: x in [-1,1] left->right, y in [1,-1] top->bottom,
    z = -1/tan(fov/2); ray index = row*W + col (volumetric_rendering.py:113-121)."""
    W, H = resolution
    xs = torch.linspace(-1, 1, W, device=device)
    ys = torch.linspace(1, -1, H, device=device)
    x = xs.repeat(H)                       # col varies fastest
    y = ys.repeat_interleave(W)
    z = -torch.ones_like(x) / np.tan((2 * math.pi * fov / 360) / 2)
    return normalize_vecs(torch.stack([x, y, z], -1))


def get_initial_rays_trig(n, num_steps, device, fov, resolution, ray_start, ray_end):
    """Camera-space sample points [n,R,N,3], z_vals [n,R,N,1] and ray directions [n,R,3] (volumetric_rendering.py:109-131)."""
    d_cam = _camera_ray_dirs(resolution, fov, device)
    depths = torch.linspace(ray_start, ray_end, num_steps, device=device)
    z_vals = depths.reshape(1, num_steps, 1).repeat(d_cam.shape[0], 1, 1)
    points = d_cam.unsqueeze(1) * z_vals
    batch = lambda t: t.unsqueeze(0).repeat(n, *([1] * t.dim()))
    return batch(points), batch(z_vals), batch(d_cam)


def perturb_points(points, z_vals, ray_directions, device, draws=_DEFAULT_DRAWS):
    """Stratified jitter: every sample moves by (U[0,1) - 0.5) * bin width along its ray (:133-139)."""
    bin_width = z_vals[:, :, 1:2, :] - z_vals[:, :, 0:1, :]
    shift = (draws.rand(z_vals.shape, device) - 0.5) * bin_width
    return points + shift * ray_directions.unsqueeze(2), z_vals + shift


def truncated_normal_(tensor, mean=0, std=1, draws=_DEFAULT_DRAWS):
    """In place: N(mean, std) truncated to +-2 std by taking the first of 4 candidate draws that falls inside (:170-177)."""
    cand = draws.normal_candidates(tuple(tensor.shape) + (4,), tensor.device)
    first_ok = ((cand < 2) & (cand > -2)).max(-1, keepdim=True)[1]
    tensor.data.copy_(cand.gather(-1, first_ok).squeeze(-1))
    tensor.data.mul_(std).add_(mean)
    return tensor


def sample_camera_angles(device, n, horizontal_stddev, vertical_stddev, horizontal_mean, vertical_mean, mode, draws=_DEFAULT_DRAWS):
    """The random part of sample_camera_positions (:188-218): (theta, phi) [n,1] BEFORE the phi clamp, same draws/order."""
    if mode == "uniform":
        theta = (draws.rand((n, 1), device) - 0.5) * 2 * horizontal_stddev + horizontal_mean
        phi = (draws.rand((n, 1), device) - 0.5) * 2 * vertical_stddev + vertical_mean
    elif mode == "normal" or mode == "gaussian":
        theta = draws.randn((n, 1), device) * horizontal_stddev + horizontal_mean
        phi = draws.randn((n, 1), device) * vertical_stddev + vertical_mean
    elif mode == "hybrid":
        if draws.coin() < 0.5:
            theta = (draws.rand((n, 1), device) - 0.5) * 2 * horizontal_stddev * 2 + horizontal_mean
            phi = (draws.rand((n, 1), device) - 0.5) * 2 * vertical_stddev * 2 + vertical_mean
        else:
            theta = draws.randn((n, 1), device) * horizontal_stddev + horizontal_mean
            phi = draws.randn((n, 1), device) * vertical_stddev + vertical_mean
    elif mode == "truncated_gaussian":
        theta = truncated_normal_(torch.zeros((n, 1), device=device), draws=draws) * horizontal_stddev + horizontal_mean
        phi = truncated_normal_(torch.zeros((n, 1), device=device), draws=draws) * vertical_stddev + vertical_mean
    elif mode == "spherical_uniform":
        theta = (draws.rand((n, 1), device) - .5) * 2 * horizontal_stddev + horizontal_mean
        v_stddev, v_mean = vertical_stddev / math.pi, vertical_mean / math.pi
        v = ((draws.rand((n, 1), device) - .5) * 2 * v_stddev + v_mean)
        v = torch.clamp(v, 1e-5, 1 - 1e-5)
        phi = torch.arccos(1 - 2 * v)
    else:  # just use the mean
        theta = torch.ones((n, 1), device=device, dtype=torch.float) * horizontal_mean
        phi = torch.ones((n, 1), device=device, dtype=torch.float) * vertical_mean
    return theta, phi


def sample_camera_positions(device, n=1, r=1, horizontal_stddev=1, vertical_stddev=1, horizontal_mean=math.pi * 0.5,
                            vertical_mean=math.pi * 0.5, mode="normal", draws=_DEFAULT_DRAWS):
    """Camera origins on a sphere of radius r; returns (origin [n,3], phi = pitch [n,1], theta = yaw [n,1])  (:179-228).
    phi is clamped to [1e-5, pi - 1e-5]; y is up: origin = r (sin phi cos theta, cos phi, sin phi sin theta)."""
    theta, phi = sample_camera_angles(device, n, horizontal_stddev, vertical_stddev, horizontal_mean, vertical_mean, mode, draws)
    phi = torch.clamp(phi, 1e-5, math.pi - 1e-5)
    origin = torch.cat([r * torch.sin(phi) * torch.cos(theta), r * torch.cos(phi), r * torch.sin(phi) * torch.sin(theta)], -1)
    return origin, phi, theta


def _lookat_rotation(forward_vector):
    """[n,3,3] rotation whose columns are (-left, up, -forward) for a camera looking along `forward`, world up = +y."""
    f = normalize_vecs(forward_vector)
    world_up = torch.zeros_like(f)
    world_up[:, 1] = 1
    left = normalize_vecs(torch.cross(world_up, f, dim=-1))
    up = normalize_vecs(torch.cross(f, left, dim=-1))
    return torch.stack((-left, up, -f), dim=-1)


def create_cam2world_matrix(forward_vector, origin, device=None):
    """4x4 camera-to-world matrices [n,4,4] = translate(origin) @ rotate(look-at)  (:230-248)."""
    n = forward_vector.shape[0]
    m = torch.zeros((n, 4, 4), device=forward_vector.device, dtype=forward_vector.dtype)
    m[:, :3, :3] = _lookat_rotation(forward_vector)
    m[:, :3, 3] = origin
    m[:, 3, 3] = 1
    return m


def transform_sampled_points(points, z_vals, ray_directions, device, h_stddev=1, v_stddev=1, h_mean=math.pi * 0.5,
                             v_mean=math.pi * 0.5, mode="normal", draws=_DEFAULT_DRAWS):
    """Reference-shaped API (:142-168): jitter the samples, draw a camera pose, map points / directions / origins to
    world space.  Returns (points [n,R,N,3], z_vals, dirs [n,R,3], origins [n,R,3], pitch, yaw).  Draw order: jitter,
    theta, phi.  (The fused renderer uses sample_rays / fenerf_ray_setup instead and never builds the point tensor.)"""
    n, num_rays, num_steps, _ = points.shape
    points, z_vals = perturb_points(points, z_vals, ray_directions, device, draws)
    cam_origin, pitch, yaw = sample_camera_positions(n=n, r=1, horizontal_stddev=h_stddev, vertical_stddev=v_stddev,
                                                     horizontal_mean=h_mean, vertical_mean=v_mean, device=device, mode=mode,
                                                     draws=draws)
    rot = _lookat_rotation(normalize_vecs(-cam_origin))                       # [n,3,3]
    world_points = torch.matmul(points.reshape(n, -1, 3), rot.transpose(1, 2)) + cam_origin.unsqueeze(1)
    world_dirs = torch.matmul(ray_directions.reshape(n, -1, 3), rot.transpose(1, 2))
    world_origins = cam_origin.unsqueeze(1).expand(n, num_rays, 3).contiguous()
    return world_points.reshape(n, num_rays, num_steps, 3), z_vals, world_dirs.reshape(n, num_rays, 3), world_origins, pitch, yaw


def sample_rays(n, num_steps, device, fov, resolution, ray_start, ray_end, h_stddev, v_stddev, h_mean, v_mean, mode,
                draws=_DEFAULT_DRAWS):
    """What the fused renderer needs from get_initial_rays_trig + transform_sampled_points, without materialising
    the [n,R,N,3] point tensor: world-space origins/dirs [n,R,3] and jittered z [n,R,N] (same draws, same order:
    jitter rand -> theta -> phi).  Points are origins + dirs*z inside the kernel.
    On a GPU device this is ONE HIP launch (fenerf_ray_setup) after the draws; the PyTorch formulation below is the
    host-logic statement of the same math (CPU tests pin it to the reference's vectors, GPU tests pin the kernel to it)."""
    W, H = resolution
    if torch.device(device).type == "cuda" and W == H:
        u = draws.rand((n, W * H, num_steps, 1), device)
        theta, phi = sample_camera_angles(device, n, h_stddev, v_stddev, h_mean, v_mean, mode, draws=draws)
        z_cam = (-torch.ones(1) / np.tan((2 * math.pi * fov / 360) / 2)).item()   # as the reference's fp32 tensor op rounds it
        return native.ray_setup(n, W, num_steps, z_cam, ray_start, ray_end, u, theta, phi)
    rays_d_cam = _camera_ray_dirs(resolution, fov, device)                                   # [R,3]
    z_lin = torch.linspace(ray_start, ray_end, num_steps, device=device)                      # [N]
    step = (z_lin[1] - z_lin[0]) if num_steps > 1 else torch.zeros((), device=device)
    u = draws.rand((n, W * H, num_steps, 1), device)
    z_vals = z_lin.reshape(1, 1, num_steps) + (u.squeeze(-1) - 0.5) * step                    # perturb_points
    camera_origin, pitch, yaw = sample_camera_positions(n=n, r=1, horizontal_stddev=h_stddev, vertical_stddev=v_stddev,
                                                        horizontal_mean=h_mean, vertical_mean=v_mean, device=device,
                                                        mode=mode, draws=draws)
    rot = _lookat_rotation(normalize_vecs(-camera_origin))
    dirs = torch.matmul(rays_d_cam.unsqueeze(0), rot.transpose(1, 2))                        # [n,R,3]
    origins = camera_origin.unsqueeze(1).expand(n, W * H, 3).contiguous()
    return origins, dirs.contiguous(), z_vals.contiguous(), pitch, yaw


def fancy_integration(rgb_sigma, z_vals, device, noise_std=0.5, last_back=False, white_back=False, black_back=False,
                      clamp_mode=None, fill_mode=None, fill_color="black", draws=_DEFAULT_DRAWS):
    """NeRF alpha compositing on the GPU (HIP kernel behind fenerf_composite); reference :18-106.
    rgb_sigma [B,R,M,C], z_vals [B,R,M,1] -> the reference's 3-tuple for the given fill_mode."""
    opts = _lib.composite_opts(clamp_mode, noise_std, last_back, white_back, black_back, fill_mode, fill_color)
    noise = draws.randn(rgb_sigma[..., -1:].shape, rgb_sigma.device)  # always drawn, like the reference (:27)
    rgb, depth, weights, wsum = native.composite(rgb_sigma, z_vals.squeeze(-1), noise.squeeze(-1) if noise_std != 0 else None, opts)
    depth = depth.unsqueeze(-1)
    if fill_mode in ("weight", "eval_seg_padding_background", "eval_white_back"):
        return rgb, depth, wsum.unsqueeze(-1).expand_as(rgb)
    return rgb, depth, weights.unsqueeze(-1)


def sample_pdf(bins, weights, N_importance, det=False, eps=1e-5, draws=_DEFAULT_DRAWS):
    """Inverse-CDF sampling (:259-300) as a HIP kernel (fenerf_sample_pdf): bins [R,K+1], weights [R,K]."""
    if eps != 1e-5:
        raise ValueError("sample_pdf: eps is fixed at 1e-5 in the HIP kernel")
    R, K = weights.shape
    dev = bins.device
    if det:
        u = torch.linspace(0, 1, N_importance, device=dev).expand(R, N_importance)
    else:
        u = draws.rand((R, N_importance), dev)
    return native.sample_pdf(bins, weights, u.contiguous())
