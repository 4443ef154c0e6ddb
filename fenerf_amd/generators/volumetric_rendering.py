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


class RecordedDraws:
    """Replays a list of recorded arrays (tests): each call pops the next one and checks its shape."""

    def __init__(self, arrays):
        self.arrays = list(arrays)

    def _next(self, shape, device):
        a = self.arrays.pop(0)
        t = torch.as_tensor(np.asarray(a), dtype=torch.float32, device=device)
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t

    rand = _next
    randn = _next


_DEFAULT_DRAWS = TorchDraws()


def get_initial_rays_trig(n, num_steps, device, fov, resolution, ray_start, ray_end):
    """Camera-space sample points, z_vals and ray directions (volumetric_rendering.py:109-131)."""
    W, H = resolution
    x, y = torch.meshgrid(torch.linspace(-1, 1, W, device=device), torch.linspace(1, -1, H, device=device), indexing="ij")
    x = x.T.flatten()
    y = y.T.flatten()
    z = -torch.ones_like(x, device=device) / np.tan((2 * math.pi * fov / 360) / 2)
    rays_d_cam = normalize_vecs(torch.stack([x, y, z], -1))
    z_vals = torch.linspace(ray_start, ray_end, num_steps, device=device).reshape(1, num_steps, 1).repeat(W * H, 1, 1)
    points = rays_d_cam.unsqueeze(1).repeat(1, num_steps, 1) * z_vals
    points = torch.stack(n * [points])
    z_vals = torch.stack(n * [z_vals])
    rays_d_cam = torch.stack(n * [rays_d_cam]).to(device)
    return points, z_vals, rays_d_cam


def perturb_points(points, z_vals, ray_directions, device, draws=_DEFAULT_DRAWS):
    distance_between_points = z_vals[:, :, 1:2, :] - z_vals[:, :, 0:1, :]
    offset = (draws.rand(z_vals.shape, device) - 0.5) * distance_between_points
    z_vals = z_vals + offset
    points = points + offset * ray_directions.unsqueeze(2)
    return points, z_vals


def truncated_normal_(tensor, mean=0, std=1):
    size = tensor.shape
    tmp = tensor.new_empty(size + (4,)).normal_()
    valid = (tmp < 2) & (tmp > -2)
    ind = valid.max(-1, keepdim=True)[1]
    tensor.data.copy_(tmp.gather(-1, ind).squeeze(-1))
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
        if random.random() < 0.5:
            theta = (draws.rand((n, 1), device) - 0.5) * 2 * horizontal_stddev * 2 + horizontal_mean
            phi = (draws.rand((n, 1), device) - 0.5) * 2 * vertical_stddev * 2 + vertical_mean
        else:
            theta = draws.randn((n, 1), device) * horizontal_stddev + horizontal_mean
            phi = draws.randn((n, 1), device) * vertical_stddev + vertical_mean
    elif mode == "truncated_gaussian":
        theta = truncated_normal_(torch.zeros((n, 1), device=device)) * horizontal_stddev + horizontal_mean
        phi = truncated_normal_(torch.zeros((n, 1), device=device)) * vertical_stddev + vertical_mean
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
    """Camera origins on a sphere; returns (origin [n,3], phi/pitch [n,1], theta/yaw [n,1])  (:179-228)."""
    if mode == "uniform":
        theta = (draws.rand((n, 1), device) - 0.5) * 2 * horizontal_stddev + horizontal_mean
        phi = (draws.rand((n, 1), device) - 0.5) * 2 * vertical_stddev + vertical_mean
    elif mode == "normal" or mode == "gaussian":
        theta = draws.randn((n, 1), device) * horizontal_stddev + horizontal_mean
        phi = draws.randn((n, 1), device) * vertical_stddev + vertical_mean
    elif mode == "hybrid":
        if random.random() < 0.5:
            theta = (draws.rand((n, 1), device) - 0.5) * 2 * horizontal_stddev * 2 + horizontal_mean
            phi = (draws.rand((n, 1), device) - 0.5) * 2 * vertical_stddev * 2 + vertical_mean
        else:
            theta = draws.randn((n, 1), device) * horizontal_stddev + horizontal_mean
            phi = draws.randn((n, 1), device) * vertical_stddev + vertical_mean
    elif mode == "truncated_gaussian":
        theta = truncated_normal_(torch.zeros((n, 1), device=device)) * horizontal_stddev + horizontal_mean
        phi = truncated_normal_(torch.zeros((n, 1), device=device)) * vertical_stddev + vertical_mean
    elif mode == "spherical_uniform":
        theta = (draws.rand((n, 1), device) - .5) * 2 * horizontal_stddev + horizontal_mean
        v_stddev, v_mean = vertical_stddev / math.pi, vertical_mean / math.pi
        v = ((draws.rand((n, 1), device) - .5) * 2 * v_stddev + v_mean)
        v = torch.clamp(v, 1e-5, 1 - 1e-5)
        phi = torch.arccos(1 - 2 * v)
    else:  # just use the mean
        theta = torch.ones((n, 1), device=device, dtype=torch.float) * horizontal_mean
        phi = torch.ones((n, 1), device=device, dtype=torch.float) * vertical_mean
    phi = torch.clamp(phi, 1e-5, math.pi - 1e-5)
    output_points = torch.zeros((n, 3), device=device)
    output_points[:, 0:1] = r * torch.sin(phi) * torch.cos(theta)
    output_points[:, 2:3] = r * torch.sin(phi) * torch.sin(theta)
    output_points[:, 1:2] = r * torch.cos(phi)
    return output_points, phi, theta


def create_cam2world_matrix(forward_vector, origin, device=None):
    """Look-at matrix with +y up (:230-248)."""
    forward_vector = normalize_vecs(forward_vector)
    up_vector = torch.tensor([0, 1, 0], dtype=torch.float, device=device).expand_as(forward_vector)
    left_vector = normalize_vecs(torch.cross(up_vector, forward_vector, dim=-1))
    up_vector = normalize_vecs(torch.cross(forward_vector, left_vector, dim=-1))
    rotation_matrix = torch.eye(4, device=device).unsqueeze(0).repeat(forward_vector.shape[0], 1, 1)
    rotation_matrix[:, :3, :3] = torch.stack((-left_vector, up_vector, -forward_vector), axis=-1)
    translation_matrix = torch.eye(4, device=device).unsqueeze(0).repeat(forward_vector.shape[0], 1, 1)
    translation_matrix[:, :3, 3] = origin
    return translation_matrix @ rotation_matrix


def transform_sampled_points(points, z_vals, ray_directions, device, h_stddev=1, v_stddev=1, h_mean=math.pi * 0.5,
                             v_mean=math.pi * 0.5, mode="normal", draws=_DEFAULT_DRAWS):
    """Reference-shaped API: jitter + camera pose + cam->world of points, dirs, origins (:142-168)."""
    n, num_rays, num_steps, channels = points.shape
    points, z_vals = perturb_points(points, z_vals, ray_directions, device, draws)
    camera_origin, pitch, yaw = sample_camera_positions(n=points.shape[0], r=1, horizontal_stddev=h_stddev,
                                                        vertical_stddev=v_stddev, horizontal_mean=h_mean,
                                                        vertical_mean=v_mean, device=device, mode=mode, draws=draws)
    forward_vector = normalize_vecs(-camera_origin)
    cam2world_matrix = create_cam2world_matrix(forward_vector, camera_origin, device=device)
    points_homogeneous = torch.ones((points.shape[0], points.shape[1], points.shape[2], points.shape[3] + 1), device=device)
    points_homogeneous[:, :, :, :3] = points
    transformed_points = torch.bmm(cam2world_matrix, points_homogeneous.reshape(n, -1, 4).permute(0, 2, 1)).permute(0, 2, 1).reshape(n, num_rays, num_steps, 4)
    transformed_ray_directions = torch.bmm(cam2world_matrix[..., :3, :3], ray_directions.reshape(n, -1, 3).permute(0, 2, 1)).permute(0, 2, 1).reshape(n, num_rays, 3)
    homogeneous_origins = torch.zeros((n, 4, num_rays), device=device)
    homogeneous_origins[:, 3, :] = 1
    transformed_ray_origins = torch.bmm(cam2world_matrix, homogeneous_origins).permute(0, 2, 1).reshape(n, num_rays, 4)[..., :3]
    return transformed_points[..., :3], z_vals, transformed_ray_directions, transformed_ray_origins, pitch, yaw


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
    x, y = torch.meshgrid(torch.linspace(-1, 1, W, device=device), torch.linspace(1, -1, H, device=device), indexing="ij")
    x = x.T.flatten()
    y = y.T.flatten()
    zc = -torch.ones_like(x) / np.tan((2 * math.pi * fov / 360) / 2)
    rays_d_cam = normalize_vecs(torch.stack([x, y, zc], -1))                                  # [R,3]
    z_lin = torch.linspace(ray_start, ray_end, num_steps, device=device)                      # [N]
    step = (z_lin[1] - z_lin[0]) if num_steps > 1 else torch.zeros((), device=device)
    u = draws.rand((n, W * H, num_steps, 1), device)
    z_vals = z_lin.reshape(1, 1, num_steps) + (u.squeeze(-1) - 0.5) * step                    # perturb_points
    camera_origin, pitch, yaw = sample_camera_positions(n=n, r=1, horizontal_stddev=h_stddev, vertical_stddev=v_stddev,
                                                        horizontal_mean=h_mean, vertical_mean=v_mean, device=device,
                                                        mode=mode, draws=draws)
    cam2world = create_cam2world_matrix(normalize_vecs(-camera_origin), camera_origin, device=device)
    dirs = torch.matmul(rays_d_cam.unsqueeze(0), cam2world[:, :3, :3].transpose(1, 2))       # [n,R,3]
    origins = cam2world[:, :3, 3].unsqueeze(1).expand(n, W * H, 3).contiguous()
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
