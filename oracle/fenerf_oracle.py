"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

A numpy restatement, written from the behaviour spec in SURVEY.md appendix A, of
FENeRF's generator hot path (reference = /root/reference, pure PyTorch):

    generators/volumetric_rendering.py   rays, jitter, camera pose, compositing, inverse-CDF resampling
    siren/siren.py                       FiLM-SIREN radiance field + 3-D feature grid
    generators/generators.py             coarse -> resample -> fine -> merge -> composite orchestration

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker.  PARITY PINNING: the reference ships no tests or
golden vectors of its own, so this oracle is pinned against outputs
of the reference itself, imported in the build container by tools/make_golden.py
and committed as tests/golden/*.npz (tests/test_oracle_golden.py).

All randomness is an *input* (SURVEY §0.6): functions take the uniform / normal
draws the reference would have made, in the reference's draw order (appendix A.6).
Arithmetic is fp32 by default (like the reference CPU path); pass dtype=np.float64
for a high-precision arbiter.
"""
import math

import numpy as np

# ---------------------------------------------------------------------------
# a1  get_initial_rays_trig            generators/volumetric_rendering.py:109-131
# ---------------------------------------------------------------------------
def get_initial_rays_trig(n, num_steps, fov, resolution, ray_start, ray_end, dtype=np.float32):
    W, H = resolution
    # torch.linspace in fp32: start + i*step for the first half, end - (N-1-i)*step for the second
    xs = _torch_linspace(-1.0, 1.0, W, dtype)
    ys = _torch_linspace(1.0, -1.0, H, dtype)
    # meshgrid(indexing='ij') then .T.flatten(): ray p = i*W + j has x = xs[j], y = ys[i]
    x = np.tile(xs, H)
    y = np.repeat(ys, W)
    z = -np.ones_like(x) / dtype(np.tan((2 * math.pi * fov / 360) / 2))
    d = np.stack([x, y, z], -1).astype(dtype)
    d = d / np.sqrt((d * d).sum(-1, keepdims=True, dtype=dtype))
    z_vals = _torch_linspace(ray_start, ray_end, num_steps, dtype).reshape(1, num_steps, 1).repeat(W * H, 0)
    points = d[:, None, :] * z_vals
    points = np.broadcast_to(points, (n,) + points.shape).copy()
    z_vals = np.broadcast_to(z_vals, (n,) + z_vals.shape).copy()
    rays_d = np.broadcast_to(d, (n,) + d.shape).copy()
    return points, z_vals, rays_d


def _torch_linspace(start, end, steps, dtype=np.float32):
    """torch.linspace's CPU kernel: step=(end-start)/(steps-1); i<steps/2 ? start+step*i : end-step*(steps-1-i)."""
    start, end = dtype(start), dtype(end)
    if steps == 1:
        return np.array([start], dtype=dtype)
    step = dtype((end - start) / dtype(steps - 1))
    i = np.arange(steps)
    half = steps // 2
    lo = (start + step * i.astype(dtype)).astype(dtype)
    hi = (end - step * (steps - 1 - i).astype(dtype)).astype(dtype)
    return np.where(i < half, lo, hi).astype(dtype)


# ---------------------------------------------------------------------------
# a2  perturb_points                    volumetric_rendering.py:133-139
# ---------------------------------------------------------------------------
def perturb_points(points, z_vals, ray_directions, u_jitter):
    """u_jitter ~ U[0,1) of z_vals' shape (the reference's torch.rand)."""
    dist = z_vals[:, :, 1:2, :] - z_vals[:, :, 0:1, :]
    offset = (u_jitter.astype(z_vals.dtype) - z_vals.dtype.type(0.5)) * dist
    z_vals = z_vals + offset
    points = points + offset * ray_directions[:, :, None, :]
    return points, z_vals


# ---------------------------------------------------------------------------
# a3  sample_camera_positions           volumetric_rendering.py:179-228
# ---------------------------------------------------------------------------
def camera_angles(mode, n, h_stddev, v_stddev, h_mean, v_mean, r_theta=None, r_phi=None, dtype=np.float32, coin=None):
    """Turns the reference's raw random draws into (theta, phi) *before* the phi clamp.
    r_theta / r_phi are the [n,1] draws the reference makes, in that order:
      'uniform'            : torch.rand            (:188-190)
      'normal'/'gaussian'  : torch.randn           (:192-194)
      'spherical_uniform'  : torch.rand            (:208-213)
      anything else        : no draw (mean pose)   (:215-218)
      'hybrid'             : coin = random.random() first; < 0.5 -> torch.rand x2 at twice the spread, else randn x2 (:195-201)
      'truncated_gaussian' : r_theta / r_phi are the [n,1,4] candidate blocks `new_empty(...).normal_()` of truncated_normal_
                             (:170-177): the first candidate strictly inside (-2, 2), candidate 0 if none is (:203-205)"""
    if mode == "hybrid":
        if coin < 0.5:
            theta = (r_theta - 0.5) * 2 * h_stddev * 2 + h_mean
            phi = (r_phi - 0.5) * 2 * v_stddev * 2 + v_mean
        else:
            theta = r_theta * h_stddev + h_mean
            phi = r_phi * v_stddev + v_mean
    elif mode == "truncated_gaussian":
        def first_inside(c):
            ok = (c < 2) & (c > -2)
            idx = ok.argmax(-1)                      # first True; 0 when none (torch .max(-1)[1] on a bool tensor, same rule)
            return np.take_along_axis(c, idx[..., None], -1)[..., 0]
        theta = first_inside(r_theta) * h_stddev + h_mean
        phi = first_inside(r_phi) * v_stddev + v_mean
    elif mode == "uniform":
        theta = (r_theta - 0.5) * 2 * h_stddev + h_mean
        phi = (r_phi - 0.5) * 2 * v_stddev + v_mean
    elif mode in ("normal", "gaussian"):
        theta = r_theta * h_stddev + h_mean
        phi = r_phi * v_stddev + v_mean
    elif mode == "spherical_uniform":
        theta = (r_theta - 0.5) * 2 * h_stddev + h_mean
        vs, vm = v_stddev / math.pi, v_mean / math.pi
        v = np.clip((r_phi - 0.5) * 2 * vs + vm, 1e-5, 1 - 1e-5)
        phi = np.arccos(1 - 2 * v)
    else:
        theta = np.ones((n, 1)) * h_mean
        phi = np.ones((n, 1)) * v_mean
    return theta.astype(dtype), phi.astype(dtype)


def camera_origin(theta, phi, r=1.0):
    """volumetric_rendering.py:220-228 -- clamp phi, spherical -> cartesian (y up)."""
    dt = theta.dtype.type
    phi = np.clip(phi, dt(1e-5), dt(math.pi - 1e-5))
    o = np.zeros((theta.shape[0], 3), dtype=theta.dtype)
    o[:, 0:1] = r * np.sin(phi) * np.cos(theta)
    o[:, 2:3] = r * np.sin(phi) * np.sin(theta)
    o[:, 1:2] = r * np.cos(phi)
    return o, phi, theta


def normalize_vecs(v):
    """generators/math_utils_torch.py:16-20"""
    return v / np.sqrt((v * v).sum(-1, keepdims=True))


# ---------------------------------------------------------------------------
# a4  create_cam2world_matrix           volumetric_rendering.py:230-248
# ---------------------------------------------------------------------------
def create_cam2world_matrix(forward, origin):
    dt = forward.dtype
    forward = normalize_vecs(forward)
    up = np.broadcast_to(np.array([0, 1, 0], dtype=dt), forward.shape)
    left = normalize_vecs(np.cross(up, forward))
    up = normalize_vecs(np.cross(forward, left))
    n = forward.shape[0]
    rot = np.tile(np.eye(4, dtype=dt), (n, 1, 1))
    rot[:, :3, :3] = np.stack((-left, up, -forward), axis=-1)
    tr = np.tile(np.eye(4, dtype=dt), (n, 1, 1))
    tr[:, :3, 3] = origin
    return tr @ rot


# ---------------------------------------------------------------------------
# a5  transform_sampled_points          volumetric_rendering.py:142-168
# ---------------------------------------------------------------------------
def transform_sampled_points(points, z_vals, ray_directions, u_jitter, theta, phi):
    """theta/phi: pre-clamp camera angles [n,1] (see camera_angles).  Returns the reference's
    6-tuple (points_world, z_vals, dirs_world, origins_world, pitch(=phi), yaw(=theta))."""
    n, R, N, _ = points.shape
    dt = points.dtype
    points, z_vals = perturb_points(points, z_vals, ray_directions, u_jitter)
    o, pitch, yaw = camera_origin(theta.astype(dt), phi.astype(dt))
    fwd = normalize_vecs(-o)
    c2w = create_cam2world_matrix(fwd, o)
    ph = np.ones((n, R, N, 4), dtype=dt)
    ph[..., :3] = points
    tp = np.einsum("nij,npj->npi", c2w, ph.reshape(n, -1, 4)).reshape(n, R, N, 4)
    td = np.einsum("nij,npj->npi", c2w[:, :3, :3], ray_directions.reshape(n, -1, 3)).reshape(n, R, 3)
    ho = np.zeros((n, R, 4), dtype=dt)
    ho[..., 3] = 1
    to = np.einsum("nij,npj->npi", c2w, ho)[..., :3]
    return tp[..., :3].astype(dt), z_vals, td.astype(dt), to.astype(dt), pitch, yaw


# ---------------------------------------------------------------------------
# a6  CustomMappingNetwork              siren/siren.py:82-102
# ---------------------------------------------------------------------------
def mapping_network(sd, prefix, z):
    """z [B, z_dim] -> (frequencies, phase_shifts); Linear/LeakyReLU(0.2) x4 then Linear."""
    x = z
    j = 0
    while f"{prefix}.network.{2 * j}.weight" in sd:
        w, b = sd[f"{prefix}.network.{2 * j}.weight"], sd[f"{prefix}.network.{2 * j}.bias"]
        x = x @ w.T.astype(x.dtype) + b.astype(x.dtype)
        if f"{prefix}.network.{2 * (j + 1)}.weight" in sd:
            x = np.where(x >= 0, x, x * x.dtype.type(0.2))
        j += 1
    half = x.shape[-1] // 2
    return x[..., :half], x[..., half:]


def truncate(avg, raw, psi):
    """a7 truncation trick, generators/generators.py:561-564"""
    return avg + psi * (raw - avg)


# ---------------------------------------------------------------------------
# a9  sample_from_3dgrid                siren/siren.py:314-330
#     (= F.grid_sample 5-D, mode='bilinear' i.e. trilinear, zeros padding, align_corners=True)
# ---------------------------------------------------------------------------
def sample_from_3dgrid(coords, grid):
    """coords [B,P,3] in [-1,1] (x->last grid dim W, y->H, z->D); grid [1,C,D,H,W] -> [B,P,C]."""
    dt = coords.dtype
    _, C, D, Hh, W = grid.shape
    g = grid[0].astype(dt)

    def unnorm(c, size):  # align_corners=True
        return (c + dt.type(1)) / dt.type(2) * dt.type(size - 1)

    ix, iy, iz = unnorm(coords[..., 0], W), unnorm(coords[..., 1], Hh), unnorm(coords[..., 2], D)
    x0, y0, z0 = np.floor(ix), np.floor(iy), np.floor(iz)
    out = np.zeros(coords.shape[:2] + (C,), dtype=dt)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                xi, yi, zi = x0 + dx, y0 + dy, z0 + dz
                wx = (ix - x0) if dx else (x0 + 1 - ix)
                wy = (iy - y0) if dy else (y0 + 1 - iy)
                wz = (iz - z0) if dz else (z0 + 1 - iz)
                w = (wx * wy * wz).astype(dt)
                ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= Hh - 1) & (zi >= 0) & (zi <= D - 1)
                xc = np.clip(xi, 0, W - 1).astype(np.int64)
                yc = np.clip(yi, 0, Hh - 1).astype(np.int64)
                zc = np.clip(zi, 0, D - 1).astype(np.int64)
                vals = g[:, zc, yc, xc]  # [C,B,P]
                out += (np.moveaxis(vals, 0, -1) * (w * ok)[..., None]).astype(dt)
    return out


# ---------------------------------------------------------------------------
# f4 SPATIALSIRENGRID host pieces       siren/siren.py:479-518
# ---------------------------------------------------------------------------
def sample_local_latents(latents, xyz):
    """latents [B,Cz,h,w], xyz [B,P,3] (box-warped) -> [B,P,Cz]: F.grid_sample(bilinear, align_corners=False, zeros padding)
    at (x, z) -- x indexes the width axis, z the height axis (siren.py:479-499)."""
    lat = np.asarray(latents, dtype=np.float64)
    B, Cz, h, w = lat.shape
    gx, gy = np.asarray(xyz)[..., 0].astype(np.float64), np.asarray(xyz)[..., 2].astype(np.float64)
    ix, iy = ((gx + 1) * w - 1) / 2, ((gy + 1) * h - 1) / 2          # align_corners=False unnormalisation
    x0, y0 = np.floor(ix), np.floor(iy)
    out = np.zeros(gx.shape + (Cz,))
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            wgt = (ix - x0 if dx else x0 + 1 - ix) * (iy - y0 if dy else y0 + 1 - iy)
            ok = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1)
            xc, yc = np.clip(xi, 0, w - 1).astype(np.int64), np.clip(yi, 0, h - 1).astype(np.int64)
            for b in range(B):
                out[b] += (lat[b][:, yc[b], xc[b]].T * (wgt[b] * ok[b])[:, None])
    return out.astype(np.float32)


def get_local_coordinates(global_coords, local_grid_length, preserve_y=True):
    """siren.py:501-518; torch.round and np.round both round half to even."""
    gc = np.asarray(global_coords, dtype=np.float32)
    local = (gc + np.float32(1)) / np.float32(2) * np.float32(local_grid_length)
    local = local - np.round(local - np.float32(0.5))
    local = local * np.float32(2) - np.float32(1)
    if preserve_y:
        return np.concatenate([local[..., 0:1], gc[..., 1:2], local[..., 2:3]], -1)
    return local


# ---------------------------------------------------------------------------
# a10 FiLMLayer                         siren/siren.py:113-123
# ---------------------------------------------------------------------------
def film_layer(x, w, b, freq, phase, rev_tap=None):
    """sin(freq * (x W^T + b) + phase); freq/phase [B,H] broadcast over the point axis, or [B,P,H] taken per point
    (FiLMLayer skips the broadcast when the shapes agree, siren.py:119-122: SPATIALSIRENGRID's per-point modulation).
    rev_tap: optional list that receives max|sine argument| of the layer in REVOLUTIONS (tests: how far beyond the init range)."""
    y = x @ w.T.astype(x.dtype) + b.astype(x.dtype)
    th = freq * y + phase if freq.ndim == 3 else freq[:, None, :] * y + phase[:, None, :]
    if rev_tap is not None:
        rev_tap.append(float(np.abs(th).max() / (2 * math.pi)))
    return np.sin(th)


# ---------------------------------------------------------------------------
# a11/a12 SIREN heads                   siren/siren.py:1509-1530 (texture), :1210-1229 (baseline), :227-244 (spatial)
# ---------------------------------------------------------------------------
BOX_SCALE = 2 / 0.24  # UniformBoxWarp(0.24), siren.py:181-187


def siren_forward(sd, spec, points, ray_dirs, freq_geo, phase_geo, freq_app=None, phase_app=None, dtype=np.float32, rev_tap=None):
    """points, ray_dirs [B,P,3]; raw frequencies/phases [B, n*H] (pre '*15+30').
    Returns [B,P,output_dim] = [labels(18) | rgb(3) | sigma(1)] ('spatial': [rgb | sigma]).
    rev_tap: optional list, gets one max|sine argument| (revolutions) per FiLM layer."""
    H = spec["hidden_dim"]
    dt = np.dtype(dtype)
    P = lambda a: np.asarray(a).astype(dt)
    pts, dirs = P(points), P(ray_dirs)
    fg = P(freq_geo) * dt.type(15) + dt.type(30)
    pg = P(phase_geo)
    x = pts * dt.type(BOX_SCALE)
    feats = sample_from_3dgrid(x, P(sd["spatial_embeddings"])) if spec["grid_ch"] else None
    for i in range(spec["n_geo"]):
        x = film_layer(x, P(sd[f"network.{i}.layer.weight"]), P(sd[f"network.{i}.layer.bias"]),
                       fg[..., i * H:(i + 1) * H], pg[..., i * H:(i + 1) * H], rev_tap)
    sigma = x @ P(sd["final_layer.weight"]).T + P(sd["final_layer.bias"])
    if spec["kind"] == "spatial":
        c = np.concatenate([dirs, x], -1)
        c = film_layer(c, P(sd["color_layer_sine.layer.weight"]), P(sd["color_layer_sine.layer.bias"]),
                       fg[..., -H:], pg[..., -H:], rev_tap)
        rgb = _sigmoid(c @ P(sd["color_layer_linear.0.weight"]).T + P(sd["color_layer_linear.0.bias"]))
        return np.concatenate([rgb, sigma], -1)
    fa = P(freq_app) * dt.type(15) + dt.type(30)
    pa = P(phase_app)
    labels = x
    for i in range(spec["n_label_layers"]):  # linear layers, NO activation between (siren.py:1490-1494)
        labels = labels @ P(sd[f"label_layer_linear.{i}.weight"]).T + P(sd[f"label_layer_linear.{i}.bias"])
    c = np.concatenate([dirs, feats, x], -1) if feats is not None else np.concatenate([dirs, x], -1)
    for i in range(spec["n_color"]):
        c = film_layer(c, P(sd[f"color_layer_sine.{i}.layer.weight"]), P(sd[f"color_layer_sine.{i}.layer.bias"]),
                       fa[:, i * H:(i + 1) * H], pa[:, i * H:(i + 1) * H], rev_tap)
    rgb = _sigmoid(c @ P(sd["color_layer_linear.0.weight"]).T + P(sd["color_layer_linear.0.bias"]))
    return np.concatenate([labels, rgb, sigma], -1)


def _sigmoid(x):
    return (1 / (1 + np.exp(-x))).astype(x.dtype)


def _softplus(x):  # F.softplus, beta=1, threshold=20
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20)))).astype(x.dtype)


# ---------------------------------------------------------------------------
# a13 fancy_integration                 volumetric_rendering.py:18-106
# ---------------------------------------------------------------------------
FILL_COLORS = {"white": 1.0, "black": 0.0, "grey": 0.5, "light_grey": 0.81}


def fancy_integration(rgb_sigma, z_vals, noise=None, noise_std=0.5, last_back=False, white_back=False,
                      black_back=False, clamp_mode=None, fill_mode=None, fill_color="black"):
    """rgb_sigma [B,R,M,C], z_vals [B,R,M,1]; noise = the reference's torch.randn(sigmas.shape)
    draw (un-scaled; None == zeros).  Returns the reference's 3-tuple for the given fill_mode."""
    dt = rgb_sigma.dtype
    rgbs, sigmas = rgb_sigma[..., :-1], rgb_sigma[..., -1:]
    deltas = z_vals[:, :, 1:] - z_vals[:, :, :-1]
    deltas = np.concatenate([deltas, np.full_like(deltas[:, :, :1], 1e10)], -2)
    nz = (noise.astype(dt) * dt.type(noise_std)) if noise is not None else dt.type(0)
    if clamp_mode == "softplus":
        alphas = 1 - np.exp(-deltas * _softplus(sigmas + nz))
    elif clamp_mode == "relu":
        alphas = 1 - np.exp(-deltas * np.maximum(sigmas + nz, 0))
    else:
        raise TypeError("Need to choose clamp mode")  # reference raises a str -> TypeError (:34)
    alphas = alphas.astype(dt)
    shifted = np.concatenate([np.ones_like(alphas[:, :, :1]), 1 - alphas + dt.type(1e-10)], -2)
    weights = alphas * np.cumprod(shifted, -2, dtype=dt)[:, :, :-1]
    weights_sum = weights.sum(2, dtype=dt)
    if last_back:
        weights = weights.copy()
        weights[:, :, -1] += (1 - weights_sum)
    rgb_final = np.sum(weights * rgbs, -2, dtype=dt)
    depth_final = np.sum(weights * z_vals, -2, dtype=dt)
    if white_back:
        rgb_final = rgb_final + 1 - weights_sum
    if black_back:
        rgb_final = rgb_final + (1 - weights_sum) * -1
    low = weights_sum[..., 0] < dt.type(0.9)
    if fill_mode in ("debug", "weight_debug"):
        # the reference assigns a fixed 22-vector (:54,:66): only shape-valid for 22 colour channels,
        # i.e. it raises RuntimeError for the output_dim=22 models (21 channels here)
        if rgb_final.shape[-1] != 22:
            raise RuntimeError("shape mismatch: value tensor of shape [22] cannot be broadcast to indexing result")
        rgb_final = rgb_final.copy()
        rgb_final[low] = np.array([1.0] + [0.0] * 21, dtype=dt)
        if fill_mode == "debug":
            return rgb_final, depth_final, weights
        return rgb_final, depth_final, np.broadcast_to(weights_sum, rgb_final.shape)
    if fill_mode == "weight":
        return rgb_final, depth_final, np.broadcast_to(weights_sum, rgb_final.shape)
    if fill_mode in ("seg_padding_background", "eval_seg_padding_background"):
        rgb_final = np.concatenate([np.zeros_like(rgb_final[..., :1]), rgb_final], -1)
        if fill_color in FILL_COLORS:
            c = FILL_COLORS[fill_color]
            rgb_final[low] = np.array([1.0] + [c] * (rgb_final.shape[-1] - 1), dtype=dt)
        if fill_mode == "seg_padding_background":
            return rgb_final, depth_final, weights
        return rgb_final, depth_final, np.broadcast_to(weights_sum, rgb_final.shape)
    if fill_mode == "eval_white_back":
        rgb_final = rgb_final.copy()
        rgb_final[low] = np.ones(rgb_final.shape[-1], dtype=dt)
        return rgb_final, depth_final, np.broadcast_to(weights_sum, rgb_final.shape)
    return rgb_final, depth_final, weights


# ---------------------------------------------------------------------------
# a14 sample_pdf                        volumetric_rendering.py:259-300
# ---------------------------------------------------------------------------
def sample_pdf(bins, weights, u, eps=1e-5):
    """bins [R,K+1], weights [R,K], u [R,Nimp] ~ U[0,1) (or linspace(0,1,Nimp) for det=True)."""
    dt = bins.dtype
    weights = weights + dt.type(eps)
    pdf = weights / weights.sum(-1, keepdims=True, dtype=dt)
    cdf = np.cumsum(pdf, -1, dtype=dt)
    cdf = np.concatenate([np.zeros_like(cdf[:, :1]), cdf], -1)
    K = weights.shape[1]
    # torch.searchsorted(cdf, u) (right=False): first index with cdf[idx] >= u
    inds = (cdf[:, None, :] < u[:, :, None]).sum(-1)
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, K)
    cg0 = np.take_along_axis(cdf, below, 1)
    cg1 = np.take_along_axis(cdf, above, 1)
    bg0 = np.take_along_axis(bins, below, 1)
    bg1 = np.take_along_axis(bins, above, 1)
    denom = cg1 - cg0
    denom = np.where(denom < dt.type(eps), dt.type(1), denom)
    return (bg0 + (u - cg0) / denom * (bg1 - bg0)).astype(dt)


# ---------------------------------------------------------------------------
# a15 hierarchical resample + merge     generators/generators.py:482-515
# ---------------------------------------------------------------------------
def fine_z_from_coarse(coarse_weights, z_vals, u):
    """coarse_weights [B,R,N,1] (fancy_integration's 3rd output), z_vals [B,R,N,1], u [B*R,N]
    -> fine z [B,R,N,1] (unsorted)."""
    B, R, N, _ = z_vals.shape
    dt = z_vals.dtype
    w = coarse_weights.reshape(B * R, N) + dt.type(1e-5)
    z = z_vals.reshape(B * R, N)
    z_mid = dt.type(0.5) * (z[:, :-1] + z[:, 1:])
    fine = sample_pdf(z_mid, w[:, 1:-1], u.astype(dt))
    return fine.reshape(B, R, N, 1)


def merge_sorted(fine_out, coarse_out, fine_z, coarse_z):
    """cat([fine, coarse]) -> ascending sort on z -> gather (generators.py:508-512)."""
    all_out = np.concatenate([fine_out, coarse_out], -2)
    all_z = np.concatenate([fine_z, coarse_z], -2)
    idx = np.argsort(all_z, axis=-2, kind="stable")
    return np.take_along_axis(all_out, idx, -2), np.take_along_axis(all_z, idx, -2)


# ---------------------------------------------------------------------------
# a15-a17 the whole forward render with all random draws as inputs
# ---------------------------------------------------------------------------
def render_forward(sd, spec, film, img_size, fov, ray_start, ray_end, num_steps, rand, hierarchical_sample=True,
                   lock_view_dependence=False, clamp_mode="relu", nerf_noise=0.0, last_back=False,
                   white_back=False, black_back=False, fill_mode=None, fill_color="black",
                   softmax_label=False, dtype=np.float32, return_stages=False):
    """film: dict(freq_geo, phase_geo, freq_app, phase_app) raw mapping-net outputs (already truncated if wanted).
    rand: dict(u_jitter [B,R,N,1], theta [B,1], phi [B,1] (pre-clamp angles), noise_coarse [B,R,N,1] or None,
               u_fine [B*R,N], noise_fine [B,R,2N,1] or None)  -- appendix A.6 order.
    Mirrors DoubleImplicitGenerator3d.forward (generators.py:452-527) when fill_mode is None and
    .staged_forward (:546-646) when fill_mode is given.  Returns (pixels [B,C,S,S] in [-1,1], depth [B,S,S],
    third) where third is whatever fancy_integration returned third."""
    B = film["freq_geo"].shape[0]
    S, N = img_size, num_steps
    dt = np.dtype(dtype)
    pts_cam, z_vals, d_cam = get_initial_rays_trig(B, N, fov, (S, S), ray_start, ray_end, dt.type)
    pts, z_vals, dirs, origins, pitch, yaw = transform_sampled_points(
        pts_cam, z_vals, d_cam, rand["u_jitter"].astype(dt), rand["theta"], rand["phi"])
    R = S * S
    dirs_exp = np.broadcast_to(dirs[:, :, None, :], (B, R, N, 3)).reshape(B, R * N, 3)
    if lock_view_dependence:
        dirs_exp = np.zeros_like(dirs_exp)
        dirs_exp[..., -1] = -1
    args = (film["freq_geo"], film["phase_geo"], film.get("freq_app"), film.get("phase_app"))
    coarse = siren_forward(sd, spec, pts.reshape(B, R * N, 3), dirs_exp, *args, dtype=dt).reshape(B, R, N, -1)
    stages = dict(points=pts, z_coarse=z_vals, dirs=dirs, origins=origins, pitch=pitch, yaw=yaw, coarse=coarse)
    if hierarchical_sample:
        _, _, w = fancy_integration(coarse, z_vals, noise=rand.get("noise_coarse"), noise_std=nerf_noise,
                                    clamp_mode=clamp_mode)
        fine_z = fine_z_from_coarse(w, z_vals, rand["u_fine"])
        fine_pts = origins[:, :, None, :] + dirs[:, :, None, :] * fine_z
        fine = siren_forward(sd, spec, fine_pts.reshape(B, R * N, 3), dirs_exp, *args, dtype=dt).reshape(B, R, N, -1)
        all_out, all_z = merge_sorted(fine, coarse, fine_z, z_vals)
        stages.update(coarse_weights=w, z_fine=fine_z, fine=fine, all_out=all_out, all_z=all_z)
        noise_final = rand.get("noise_fine")
    else:
        all_out, all_z = coarse, z_vals
        noise_final = rand.get("noise_fine", rand.get("noise_coarse"))
    pixels, depth, third = fancy_integration(all_out, all_z, noise=noise_final, noise_std=nerf_noise,
                                             white_back=white_back, last_back=last_back, black_back=black_back,
                                             clamp_mode=clamp_mode, fill_mode=fill_mode, fill_color=fill_color)
    if softmax_label:
        seg, rgb = pixels[..., :-3], pixels[..., -3:]
        e = np.exp(seg - seg.max(-1, keepdims=True))
        pixels = np.concatenate([e / e.sum(-1, keepdims=True), rgb], -1)
    img = pixels.reshape(B, S, S, -1).transpose(0, 3, 1, 2) * dt.type(2) - dt.type(1)
    out = (np.ascontiguousarray(img), depth.reshape(B, S, S), third)
    if return_stages:
        stages.update(pixels_flat=pixels, depth_flat=depth)
        return out + (stages,)
    return out


# ---------------------------------------------------------------------------
# mask2color (caller side; train_double_latent_semantic.py:36-72) -- exact-argmax semantics check
# ---------------------------------------------------------------------------
def label_argmax(pixels_nchw, n_rgb=3):
    """argmax over the semantic channels of a [B,C,S,S] render -> [B,S,S] int64."""
    return np.argmax(pixels_nchw[:, :-n_rgb], axis=1)
