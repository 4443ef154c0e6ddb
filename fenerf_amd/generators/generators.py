"""Implicit generators for 3-D volumes -- drop-in for the reference's generators/generators.py API
(ImplicitGenerator3d :13-431, DoubleImplicitGenerator3d :434-910): same constructor, attributes, method
signatures, kwargs bag (the curriculum dict is splatted into every call), return shapes and RNG draw order.

What differs is everything underneath: per call the host draws the random tensors with torch (reference
order), builds rays (origins / dirs / jittered z), runs the tiny mapping networks in
PyTorch, and hands the rest -- coarse SIREN, compositing, importance resampling, fine SIREN, sorted merge,
final compositing (generators.py:479-519) -- to ONE C-ABI call, fenerf_render_forward, i.e. hand-written HIP
kernels for gfx950.  There is no PyTorch fallback for that part.

Under torch.no_grad() (both D-steps, FID dumps, every inference script) a render is that one fused call.  With grad
enabled (generator step, inversion) forward / forward_with_frequencies run the differentiable composition of the same
kernels (_render_grad): fenerf_siren_forward_save + fenerf_siren_backward for the two SIREN passes and
fenerf_merge_composite + fenerf_composite_backward for the final integration, wired into torch autograd.
"""
import torch
import torch.nn as nn

from .. import _lib, native
from ..siren import autograd as _siren_autograd
from .autograd import (CompositeFunction, HierarchicalRenderFunction, ImageLayoutFunction, MergeCompositeFunction, SparseHierarchicalRenderFunction,
                       SparseSinglePassRenderFunction, hierarchical_render_split, sparse_auto_choice)
from . import volumetric_rendering as VR
from .volumetric_rendering import _DEFAULT_DRAWS, sample_rays


def _rng_state(device):
    device = torch.device(device) if device is not None else torch.device("cpu")
    return torch.cuda.get_rng_state(device) if device.type == "cuda" else torch.get_rng_state()


def _set_rng_state(device, state):
    device = torch.device(device) if device is not None else torch.device("cpu")
    if device.type == "cuda":
        torch.cuda.set_rng_state(state, device)
    else:
        torch.set_rng_state(state)


def _avg_cache_lookup(gen, nets):
    """generate_avg_frequencies cache (see DoubleImplicitGenerator3d.generate_avg_frequencies).  Key: the mapping networks'
    parameter versions / storages, the device, and the device generator's state before the draws.  Only with the default
    random source (recorded draws in tests are replayed as given).  Writes through `param.data` (torch_ema's copy_to / restore)
    bypass version counters -- the same caveat as for the packed render weights, and the same remedy: the generator's train() /
    eval() (which the reference calls around every EMA swap) and invalidate_native() drop this cache too."""
    gen.__dict__.pop("_avg_pending", None)
    if not isinstance(gen.draws, VR.TorchDraws):
        return None
    dev = gen.siren.device
    params = [p for net in nets for p in net.parameters()]
    key = (str(dev), tuple((p._version, p.data_ptr()) for p in params), _rng_state(dev).numpy().tobytes())
    cached = gen.__dict__.get("_avg_cache")
    if cached is not None and cached[0] == key:
        _set_rng_state(dev, cached[1])
        return cached[2]
    gen.__dict__["_avg_pending"] = key
    return None


def _avg_cache_store(gen, values):
    key = gen.__dict__.pop("_avg_pending", None)
    if key is not None:
        gen.__dict__["_avg_cache"] = (key, _rng_state(gen.siren.device).clone(), values)


class _Generator3dBase(nn.Module):
    """Shared host-side orchestration of one render (a15-a17 of SURVEY §8)."""

    draws = _DEFAULT_DRAWS   # random source; tests replace it with RecordedDraws

    # ---- helpers -------------------------------------------------------------------------------------
    def __getstate__(self):            # pickled generators (the reference's checkpoint format) carry no caches
        st = self.__dict__.copy()
        st.pop("_avg_cache", None)
        st.pop("_avg_pending", None)
        return st

    def invalidate_native(self):
        """Weights may have changed behind the version counters (writes through `param.data`): drop the cached average FiLM
        parameters and force a re-pack of the SIREN's native models at the next render."""
        self.__dict__.pop("_avg_cache", None)
        self.__dict__.pop("_avg_pending", None)
        if hasattr(self.siren, "invalidate_native"):
            self.siren.invalidate_native()

    def train(self, mode=True):
        # a mode switch = "weights may have changed" (train_double_latent_semantic.py:487-489, :522 -> :267 bracket every EMA swap
        # with eval() / train()); nn.Module.train recurses into self.siren, whose own train() invalidates the packed weights
        self.__dict__.pop("_avg_cache", None)
        self.__dict__.pop("_avg_pending", None)
        return super().train(mode)

    def _render(self, film, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean, v_mean,
                hierarchical_sample, sample_dist, lock_view_dependence, kwargs, use_fill, third):
        """film = (freq_geo, phase_geo, freq_app, phase_app) raw mapping outputs.
        third: None | 'weights' | 'auto' (what fancy_integration would have returned third for this fill_mode).
        Returns (pixels [B,R,C'], depth [B,R], third tensor or None, pitch, yaw)."""
        B = film[0].shape[0]
        dev = self.device
        R, N = img_size * img_size, num_steps
        d = self.draws
        # draw order: jitter rand [B,R,N,1] -> theta randn [B,1] -> phi randn [B,1]   (volumetric_rendering.py:135,193-194)
        origins, dirs, z_vals, pitch, yaw = sample_rays(B, N, dev, fov, (img_size, img_size), ray_start, ray_end, h_stddev,
                                                        v_stddev, h_mean, v_mean, sample_dist, draws=d)
        noise_std = kwargs["nerf_noise"]
        clamp_mode = kwargs["clamp_mode"]
        opts = _lib.composite_opts(clamp_mode, noise_std, kwargs.get("last_back", False), kwargs.get("white_back", False),
                                   kwargs.get("black_back", False), kwargs.get("fill_mode", None) if use_fill else None,
                                   kwargs.get("fill_color", "black"))
        u = noise_c = None
        if hierarchical_sample:
            # -> coarse noise randn [B,R,N,1] (always drawn, volumetric_rendering.py:27) -> u rand [B*R,N] (:283)
            noise_c = d.randn((B, R, N, 1), dev)
            u = d.rand((B * R, N), dev)
        M = 2 * N if hierarchical_sample else N
        noise_f = d.randn((B, R, M, 1), dev)   # final composite's noise
        use_noise = noise_std != 0
        fill_mode = kwargs.get("fill_mode", None) if use_fill else None
        want_w = third == "weights" or (third == "auto" and fill_mode not in ("weight", "eval_seg_padding_background", "eval_white_back"))
        want_ws = third == "auto" and not want_w
        nat = self.siren.native(dev)
        rgb, depth, weights, wsum = nat.render(
            origins, dirs, z_vals, u, noise_c.reshape(B, R, N) if (use_noise and noise_c is not None) else None,
            noise_f.reshape(B, R, M) if use_noise else None, film[0], film[1], film[2], film[3], opts,
            hierarchical=bool(hierarchical_sample), lock_view=bool(lock_view_dependence), want_weights=want_w, want_wsum=want_ws)
        t = None
        if want_w:
            t = weights.unsqueeze(-1)
        elif want_ws:
            t = wsum.unsqueeze(-1).expand_as(rgb)
        return rgb, depth, t, pitch, yaw

    def _wants_grad(self, film):
        return torch.is_grad_enabled() and (any(t.requires_grad for t in film) or
                                            any(p.requires_grad for p in self.siren._render_params()))

    def _render_grad(self, film, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean, v_mean,
                     hierarchical_sample, sample_dist, lock_view_dependence, kwargs):
        """The differentiable render of generators.py:468-519: same random draws in the same order as _render, same kernels,
        but the two SIREN passes and the final integration are autograd nodes with native backward kernels.  The coarse
        weights and the resampled depths are constants (computed under no_grad in the reference too, :485-503).
        With kwargs['grad_points'] < img_size^2 only a random subset of the rays is differentiable (part_forward, :858-910).
        Returns (pixels [B,R,C-1], depth [B,R] or None, pitch, yaw)."""
        B = film[0].shape[0]
        dev = self.device
        R, N = img_size * img_size, num_steps
        d = self.draws
        origins, dirs, z_vals, pitch, yaw = sample_rays(B, N, dev, fov, (img_size, img_size), ray_start, ray_end, h_stddev,
                                                        v_stddev, h_mean, v_mean, sample_dist, draws=d)
        z_c = z_vals.reshape(B, R, N)
        grad_points = kwargs.get("grad_points", R)
        if grad_points == R:
            rgb, depth = self._render_rays(film, origins, dirs, z_c, hierarchical_sample, lock_view_dependence, kwargs,
                                           self._wants_grad(film))
            return rgb, depth, pitch, yaw
        # part_forward: randperm AFTER the camera draws, then the gradient part's draws, then the rest's (:880-900)
        assert R > grad_points
        perm = d.randperm(R, dev)
        idx_g, idx_n = perm[:grad_points], perm[grad_points:]
        take = lambda t, idx: t[:, idx].contiguous()
        rgb_g, _ = self._render_rays(film, take(origins, idx_g), take(dirs, idx_g), take(z_c, idx_g), hierarchical_sample,
                                     lock_view_dependence, kwargs, self._wants_grad(film))
        with torch.no_grad():
            rgb_n, _ = self._render_rays(film, take(origins, idx_n), take(dirs, idx_n), take(z_c, idx_n), hierarchical_sample,
                                         lock_view_dependence, kwargs, False)
        pixels = torch.zeros((B, R, rgb_g.shape[-1]), dtype=rgb_g.dtype, device=dev)
        pixels = pixels.index_copy(1, idx_g, rgb_g).index_copy(1, idx_n, rgb_n)
        return pixels, None, pitch, yaw

    def _render_rays(self, film, origins, dirs, z_c, hierarchical_sample, lock_view_dependence, kwargs, differentiable):
        """origins / dirs [B,R,3], z_c [B,R,N] -> (rgb [B,R,C-1], depth [B,R]).  Draws: coarse noise randn [B,R,N,1] and u rand
        [B*R,N] (hierarchical only), final noise randn [B,R,M,1].  differentiable=False is the fused no-grad call."""
        fg, pg, fa, pa = film
        B, R, N = z_c.shape
        dev = self.device
        d = self.draws
        noise_std = kwargs["nerf_noise"]
        opts = _lib.composite_opts(kwargs["clamp_mode"], noise_std, kwargs.get("last_back", False), kwargs.get("white_back", False),
                                   kwargs.get("black_back", False), None, kwargs.get("fill_color", "black"))
        u = noise_c = None
        if hierarchical_sample:
            noise_c = d.randn((B, R, N, 1), dev)
            u = d.rand((B * R, N), dev)
        M = 2 * N if hierarchical_sample else N
        noise_f = d.randn((B, R, M, 1), dev)
        use_noise = noise_std != 0
        C = self.siren.output_dim
        if not differentiable:
            rgb, depth, _, _ = self.siren.native(dev).render(
                origins, dirs, z_c, u, noise_c.reshape(B, R, N) if (use_noise and noise_c is not None) else None,
                noise_f.reshape(B, R, M) if use_noise else None, fg, pg, fa, pa, opts, hierarchical=bool(hierarchical_sample),
                lock_view=bool(lock_view_dependence))
            return rgb, depth

        def field(z):      # [B,R,N] depths -> [B,R*N,C] radiance-field samples, an autograd node
            pts = origins.unsqueeze(2) + dirs.unsqueeze(2) * z.unsqueeze(-1)              # generators.py:504
            rd = None if lock_view_dependence else dirs.unsqueeze(2).expand(-1, -1, N, -1).reshape(B, R * N, 3)
            return _siren_autograd.siren_apply(self.siren, pts.reshape(B, R * N, 3), rd, fg, pg, fa, pa)

        sparse = getattr(self.siren, "sparse_backward", False)
        if sparse == "auto":
            sparse = sparse_auto_choice(self.siren, B * R * N * (2 if hierarchical_sample else 1))
        elif sparse not in (True, False):
            raise ValueError(f"siren.sparse_backward must be True, False or 'auto', got {sparse!r}")
        if not hierarchical_sample:
            if sparse:      # opt-in exact-sparsity backward (autograd.py), the render without importance resampling: the reference's inversion renders
                return SparseSinglePassRenderFunction.apply(self.siren, opts, None, bool(lock_view_dependence), origins, dirs, z_c, None, None,
                                                            noise_f.reshape(B * R, M) if use_noise else None, fg, pg, fa, pa,
                                                            *self.siren._render_params())
            coarse = field(z_c)
            rgb, depth = CompositeFunction.apply(coarse.reshape(B * R, N, C), z_c.reshape(B * R, N),
                                                 noise_f.reshape(B * R, M) if use_noise else None, opts)
            return rgb.reshape(B, R, C - 1), depth.reshape(B, R)
        # both SIREN passes + the merged composite as one autograd node (one chain launch and one set of weight-gradient
        # launches for the two passes)
        copts = _lib.composite_opts(kwargs["clamp_mode"], noise_std)
        nc_, nf_ = (noise_c.reshape(B * R, N) if use_noise else None), (noise_f.reshape(B * R, M) if use_noise else None)
        params = self.siren._render_params()
        grid = self.siren._roles(params)["grid"]
        if sparse:
            # opt-in: the backward runs only over the samples whose upstream gradient row is not all zero (autograd.py: exact).  It goes before
            # the two-node form below: what that form hides behind the weight-gradient kernels (the grid's all-reduce, ~1.5 ms) is less than
            # what this one removes, and its backward is short
            return SparseHierarchicalRenderFunction.apply(self.siren, opts, copts, bool(lock_view_dependence), origins, dirs, z_c, u, nc_, nf_, fg, pg,
                                                          fa, pa, *params)
        if getattr(self.siren, "split_backward", False) and grid is not None and grid.requires_grad and \
                any(p.requires_grad for p in params if p is not grid):
            # two autograd nodes: the grid gradient reaches DistributedDataParallel before the weight-gradient kernels run (autograd.py)
            return hierarchical_render_split(self.siren, opts, copts, bool(lock_view_dependence), origins, dirs, z_c, u, nc_, nf_, fg, pg, fa, pa)
        return HierarchicalRenderFunction.apply(self.siren, opts, copts, bool(lock_view_dependence), origins, dirs, z_c, u, nc_, nf_, fg, pg, fa, pa,
                                                *params)

    def _finish(self, pixels, batch_size, img_size):
        if self.softmax_label:
            seg, rgb = pixels[..., :-3], pixels[..., -3:]
            seg = torch.nn.Softmax(dim=-1)(seg)
            pixels = torch.cat([seg, rgb], dim=-1)
        pixels = pixels.reshape((batch_size, img_size, img_size, -1))
        return pixels.permute(0, 3, 1, 2).contiguous()

    def _finish_scaled(self, pixels, batch_size, img_size):
        """_finish(...) * 2 - 1 on the device, in one launch (and one in backward) when there is no softmax in between"""
        if self.softmax_label or not pixels.is_cuda or pixels.dtype != torch.float32:
            return self._finish(pixels, batch_size, img_size) * 2 - 1
        return ImageLayoutFunction.apply(pixels, batch_size, img_size)


class DoubleImplicitGenerator3d(_Generator3dBase):
    """Two-latent generator (z_geo, z_app) -> 18 semantic logits + rgb (generators.py:434-910)."""

    def __init__(self, siren, z_geo_dim, z_app_dim, output_dim, softmax_label=False, **kwargs):
        super().__init__()
        self.z_geo_dim = z_geo_dim
        self.z_app_dim = z_app_dim
        self.output_dim = output_dim
        self.siren = siren(output_dim=self.output_dim, z_geo_dim=self.z_geo_dim, z_app_dim=self.z_app_dim, input_dim=3, device=None)
        self.epoch = 0
        self.step = 0
        self.channel_dim = self.output_dim - 1
        self.softmax_label = softmax_label

    def set_device(self, device):
        self.device = device
        self.siren.device = device
        self.generate_avg_frequencies()

    def generate_avg_frequencies(self):
        """Mean FiLM parameters over 10 000 random latents (generators.py:530-543); draws randn twice.
        The reference redoes this on EVERY staged_forward.  It is a pure function of (mapping-network weights, generator state
        before the two draws), so the result is cached under exactly that key and, on a hit, the generator is advanced to the
        state the draws would have left it in: a seeded caller (render_multiview_images_double_semantic.py:77 re-seeds per view)
        gets bit-identical values and identical RNG consumption without re-running 2 x 10 000 latents."""
        hit = _avg_cache_lookup(self, (self.siren.geo_mapping_network, self.siren.app_mapping_network))
        if hit is not None:
            (self.avg_frequencies_geo, self.avg_phase_shifts_geo, self.avg_frequencies_app, self.avg_phase_shifts_app) = hit
            return hit
        z_geo = self.draws.randn((10000, self.z_geo_dim), self.siren.device)
        z_app = self.draws.randn((10000, self.z_app_dim), self.siren.device)
        with torch.no_grad():
            frequencies_geo, phase_shifts_geo = self.siren.geo_mapping_network(z_geo)
            frequencies_app, phase_shifts_app = self.siren.app_mapping_network(z_app)
        self.avg_frequencies_geo = frequencies_geo.mean(0, keepdim=True)
        self.avg_phase_shifts_geo = phase_shifts_geo.mean(0, keepdim=True)
        self.avg_frequencies_app = frequencies_app.mean(0, keepdim=True)
        self.avg_phase_shifts_app = phase_shifts_app.mean(0, keepdim=True)
        out = (self.avg_frequencies_geo, self.avg_phase_shifts_geo, self.avg_frequencies_app, self.avg_phase_shifts_app)
        _avg_cache_store(self, out)
        return out

    def forward(self, z_geo, z_app, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean, v_mean,
                hierarchical_sample, sample_dist=None, lock_view_dependence=False, **kwargs):
        """-> (pixels [B, output_dim-1, S, S] in [-1,1], cat(pitch, yaw) [B,2])   (generators.py:452-527)."""
        batch_size = z_app.shape[0]
        fg, pg = self.siren.geo_mapping_network(z_geo)
        fa, pa = self.siren.app_mapping_network(z_app)
        if kwargs.get("grad_points", img_size * img_size) != img_size * img_size:
            # generators.py:459-461 -- NB the reference drops the caller's sample_dist / lock_view_dependence on this path
            # (camera at the mean pose, view dependence on); kept as is
            return self.part_forward(z_geo, z_app, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean, v_mean,
                                     hierarchical_sample, sample_dist=None, lock_view_dependence=False, **kwargs)
        if self._wants_grad((fg, pg, fa, pa)):
            pixels, depth, pitch, yaw = self._render_grad((fg, pg, fa, pa), img_size, fov, ray_start, ray_end, num_steps, h_stddev,
                                                          v_stddev, h_mean, v_mean, hierarchical_sample, sample_dist,
                                                          lock_view_dependence, kwargs)
            return self._finish_scaled(pixels, batch_size, img_size), torch.cat([pitch, yaw], -1)
        # forward() ignores fill_mode (generators.py:519) -> C-1 channels
        pixels, depth, _, pitch, yaw = self._render((fg, pg, fa, pa), img_size, fov, ray_start, ray_end, num_steps, h_stddev,
                                                    v_stddev, h_mean, v_mean, hierarchical_sample, sample_dist,
                                                    lock_view_dependence, kwargs, use_fill=False, third=None)
        pixels = self._finish_scaled(pixels, batch_size, img_size)
        return pixels, torch.cat([pitch, yaw], -1)

    def part_forward(self, z_geo, z_app, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean, v_mean,
                     hierarchical_sample, sample_dist=None, lock_view_dependence=False, **kwargs):
        """Gradient on kwargs['grad_points'] random rays only, the rest rendered without (generators.py:858-910)."""
        batch_size = z_app.shape[0]
        fg, pg = self.siren.geo_mapping_network(z_geo)
        fa, pa = self.siren.app_mapping_network(z_app)
        pixels, _, pitch, yaw = self._render_grad((fg, pg, fa, pa), img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev,
                                                  h_mean, v_mean, hierarchical_sample, sample_dist, lock_view_dependence, kwargs)
        return self._finish_scaled(pixels, batch_size, img_size), torch.cat([pitch, yaw], -1)

    def point_forward(self, transformed_points, transformed_ray_directions_expanded, transformed_ray_origins,
                      transformed_ray_directions, z_vals, z_geo, z_app, num_steps, hierarchical_sample,
                      lock_view_dependence=False, **kwargs):
        """Render explicit samples: points / per-point view directions [B,G,N,3], ray origins / directions [B,G,3], z_vals
        [B,G,N,1] -> pixels [B,G,C-1] (generators.py:800-855).  Stage by stage on the native kernels (per-point view
        directions are honoured, as in the reference); differentiable when grad is enabled."""
        B, Gn = transformed_points.shape[:2]
        N, C = num_steps, self.siren.output_dim
        d = self.draws
        fg, pg = self.siren.geo_mapping_network(z_geo)
        fa, pa = self.siren.app_mapping_network(z_app)
        noise_std = kwargs["nerf_noise"]
        opts = _lib.composite_opts(kwargs["clamp_mode"], noise_std, kwargs.get("last_back", False), kwargs.get("white_back", False),
                                   kwargs.get("black_back", False), None, kwargs.get("fill_color", "black"))
        rd = transformed_ray_directions_expanded.reshape(B, Gn * N, 3)
        siren = self.siren.forward_with_frequencies_phase_shifts
        coarse = siren(transformed_points.reshape(B, Gn * N, 3), fg, fa, pg, pa, rd)
        z_c = z_vals.reshape(B * Gn, N)
        use_noise = noise_std != 0
        if hierarchical_sample:
            noise_c = d.randn((B, Gn, N, 1), transformed_points.device)
            u = d.rand((B * Gn, N), transformed_points.device)
            with torch.no_grad():
                _, _, w_c, _ = native.composite(coarse.detach().reshape(B * Gn, N, C), z_c, noise_c.reshape(B * Gn, N) if use_noise else None,
                                                _lib.composite_opts(kwargs["clamp_mode"], noise_std), want_wsum=False)
                z_f = native.resample(z_c, w_c, u)
                fine_pts = transformed_ray_origins.unsqueeze(2) + transformed_ray_directions.unsqueeze(2) * z_f.reshape(B, Gn, N, 1)
                if lock_view_dependence:
                    rd = torch.zeros_like(rd)
                    rd[..., -1] = -1
            fine = siren(fine_pts.reshape(B, Gn * N, 3), fg, fa, pg, pa, rd)
            noise_f = d.randn((B, Gn, 2 * N, 1), transformed_points.device)
            rgb, _ = MergeCompositeFunction.apply(fine.reshape(B * Gn, N, C), coarse.reshape(B * Gn, N, C), z_f, z_c,
                                                  noise_f.reshape(B * Gn, 2 * N) if use_noise else None, opts)
        else:
            noise_f = d.randn((B, Gn, N, 1), transformed_points.device)
            rgb, _ = CompositeFunction.apply(coarse.reshape(B * Gn, N, C), z_c, noise_f.reshape(B * Gn, N) if use_noise else None, opts)
        pixels = rgb.reshape(B, Gn, C - 1)
        if self.softmax_label:
            pixels = torch.cat([torch.nn.Softmax(dim=-1)(pixels[..., :-3]), pixels[..., -3:]], dim=-1)
        return pixels

    def staged_forward(self, z_geo, z_app, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean, v_mean,
                       psi=1, lock_view_dependence=False, max_batch_size=50000, depth_map=False, near_clip=0, far_clip=2,
                       sample_dist=None, hierarchical_sample=False, **kwargs):
        """Inference render with the truncation trick -> (pixels.cpu(), depth_map.cpu())   (generators.py:546-646).
        max_batch_size is accepted and ignored: the fused kernel never materialises per-point activations."""
        batch_size = z_app.shape[0]
        self.generate_avg_frequencies()
        with torch.no_grad():
            raw_fg, raw_pg = self.siren.geo_mapping_network(z_geo)
            raw_fa, raw_pa = self.siren.app_mapping_network(z_app)
            fg = self.avg_frequencies_geo + psi * (raw_fg - self.avg_frequencies_geo)
            pg = self.avg_phase_shifts_geo + psi * (raw_pg - self.avg_phase_shifts_geo)
            fa = self.avg_frequencies_app + psi * (raw_fa - self.avg_frequencies_app)
            pa = self.avg_phase_shifts_app + psi * (raw_pa - self.avg_phase_shifts_app)
            pixels, depth, _, pitch, yaw = self._render((fg, pg, fa, pa), img_size, fov, ray_start, ray_end, num_steps, h_stddev,
                                                        v_stddev, h_mean, v_mean, hierarchical_sample, sample_dist,
                                                        lock_view_dependence, kwargs, use_fill=True, third=None)
            depth_map = native.to_host(depth.reshape(batch_size, img_size, img_size).contiguous())
            # `.cpu() * 2 - 1` of the reference: the same two fp32 operations, on the device (one launch), then one copy through pinned memory
            pixels = native.to_host(self._finish_scaled(pixels, batch_size, img_size))
        return pixels, depth_map

    def staged_forward_with_frequencies(self, truncated_frequencies_geo, truncated_frequencies_app, truncated_phase_shifts_geo,
                                        truncated_phase_shifts_app, img_size, fov, ray_start, ray_end, num_steps, h_stddev,
                                        v_stddev, h_mean, v_mean, psi=0.7, lock_view_dependence=False, max_batch_size=50000,
                                        depth_map=False, near_clip=0, far_clip=2, sample_dist=None, hierarchical_sample=False,
                                        **kwargs):
        """-> (pixels.cpu(), depth.cpu(), third.cpu()*2-1) (generators.py:649-732); `third` is weights_sum for the
        eval_*/weight fill modes and the per-sample weights [B,M,S,S] otherwise (reference quirk)."""
        batch_size = truncated_frequencies_app.shape[0]
        with torch.no_grad():
            pixels, depth, third, pitch, yaw = self._render(
                (truncated_frequencies_geo, truncated_phase_shifts_geo, truncated_frequencies_app, truncated_phase_shifts_app),
                img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean, v_mean, hierarchical_sample,
                sample_dist, lock_view_dependence, kwargs, use_fill=True, third="auto")
            depth_map = native.to_host(depth.reshape(batch_size, img_size, img_size).contiguous())
            weights_sum = third.reshape((batch_size, img_size, img_size, -1))
            weights_sum = native.to_host(weights_sum.permute(0, 3, 1, 2).contiguous() * 2 - 1)
            pixels = native.to_host(self._finish_scaled(pixels, batch_size, img_size))
        return pixels, depth_map, weights_sum

    def forward_with_frequencies(self, frequencies_geo, frequencies_app, phase_shifts_geo, phase_shifts_app, img_size, fov,
                                 ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean, v_mean, hierarchical_sample,
                                 sample_dist=None, lock_view_dependence=False, **kwargs):
        """-> (pixels [B, output_dim-1, S, S], poses)   (generators.py:735-797)."""
        batch_size = frequencies_app.shape[0]
        film = (frequencies_geo, phase_shifts_geo, frequencies_app, phase_shifts_app)
        if self._wants_grad(film):
            pixels, depth, pitch, yaw = self._render_grad(film, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev,
                                                          h_mean, v_mean, hierarchical_sample, sample_dist, lock_view_dependence,
                                                          kwargs)
            return self._finish_scaled(pixels, batch_size, img_size), torch.cat([pitch, yaw], -1)
        pixels, depth, _, pitch, yaw = self._render((frequencies_geo, phase_shifts_geo, frequencies_app, phase_shifts_app),
                                                    img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean,
                                                    v_mean, hierarchical_sample, sample_dist, lock_view_dependence, kwargs,
                                                    use_fill=False, third=None)
        pixels = self._finish_scaled(pixels, batch_size, img_size)
        return pixels, torch.cat([pitch, yaw], -1)


class ImplicitGenerator3d(_Generator3dBase):
    """Single-latent pi-GAN generator (generators.py:13-431), used with SPATIALSIRENBASELINE (curriculum `CelebA`)."""

    def __init__(self, siren, z_dim, output_dim, neural_renderer_img=None, neural_renderer_seg=None, softmax_label=False, **kwargs):
        super().__init__()
        if neural_renderer_img is not None or neural_renderer_seg is not None:
            raise NotImplementedError("2-D neural renderers are outside the rendering-core scope (SURVEY §2)")
        self.z_dim = z_dim
        self.output_dim = output_dim
        self.siren = siren(output_dim=self.output_dim, z_dim=self.z_dim, input_dim=3, device=None)
        self.epoch = 0
        self.step = 0
        self.channel_dim = self.output_dim - 1
        self.softmax_label = softmax_label
        self.neural_renderer_img = None
        self.neural_renderer_seg = None

    def set_device(self, device):
        self.device = device
        self.siren.device = device
        self.generate_avg_frequencies()

    def generate_avg_frequencies(self):
        """(generators.py:121-130); cached like DoubleImplicitGenerator3d.generate_avg_frequencies"""
        hit = _avg_cache_lookup(self, (self.siren.mapping_network,))
        if hit is not None:
            self.avg_frequencies, self.avg_phase_shifts = hit
            return hit
        z = self.draws.randn((10000, self.z_dim), self.siren.device)
        with torch.no_grad():
            frequencies, phase_shifts = self.siren.mapping_network(z)
        self.avg_frequencies = frequencies.mean(0, keepdim=True)
        self.avg_phase_shifts = phase_shifts.mean(0, keepdim=True)
        _avg_cache_store(self, (self.avg_frequencies, self.avg_phase_shifts))
        return self.avg_frequencies, self.avg_phase_shifts

    def _film(self, frequencies, phase_shifts):
        return self.siren.split_film(frequencies, phase_shifts)

    def _staged_film(self, z, psi):
        """raw FiLM parameters staged_forward renders with: truncated towards the average ones (generators.py:158-165)"""
        self.generate_avg_frequencies()
        raw_frequencies, raw_phase_shifts = self.siren.mapping_network(z)
        return (self.avg_frequencies + psi * (raw_frequencies - self.avg_frequencies),
                self.avg_phase_shifts + psi * (raw_phase_shifts - self.avg_phase_shifts))

    def forward(self, z, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean, v_mean, hierarchical_sample,
                sample_dist=None, lock_view_dependence=False, **kwargs):
        """-> (pixels [B,3,S,S], poses)   (generators.py:32-119)."""
        batch_size = z.shape[0]
        frequencies, phase_shifts = self.siren.mapping_network(z)
        film = self._film(frequencies, phase_shifts)
        if self._wants_grad(film):
            pixels, depth, pitch, yaw = self._render_grad(film, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev,
                                                          h_mean, v_mean, hierarchical_sample, sample_dist, lock_view_dependence,
                                                          kwargs)
            return self._finish_scaled(pixels, batch_size, img_size), torch.cat([pitch, yaw], -1)
        pixels, depth, _, pitch, yaw = self._render(film, img_size, fov, ray_start, ray_end,
                                                    num_steps, h_stddev, v_stddev, h_mean, v_mean, hierarchical_sample,
                                                    sample_dist, lock_view_dependence, kwargs, use_fill=False, third=None)
        pixels = self._finish_scaled(pixels, batch_size, img_size)
        return pixels, torch.cat([pitch, yaw], -1)

    def staged_forward(self, z, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean, v_mean, psi=1,
                       lock_view_dependence=False, max_batch_size=50000, depth_map=False, near_clip=0, far_clip=2,
                       sample_dist=None, hierarchical_sample=False, **kwargs):
        """-> (pixels.cpu(), depth_map.cpu(), weights_sum.cpu()*2-1): the single-latent class returns 3 values
        (generators.py:132-248)."""
        batch_size = z.shape[0]
        with torch.no_grad():
            f, p = self._staged_film(z, psi)
            pixels, depth, third, pitch, yaw = self._render(self._film(f, p), img_size, fov, ray_start, ray_end, num_steps,
                                                            h_stddev, v_stddev, h_mean, v_mean, hierarchical_sample, sample_dist,
                                                            lock_view_dependence, kwargs, use_fill=True, third="auto")
            depth_map = native.to_host(depth.reshape(batch_size, img_size, img_size).contiguous())
            weights_sum = native.to_host(third.reshape((batch_size, img_size, img_size, -1)).permute(0, 3, 1, 2).contiguous() * 2 - 1)
            pixels = self._finish_scaled(pixels, batch_size, img_size)   # stays on the device (generators.py:231)
        return pixels, depth_map, weights_sum

    def staged_forward_with_frequencies(self, truncated_frequencies, truncated_phase_shifts, img_size, fov, ray_start, ray_end,
                                        num_steps, h_stddev, v_stddev, h_mean, v_mean, psi=0.7, lock_view_dependence=False,
                                        max_batch_size=50000, depth_map=False, near_clip=0, far_clip=2, sample_dist=None,
                                        hierarchical_sample=False, **kwargs):
        """-> (pixels (device), depth_map.cpu()): two values for the single-latent class (generators.py:251-335)."""
        batch_size = truncated_frequencies.shape[0]
        with torch.no_grad():
            pixels, depth, _, pitch, yaw = self._render(self._film(truncated_frequencies, truncated_phase_shifts), img_size,
                                                        fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean, v_mean,
                                                        hierarchical_sample, sample_dist, lock_view_dependence, kwargs,
                                                        use_fill=True, third=None)
            depth_map = native.to_host(depth.reshape(batch_size, img_size, img_size).contiguous())
            pixels = self._finish_scaled(pixels, batch_size, img_size)
        return pixels, depth_map

    def forward_with_frequencies(self, frequencies, phase_shifts, img_size, fov, ray_start, ray_end, num_steps, h_stddev,
                                 v_stddev, h_mean, v_mean, hierarchical_sample, sample_dist=None, lock_view_dependence=False,
                                 **kwargs):
        """(generators.py:353-431)"""
        batch_size = frequencies.shape[0]
        film = self._film(frequencies, phase_shifts)
        if self._wants_grad(film):
            pixels, depth, pitch, yaw = self._render_grad(film, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev,
                                                          h_mean, v_mean, hierarchical_sample, sample_dist, lock_view_dependence,
                                                          kwargs)
            return self._finish_scaled(pixels, batch_size, img_size), torch.cat([pitch, yaw], -1)
        pixels, depth, _, pitch, yaw = self._render(film, img_size, fov, ray_start, ray_end,
                                                    num_steps, h_stddev, v_stddev, h_mean, v_mean, hierarchical_sample,
                                                    sample_dist, lock_view_dependence, kwargs, use_fill=False, third=None)
        pixels = self._finish_scaled(pixels, batch_size, img_size)
        return pixels, torch.cat([pitch, yaw], -1)


class StyleGenerator3d(ImplicitGenerator3d):
    """The reference's third generator class (generators.py:914-1294; no curriculum or script of the reference instantiates it): a copy
    of ImplicitGenerator3d WITHOUT average frequencies -- `set_device` draws no 10,000 latents, `staged_forward` evaluates
    `self.siren(points, z, ray_directions)` (generators.py:1042, :1071), i.e. the raw mapping-network outputs: `psi` is accepted and
    ignored -- and whose two staged methods do not pass `fill_color` on to fancy_integration (:1084, :1188: its default, 'black').
    Same four methods, return shapes and RNG draw order otherwise; a pickled one loads through compat.install_aliases()."""

    def set_device(self, device):
        self.device = device
        self.siren.device = device

    def generate_avg_frequencies(self):
        raise AttributeError("'StyleGenerator3d' object has no attribute 'generate_avg_frequencies'")      # as in the reference: the class has none

    def _staged_film(self, z, psi):
        return self.siren.mapping_network(z)

    def staged_forward(self, z, *args, **kwargs):
        kwargs.pop("fill_color", None)
        return super().staged_forward(z, *args, **kwargs)

    def staged_forward_with_frequencies(self, truncated_frequencies, truncated_phase_shifts, *args, **kwargs):
        kwargs.pop("fill_color", None)
        return super().staged_forward_with_frequencies(truncated_frequencies, truncated_phase_shifts, *args, **kwargs)
