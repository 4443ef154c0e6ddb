#!/usr/bin/env python
"""Density volumes for shape extraction on the MI355X-native path -- the command-line surface of the reference's
extract_double_semantic_shapes.py (:89-139; extract_shapes.py is its single-latent twin), same positional argument and options:

    python tools/extract_shapes.py <path/to/generator.pth> --seeds 3 4 5 --cube_size 0.3 --voxel_resolution 256 --output_dir shapes
    python tools/extract_shapes.py <path/to/generator.pth> --latent_path <save_dir>/freq_phase_offset_<name>.pth --seeds 7

Without --latent_path: for every seed the identity z = randn(1, z_dim) under torch.manual_seed(seed) (the script feeds the same z to both
mapping networks), FiLM parameters truncated towards the mean with psi = 0.5, sigma evaluated on the voxel_resolution^3 lattice of a
cube of side --cube_size with the view direction locked to (0, 0, -1) -> <output_dir>/<seed>.mrc.  With --latent_path: the identity of
an inversion checkpoint (mean + offsets; tools/inverse_render.py or the reference's script wrote it) -> <output_dir>/<seeds[0]>.mrc.
All N^3 points go through ONE fused SIREN launch (the reference walks them in chunks of 24,000 / 100,000).  The volume is written as an
MRC2014 mode-2 map by fenerf_amd.imageio_lite.write_mrc (mrcfile is not a dependency); marching cubes on it is the downstream step
the reference also leaves to other tools.  `<prefix>ema.pth` next to the generator pickle is loaded and copied in (:103-104).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('path', type=str)
    parser.add_argument('--seeds', nargs='+', default=[3, 4, 5])
    parser.add_argument('--cube_size', type=float, default=0.3)
    parser.add_argument('--voxel_resolution', type=int, default=256)
    parser.add_argument('--output_dir', type=str, default='shapes')
    parser.add_argument('--latent_path', type=str, default=None)
    parser.add_argument('--no_ema', action='store_true', help='use the raw generator weights (not in the reference: it always loads <prefix>ema.pth)')
    return parser


def main(argv=None):
    from fenerf_amd import host
    host.respect_cpu_quota()          # torch's CPU thread pool no larger than the cores this process is granted (fenerf_amd/host.py)
    opt = build_parser().parse_args(argv)
    import torch
    from fenerf_amd import callers, imageio_lite
    if not torch.cuda.is_available():
        raise SystemExit("extract_shapes.py evaluates on the GPU (fenerf_amd has no CPU path)")
    device = torch.device('cuda')
    generator = callers.load_generator(opt.path, device, use_ema=not opt.no_ema)
    os.makedirs(opt.output_dir, exist_ok=True)
    if opt.latent_path is None:
        z_dim = callers._latent_dims(generator)[0]
        for seed in opt.seeds:
            torch.manual_seed(int(seed))
            z = torch.randn(1, z_dim, device=device)
            voxel_grid = callers.sample_generator(generator, z, cube_length=opt.cube_size, voxel_resolution=opt.voxel_resolution)
            out = os.path.join(opt.output_dir, f'{seed}.mrc')
            imageio_lite.write_mrc(out, voxel_grid)
            print(f"seed {seed}: sigma {voxel_grid.shape} in [{voxel_grid.min():.3g}, {voxel_grid.max():.3g}] -> {out}")
    else:
        meta = torch.load(opt.latent_path, map_location=device, weights_only=False)
        fg, fa, pg, pa = callers.film_from_inversion(meta, device)
        meta = dict(meta, truncated_frequencies_geo=fg, truncated_frequencies_app=fa, truncated_phase_shifts_geo=pg, truncated_phase_shifts_app=pa)
        voxel_grid = callers.sample_generator_wth_frequencies_phase_shifts(generator, meta, cube_length=opt.cube_size,
                                                                           voxel_resolution=opt.voxel_resolution)
        out = os.path.join(opt.output_dir, f'{opt.seeds[0]}.mrc')
        imageio_lite.write_mrc(out, voxel_grid)
        print(f"inverted identity {opt.latent_path}: sigma {voxel_grid.shape} in [{voxel_grid.min():.3g}, {voxel_grid.max():.3g}] -> {out}")


if __name__ == '__main__':
    main()
