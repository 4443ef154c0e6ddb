#!/usr/bin/env python
"""Multi-view renders of a trained generator on the MI355X-native path -- the command-line surface of the reference's
render_multiview_images_double_semantic.py (:31-42), same positional argument and options:

    python tools/render_multiview.py <path/to/generator.pth> --curriculum CelebA_double_semantic_texture_embedding_256_dim_96 \\
           --seeds 0 1 2 --output_dir imgs [--image_size 256] [--ray_step_multiplier 2] [--lock_view_dependence] [--max_batch_size N]

For every seed: five yaw angles (-0.5 .. 0.5 rad around h_mean) of the identity drawn from `torch.manual_seed(seed)`, written as
`grid_<seed>_RGB.png` (normalised from [-1, 1]) and `grid_<seed>_SEG.png` (argmax -> colour LUT, already in [0, 1]) with
torchvision.save_image's grid layout.  `<prefix>ema.pth` next to the generator pickle is loaded and copied in, as the reference
does (:61-64).  `--max_batch_size` is accepted and ignored: the fused renderer never materialises per-point activations.
(The reference's SEG line passes the misspelt `noralize=True`, which torchvision 0.9's save_image rejects; the colour maps are
in [0, 1] already, so this writes them unnormalised -- what that line means.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('path', type=str)
    parser.add_argument('--seeds', nargs='+', default=[0])
    parser.add_argument('--output_dir', type=str, default='imgs')
    parser.add_argument('--max_batch_size', type=int, default=2400000)
    parser.add_argument('--lock_view_dependence', action='store_true')
    parser.add_argument('--image_size', type=int, default=256)
    parser.add_argument('--ray_step_multiplier', type=int, default=2)
    parser.add_argument('--curriculum', type=str, default='CelebA')
    parser.add_argument('--no_ema', action='store_true', help='render the raw generator weights (not in the reference: it always loads <prefix>ema.pth)')
    return parser


def resolve_curriculum(name):
    """A curriculum of fenerf_amd.curriculums by name (the reference's behaviour), or -- an extension for tests and custom
    models -- a JSON file in tests/golden/curriculums.json's spelling (integer stage keys as "int:<step>")."""
    import json
    from fenerf_amd import curriculums
    if hasattr(curriculums, name):
        return dict(getattr(curriculums, name))
    if os.path.isfile(name):
        d = json.load(open(name))
        return {(int(k[4:]) if k.startswith("int:") else k): v for k, v in d.items()}
    raise SystemExit(f"unknown curriculum {name!r}")


def main(argv=None):
    from fenerf_amd import host
    host.respect_cpu_quota()          # torch's CPU thread pool no larger than the cores this process is granted (fenerf_amd/host.py)
    opt = build_parser().parse_args(argv)
    import torch
    from fenerf_amd import callers, imageio_lite
    if not torch.cuda.is_available():
        raise SystemExit("render_multiview.py renders on the GPU (fenerf_amd has no CPU path)")
    device = torch.device('cuda')
    curriculum = resolve_curriculum(opt.curriculum)
    os.makedirs(opt.output_dir, exist_ok=True)
    generator = callers.load_generator(opt.path, device, use_ema=not opt.no_ema)
    for seed in opt.seeds:
        images, segmaps = callers.render_multiview(generator, curriculum, int(seed), device, image_size=opt.image_size,
                                                   ray_step_multiplier=opt.ray_step_multiplier,
                                                   lock_view_dependence=opt.lock_view_dependence)
        imageio_lite.save_image(images, os.path.join(opt.output_dir, f'grid_{seed}_RGB.png'), normalize=True, value_range=(-1, 1))
        imageio_lite.save_image(segmaps, os.path.join(opt.output_dir, f'grid_{seed}_SEG.png'))
        print(f"seed {seed}: {tuple(images.shape)} -> {opt.output_dir}/grid_{seed}_RGB.png, grid_{seed}_SEG.png")


if __name__ == '__main__':
    main()
