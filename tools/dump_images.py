#!/usr/bin/env python
"""The image dump of the reference's FID evaluation (fid_evaluation.py:96-150, called from train_double_latent_semantic.py:541-548) as
a command, on one GPU or -- the reference's way of running its forward path on several -- one process per GPU, each writing the
image ids rank, rank + world, ...:

    python tools/dump_images.py <path/to/generator.pth> --curriculum CelebA_double_semantic_texture_embedding_256_dim_96 \\
           --output_dir generated --num_imgs 2048 [--step 60000] [--no_ema]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/dump_images.py ... (same arguments)

Sharding is by image id, there is no data-path collective; the only collectives are the two barriers the reference puts around the
dump (train...py:542, :548).  `--one_device` puts every rank on cuda:0 (with `--dist_backend gloo`: the N > 1 path on a one-GPU box).
The FID number itself (pytorch_fid on the dumped files) is third-party code and not part of this package.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from render_multiview import resolve_curriculum      # noqa: E402  (same directory)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('path', type=str)
    parser.add_argument('--curriculum', type=str, default='CelebA')
    parser.add_argument('--output_dir', type=str, default='generated')
    parser.add_argument('--num_imgs', type=int, default=2048)
    parser.add_argument('--step', type=int, default=100000, help='training step whose curriculum stage supplies the render options (extract_metadata) '
                        'and the density noise max(0, 1 - step / 5000) the training loop carries at that step (train...py:276)')
    parser.add_argument('--seed', type=int, default=None, help='torch.manual_seed(seed + rank) before the dump (the reference dumps from the training RNG state)')
    parser.add_argument('--no_ema', action='store_true')
    parser.add_argument('--dist_backend', type=str, default='nccl')
    parser.add_argument('--one_device', action='store_true')
    return parser


def main(argv=None):
    from fenerf_amd import host
    host.respect_cpu_quota()          # torch's CPU thread pool no larger than the cores this process is granted (fenerf_amd/host.py)
    opt = build_parser().parse_args(argv)
    import torch
    import torch.distributed as dist
    from fenerf_amd import callers, curriculums, dist as fdist
    if not torch.cuda.is_available():
        raise SystemExit("dump_images.py renders on the GPU (fenerf_amd has no CPU path)")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device('cuda', 0 if opt.one_device else local_rank)
    torch.cuda.set_device(device)
    rank, _, world = fdist.init_from_env(backend=opt.dist_backend, device=device)
    curriculum = resolve_curriculum(opt.curriculum)
    metadata = curriculums.extract_metadata(curriculum, opt.step)
    metadata['nerf_noise'] = max(0., 1. - opt.step / 5000.)
    generator = callers.load_generator(opt.path, device, use_ema=not opt.no_ema, reset_render_options=False)
    if opt.seed is not None:
        torch.manual_seed(opt.seed + rank)
    dump = callers.output_images_double if hasattr(generator, 'z_geo_dim') else callers.output_images
    if world > 1:
        dist.barrier()
    written = dump(generator, metadata, rank, world, opt.output_dir, num_imgs=opt.num_imgs)
    if world > 1:
        dist.barrier()
    print(f"rank {rank} of {world}: {len(written)} images -> {opt.output_dir} ({os.path.basename(written[0]) if written else '-'} ... "
          f"{os.path.basename(written[-1]) if written else '-'})")
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
