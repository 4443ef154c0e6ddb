#!/usr/bin/env python
"""Latent-interpolation renders of a trained double-latent generator on the MI355X-native path -- the command-line surface of
the reference's render_video_interpolation_semantic.py (:25-47) for its `video_double_latent_interpolation` mode (:314-458):

    python tools/render_video_interpolation.py <path/to/generator.pth> --curriculum CelebA_double_semantic_texture_embedding_256_dim_96 \\
           --seeds 0 --latent_type geo --trajectory front --num_frames 36 --psi 0.5 [--save_with_video] ...

Per seed, in `<output_dir>/interpolation_<latent_type>_<seed>/`: `images/<latent_type>_<trajectory>/{img,label,acc,depth,
depth_color}_<j>.png` per frame and the strips `interp.png`, `interp_seg.png`, `interp_acc_map.png`, `interp_depth_map.png`
(or, with --save_with_video, `interp_<latent_type>_<seed>.avi`: frames [image | labels | blend | depth colours] side by side
at 25 fps -- uncompressed AVI, because neither cv2 nor skvideo is a dependency here; the reference writes the same frames as
mp4v through cv2.VideoWriter).  `--interpolation_type video_latent_interpolation` runs the single-latent ImplicitGenerator3d
variant (:187-312: its own trajectories, per-frame img_<j>.png + one strip, rgb frames in the video).  --max_batch_size / --batch_size / --seed_mode / --save_with_latent are accepted for
command-line compatibility; the fused renderer needs no chunking.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('path', type=str)
    parser.add_argument('--interpolation_type', type=str, default='video_double_latent_interpolation')
    parser.add_argument('--latent_type', default='geo')  # for double latent
    parser.add_argument('--seeds', nargs='+', default=[0])
    parser.add_argument('--output_dir', type=str, default='vids')
    parser.add_argument('--batch_size', type=int, default=1)
    parser.add_argument('--max_batch_size', type=int, default=2400000)
    parser.add_argument('--depth_map', action='store_true')
    parser.add_argument('--lock_view_dependence', action='store_true')
    parser.add_argument('--image_size', type=int, default=256)
    parser.add_argument('--ray_step_multiplier', type=int, default=2)
    parser.add_argument('--num_frames', type=int, default=36)
    parser.add_argument('--curriculum', type=str, default='CelebA')
    parser.add_argument('--no_ema', action='store_true', help='render the raw generator weights (not in the reference: it always loads <prefix>ema.pth)')
    parser.add_argument('--trajectory', type=str, default='front')
    parser.add_argument('--psi', type=float, default=0.5)
    parser.add_argument("--fill_color", type=str, default='black')
    parser.add_argument("--fov", type=int, default=12)
    parser.add_argument("--save_with_video", action='store_true')
    parser.add_argument("--save_with_latent", action='store_true')
    parser.add_argument("--seed_mode", default='single', type=str, help='if the seeds are speficifed a range or a number')
    return parser


def run_single_latent(opt, generator, options, device):
    """run_video_latent_interpolation (render_video_interpolation_semantic.py:187-312): rgb + sigma generator, one latent."""
    import numpy as np
    from fenerf_amd import callers, imageio_lite
    generator.output_dim, generator.channel_dim = 4, 3          # :189-190
    options = dict(options, output_dim=4)
    trajectory = callers.camera_trajectory_single(opt.trajectory, options['num_frames'], options['fov'])
    for i, seed in enumerate(opt.seeds):
        output_dir = os.path.join(opt.output_dir, f'interpolation_{opt.latent_type}_{seed}')
        frame_dir = os.path.join(output_dir, "images", f"{opt.latent_type}_{opt.trajectory}")
        os.makedirs(frame_dir, exist_ok=True)
        out = callers.render_latent_video(generator, int(seed), dict(options, max_batch_size=opt.max_batch_size, depth_map=opt.depth_map),
                                          trajectory, latent_type=opt.latent_type, psi=opt.psi, device=device)
        images = out["images"].cpu()
        for j in range(images.shape[0]):
            imageio_lite.save_image(images[j:j + 1], os.path.join(frame_dir, f"img_{j}.png"), nrow=1, normalize=True)
        imageio_lite.save_image(images, os.path.join(output_dir, f"{opt.interpolation_type}_img_{i}.png"), nrow=opt.num_frames, normalize=True)
        if opt.save_with_video:
            writer = imageio_lite.AviWriter(os.path.join(output_dir, f'interp_{opt.latent_type}_{seed}.avi'), fps=25)
            for j in range(images.shape[0]):
                writer.write(imageio_lite.to_uint8_hwc(imageio_lite.make_grid(images[j:j + 1], normalize=True)))
            writer.release()
        print(f"seed {seed}: {images.shape[0]} frames -> {output_dir}")


def main(argv=None):
    from fenerf_amd import host
    host.respect_cpu_quota()          # torch's CPU thread pool no larger than the cores this process is granted (fenerf_amd/host.py)
    opt = build_parser().parse_args(argv)
    import numpy as np
    import torch
    from fenerf_amd import callers, imageio_lite
    from render_multiview import resolve_curriculum
    if opt.interpolation_type not in ('video_double_latent_interpolation', 'video_latent_interpolation'):
        raise SystemExit(f"--interpolation_type {opt.interpolation_type}: video_double_latent_interpolation | video_latent_interpolation")
    if not torch.cuda.is_available():
        raise SystemExit("render_video_interpolation.py renders on the GPU (fenerf_amd has no CPU path)")
    device = torch.device('cuda')
    curriculum = resolve_curriculum(opt.curriculum)
    options = callers.video_kwargs(curriculum, opt.image_size, opt.ray_step_multiplier, opt.psi, opt.lock_view_dependence,
                                   opt.num_frames, opt.fov, opt.fill_color)
    os.makedirs(opt.output_dir, exist_ok=True)
    generator = callers.load_generator(opt.path, device, use_ema=not opt.no_ema, reset_render_options=False)
    if opt.interpolation_type == 'video_latent_interpolation':
        return run_single_latent(opt, generator, options, device)
    generator.output_dim = options['output_dim']          # :316-317
    generator.channel_dim = options['output_dim'] - 1
    trajectory = callers.camera_trajectory(opt.trajectory, options['num_frames'], options['fov'])
    for seed in opt.seeds:
        output_dir = os.path.join(opt.output_dir, f'interpolation_{opt.latent_type}_{seed}')
        frame_dir = os.path.join(output_dir, "images", f"{opt.latent_type}_{opt.trajectory}")
        os.makedirs(frame_dir, exist_ok=True)
        out = callers.render_double_latent_video(generator, int(seed), dict(options, max_batch_size=opt.max_batch_size, depth_map=opt.depth_map),
                                                 trajectory, latent_type=opt.latent_type, psi=opt.psi, device=device)
        F = out["images"].shape[0]
        depth_colors = []
        for j in range(F):
            imageio_lite.save_image(out["labels"][j:j + 1], os.path.join(frame_dir, f"label_{j}.png"), nrow=1, normalize=True)
            imageio_lite.save_image(out["images"][j:j + 1], os.path.join(frame_dir, f"img_{j}.png"), nrow=1, normalize=True)
            imageio_lite.save_image(out["acc"][j:j + 1], os.path.join(frame_dir, f"acc_{j}.png"), nrow=1, normalize=True)
            imageio_lite.save_image(out["depth"][j:j + 1], os.path.join(frame_dir, f"depth_{j}.png"), nrow=1, normalize=True)
            d = np.nan_to_num(out["depth"][j].numpy() / 2.0 * 255.0).clip(0, 255).astype(np.uint8)      # :421-427
            dc = imageio_lite.jet_colormap(d)
            depth_colors.append(dc)
            from PIL import Image
            Image.fromarray(dc).save(os.path.join(frame_dir, f"depth_color_{j + 1}.png"))
        if not opt.save_with_video:
            imageio_lite.save_image(out["images"], os.path.join(output_dir, "interp.png"), nrow=opt.num_frames, normalize=True)
            imageio_lite.save_image(out["labels"], os.path.join(output_dir, "interp_seg.png"), nrow=opt.num_frames, normalize=True)
            imageio_lite.save_image(out["acc"], os.path.join(output_dir, "interp_acc_map.png"), nrow=opt.num_frames, normalize=True)
            imageio_lite.save_image(out["depth"].unsqueeze(1), os.path.join(output_dir, "interp_depth_map.png"), nrow=opt.num_frames)
        else:
            writer = imageio_lite.AviWriter(os.path.join(output_dir, f'interp_{opt.latent_type}_{seed}.avi'), fps=25)
            for j in range(F):
                img = imageio_lite.to_uint8_hwc(imageio_lite.make_grid(out["images"][j:j + 1], normalize=True)).astype(np.float32)
                lab = imageio_lite.to_uint8_hwc(imageio_lite.make_grid(out["labels"][j:j + 1], normalize=True)).astype(np.float32)
                res = np.concatenate([img, lab, img * 0.5 + lab * 0.5, depth_colors[j].astype(np.float32)], axis=1)
                writer.write(res.astype(np.uint8))
            writer.release()
        print(f"seed {seed}: {F} frames -> {output_dir}")


if __name__ == '__main__':
    main()
