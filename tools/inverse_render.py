#!/usr/bin/env python
"""GAN inversion in FiLM space on the MI355X-native path -- the command-line surface of the reference's
inverse_render_double_semantic.py (:132-169, :573-585), same positional arguments and options:

    python tools/inverse_render.py <name> <path/to/generator.pth> --image_path face.jpg --seg_path face.png --save_dir out \\
           --lambda_seg 1 --lambda_img 1 --latent_normalize --iteration 1000 [--image_size 128] [--recon --trajectory front --num_frames 100]

--image_path / --seg_path are one photo + its label map ('L' image, values 0 .. 18), or two directories (sorted *.jpg / *.png, paired in
order).  Per pair: targets built as the script's torchvision transforms build them (callers.inversion_targets), then `--iteration` Adam
steps on the four FiLM offset tensors (callers.inverse_render: every iteration is one native differentiable render; only the FiLM
gradients are computed), previews `<i>_<angle>_img.jpg` / `_seg.jpg` at eleven yaws every 200 iterations and the frontal mIoU against the
19-class map every 20 (both through staged_forward_with_frequencies at 256^2 x 48+48, as the script), and finally
`<save_dir>/freq_phase_offset_<name>.pth` (the eight tensors, the reference's keys) + `mious.npy`.  An existing --checkpoint_path is
reused unless --load_checkpoint.  --recon renders the inverted identity along --trajectory into
`reconstructed_debug_<trajectory>_<fill_color>.avi` ([image | labels | blend], 25 fps; the reference writes mp4v through cv2, which is not
a dependency here).

Differences from the script, on purpose: (1) LPIPS is not shipped (no network weights offline): --lambda_percept > 0 needs
`--percept module:callable` naming a factory of an nn.Module with LPIPS's call signature; (2) the script adds `lambda_norm * norm_loss`
on every iteration but defines norm_loss only under --latent_normalize (a NameError otherwise): here the term is simply absent without
--latent_normalize; (3) accepted and unused there as here: --seeds, --inverse_type, --img_loss, --seg_loss, --latent_type, --psi,
--depth_map, --save_with_video; tensorboard logging is dropped; (4) --preview_size / --preview_steps (default 256 / 48: the script's
fixed values) shrink the preview renders for smoke tests.
"""
import argparse
import glob
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PREVIEW_ANGLES = (-0.5, -0.4, -0.3, -0.2, -0.1, 0, 0.1, 0.2, 0.3, 0.4, 0.5)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('name', type=str, default='debug')
    parser.add_argument('generator_path', type=str)
    parser.add_argument('--image_path', type=str)
    parser.add_argument('--seg_path', type=str)
    parser.add_argument('--save_dir', type=str)
    parser.add_argument('--load_checkpoint', type=bool, default=False)
    parser.add_argument('--seeds', nargs='+', default=[0])
    parser.add_argument("--init_seed", default=0, type=int)
    parser.add_argument('--image_size', type=int, default=256)
    parser.add_argument('--fov', default=12, type=int)
    parser.add_argument('--num_frames', type=int, default=100)
    parser.add_argument('--max_batch_size', type=int, default=2400000)
    parser.add_argument("--lock_view_dependence", default=False)
    parser.add_argument("--iteration", type=int, default=1000)
    parser.add_argument("--background_mask", action='store_true')
    parser.add_argument("--white_background_mask", action='store_true')
    parser.add_argument("--inverse_type", default='semantic', help='inverse rendering signal, i.e. semantic map or image or both')
    parser.add_argument("--img_loss", default='mse')
    parser.add_argument("--seg_loss", type=str, default='mse')
    parser.add_argument("--lambda_img", type=float, default=0.)
    parser.add_argument("--lambda_seg", type=float, default=0.)
    parser.add_argument("--lambda_percept", type=float, default=0.)
    parser.add_argument("--lambda_norm", type=float, default=1.)
    parser.add_argument("--latent_normalize", action="store_true")
    parser.add_argument("--latent_type", default='app')
    parser.add_argument("--psi", type=float, default=0)
    parser.add_argument("--init_psi", type=float, default=0)
    parser.add_argument("--trajectory", default='front')
    parser.add_argument('--depth_map', action='store_true')
    parser.add_argument("--save_with_video", action='store_true')
    parser.add_argument("--recon", action="store_true")
    parser.add_argument("--fill_color", type=str, default='black', help='the rendering background color, only for segmantic 18 type models')
    parser.add_argument("--no_center_crop", action='store_true')
    parser.add_argument("--checkpoint_path", default='', type=str)
    # not in the reference:
    parser.add_argument('--no_ema', action='store_true', help='use the raw generator weights (the reference always loads <prefix>ema.pth)')
    parser.add_argument('--percept', type=str, default=None, help='module:callable returning a perceptual-loss nn.Module (LPIPS is not shipped)')
    parser.add_argument('--preview_size', type=int, default=256, help='img_size of the preview / mIoU / recon renders (the reference: 256)')
    parser.add_argument('--preview_steps', type=int, default=48, help='num_steps of the preview / mIoU / recon renders (the reference: 48)')
    parser.add_argument('--sparse_backward', choices=['off', 'on', 'auto'], default='off',
                        help="exact-sparsity backward of every iteration's render (DESIGN.md 4.5): same gradients to the order of the sums, "
                             "about half the time per iteration on a mostly empty density field; 'auto' picks per iteration")
    return parser


def run_inverse_render(opt, generator, img_path, seg_path, percept=None):
    """run_inverse_render of the script (:268-460) -> the checkpoint path"""
    import numpy as np
    import torch
    from PIL import Image
    from fenerf_amd import callers, imageio_lite
    device = generator.device
    torch.manual_seed(opt.init_seed)
    os.makedirs(opt.save_dir, exist_ok=True)
    checkpoint_path = opt.checkpoint_path
    if os.path.exists(checkpoint_path) and not opt.load_checkpoint:
        return checkpoint_path
    gt_image, gt_seg_18, gt_seg_19 = callers.inversion_targets(Image.open(img_path), Image.open(seg_path), image_size=opt.image_size,
                                                               no_center_crop=opt.no_center_crop, background_mask=opt.background_mask,
                                                               white_background_mask=opt.white_background_mask)
    gt_image, gt_seg_18 = gt_image.to(device), gt_seg_18.to(device)
    options = callers.inversion_options(opt.image_size, opt.fov, device)
    render_options = callers.inversion_render_options(opt.fov, opt.fill_color, opt.preview_size, opt.preview_steps)
    mious = []
    if tuple(gt_seg_19.shape[-2:]) != (opt.preview_size, opt.preview_size):      # smoke-test previews: the mIoU target at the preview's size
        gt_seg_19 = torch.nn.functional.interpolate(gt_seg_19, size=(opt.preview_size, opt.preview_size), mode="nearest")

    def on_step(i, loss, meta):
        if i % 200 == 0:
            for angle, img in callers.render_inversion_views(generator, meta, render_options, PREVIEW_ANGLES, opt.max_batch_size, opt.lock_view_dependence):
                imageio_lite.save_image(img[:, -3:].cpu(), os.path.join(opt.save_dir, f"{i}_{angle}_img.jpg"), normalize=True)
                imageio_lite.save_image(callers.mask2color(img[:, :-3]).cpu(), os.path.join(opt.save_dir, f"{i}_{angle}_seg.jpg"), normalize=True)
        if i % 20 == 0:
            (_, img), = callers.render_inversion_views(generator, meta, render_options, (0,), opt.max_batch_size, opt.lock_view_dependence)
            gen_masks = callers.mask2labels(torch.argmax(img[:, :-3], dim=1).float()[0].cpu().numpy(), 19)
            mious.append(callers.mIOU(torch.Tensor(gen_masks[None]), gt_seg_19).item())

    z_dim = callers._latent_dims(generator)[0]
    res = callers.inverse_render(generator, gt_image, gt_seg_18, options, n_iterations=opt.iteration, init_psi=opt.init_psi,
                                 lambda_seg=opt.lambda_seg, lambda_img=opt.lambda_img, lambda_percept=opt.lambda_percept,
                                 lambda_norm=opt.lambda_norm if opt.latent_normalize else 0.0, percept=percept, z_dim=z_dim, on_step=on_step)
    meta = {k: res[k] for k in ('w_geo_frequencies', 'w_geo_phase_shifts', 'w_geo_frequency_offsets', 'w_geo_phase_shift_offsets',
                                'w_app_frequencies', 'w_app_phase_shifts', 'w_app_frequency_offsets', 'w_app_phase_shift_offsets')}
    checkpoint_path = os.path.join(opt.save_dir, f'freq_phase_offset_{opt.name}.pth')
    torch.save(meta, checkpoint_path)
    np.save(os.path.join(opt.save_dir, 'mious.npy'), mious)
    print(f"{os.path.basename(img_path)}: loss {res['losses'][0]:.5f} -> {res['losses'][-1]:.5f} in {opt.iteration} iterations"
          + (f", frontal mIoU {mious[0]:.3f} -> {mious[-1]:.3f}" if mious else "") + f" -> {checkpoint_path}")
    return checkpoint_path


def run_render_recon_video(opt, generator, checkpoint_path):
    """run_render_recon_video of the script (:463-501)"""
    import torch
    from fenerf_amd import callers, imageio_lite
    meta = torch.load(checkpoint_path, map_location=generator.device, weights_only=False)
    render_options = callers.inversion_render_options(opt.fov, opt.fill_color, opt.preview_size, opt.preview_steps)
    frames = callers.render_inversion_recon(generator, meta, render_options, callers.inversion_trajectory(opt.trajectory, opt.num_frames, opt.fov),
                                            opt.max_batch_size, opt.lock_view_dependence)
    out = os.path.join(opt.save_dir, f'reconstructed_debug_{opt.trajectory}_{opt.fill_color}.avi')
    writer = imageio_lite.AviWriter(out, fps=25)
    for f in frames:
        writer.write(f)
    writer.release()
    print(f"{len(frames)} frames -> {out}")
    return meta


def main(argv=None):
    from fenerf_amd import host
    host.respect_cpu_quota()          # torch's CPU thread pool no larger than the cores this process is granted (fenerf_amd/host.py)
    opt = build_parser().parse_args(argv)
    import torch
    from fenerf_amd import callers
    if not torch.cuda.is_available():
        raise SystemExit("inverse_render.py optimises on the GPU (fenerf_amd has no CPU path)")
    if not opt.image_path or not opt.seg_path or not opt.save_dir:
        raise SystemExit("--image_path, --seg_path and --save_dir are required")
    percept = None
    if opt.lambda_percept:
        if not opt.percept:
            raise SystemExit("--lambda_percept > 0 needs --percept module:callable (LPIPS and its weights are not shipped)")
        mod, fn = opt.percept.split(":")
        percept = getattr(importlib.import_module(mod), fn)().to('cuda')
    generator = callers.load_generator(opt.generator_path, torch.device('cuda'), use_ema=not opt.no_ema, reset_render_options=False)
    generator.siren.sparse_backward = {'off': False, 'on': True, 'auto': 'auto'}[opt.sparse_backward]
    generator.softmax_label = False                                     # (:174)
    if os.path.isdir(opt.image_path) and os.path.isdir(opt.seg_path):
        pairs = list(zip(sorted(glob.glob(opt.image_path + '/*.jpg')), sorted(glob.glob(opt.seg_path + '/*.png'))))
    elif os.path.isfile(opt.image_path) and os.path.isfile(opt.seg_path):
        pairs = [(opt.image_path, opt.seg_path)]
    else:
        raise SystemExit("--image_path / --seg_path: two files or two directories")
    for img_path, seg_path in pairs:
        checkpoint_path = run_inverse_render(opt, generator, img_path, seg_path, percept)
        if opt.recon:
            run_render_recon_video(opt, generator, checkpoint_path)


if __name__ == "__main__":
    main()
