"""Curriculum config dicts + stage lookup helpers -- the kwargs bag every generator call receives.

reference: curriculums.py:4-39 (helpers), :42-98 (CelebA), :100-130 (CelebA_double_semantic),
:132-177 (CelebA_double_semantic_texture_embedding_256_dim_96).  int keys = stage start step, str keys = global;
extract_metadata(curriculum, step) merges them.  Values are configuration data, kept verbatim so the
reference's scripts can splat them unchanged.
"""
import math


def _stages(curriculum):
    """Stage start steps (the int keys), ascending."""
    return sorted(k for k in curriculum if isinstance(k, int))


def extract_metadata(curriculum, current_step):
    """Merged kwargs bag for `current_step`: the latest stage that has started, overlaid with the global (str) keys."""
    started = [k for k in _stages(curriculum) if k <= current_step]
    md = dict(curriculum[started[-1]]) if started else {}
    md.update({k: v for k, v in curriculum.items() if not isinstance(k, int)})
    return md


def next_upsample_step(curriculum, current_step):
    """First later stage whose img_size exceeds the current one (inf if none)."""
    size_now = extract_metadata(curriculum, current_step)["img_size"]
    later = [k for k in _stages(curriculum) if k > current_step and curriculum[k].get("img_size", 512) > size_now]
    return later[0] if later else float("Inf")


def last_upsample_step(curriculum, current_step):
    """Start step of the current resolution stage (0 if none)."""
    size_now = extract_metadata(curriculum, current_step)["img_size"]
    same = [k for k in _stages(curriculum) if k <= current_step and curriculum[k]["img_size"] == size_now]
    return same[0] if same else 0


def get_current_step(curriculum, epoch):
    return sum(1 for e in curriculum["update_epochs"] if epoch >= e)


_CAMERA = dict(fov=12, ray_start=0.88, ray_end=1.12, fade_steps=10000, h_stddev=0.3, v_stddev=0.155,
               h_mean=math.pi * 0.5, v_mean=math.pi * 0.5, sample_dist='gaussian')
_OPTIM = dict(topk_interval=2000, topk_v=0.6, betas=(0, 0.9), weight_decay=0, r1_lambda=0.2, grad_clip=10)
_DOUBLE = dict(unique_lr=True, latent_geo_dim=256, latent_app_dim=256, output_dim=22, generator='DoubleImplicitGenerator3d',
               discriminator_img='CCSDoubleEncoderDiscriminator', discriminator_seg='CCSDoubleEncoderDiscriminator',
               dataset='CelebAMaskHQ_wo_background_seg_18', clamp_mode='relu', z_dist='gaussian', hierarchical_sample=True,
               z_geo_lambda=0, z_app_lambda=0, pos_lambda=15, last_back=False, eval_last_back=False,
               d_seg_loss_lambda=0.1, g_seg_loss_lambda=0.1, softmax_label=False, target_size=128,
               fill_mode='seg_padding_background', dataset_path='data/celebahq_mask', background_mask=True)

# curriculums.py:42-80 -- single-latent pi-GAN baseline (SPATIALSIRENBASELINE + ImplicitGenerator3d)
CelebA = {
    0: {'batch_size': 24 * 2, 'num_steps': 12, 'img_size': 64, 'batch_split': 2, 'gen_lr': 6e-5, 'disc_lr': 2e-4},
    int(200e3): {},
    'dataset_path': '/media/data2/sunjx/FENeRF/data/celebahq/data512x512/*.jpg',
    **_CAMERA, **_OPTIM,
    'unique_lr': False, 'latent_dim': 512, 'output_dim': 4, 'model': 'SPATIALSIRENBASELINE',
    'generator': 'ImplicitGenerator3d', 'discriminator': 'CCSEncoderDiscriminator', 'dataset': 'CelebA',
    'clamp_mode': 'relu', 'z_dist': 'gaussian', 'hierarchical_sample': True, 'z_lambda': 0, 'pos_lambda': 15,
    'last_back': False, 'eval_last_back': True, 'fill_mode': 'eval_white_back', 'target_size': 128,
}

# curriculums.py:83-130 -- FENeRF w/o latent grid
CelebA_double_semantic = {
    0: {'batch_size': 24, 'num_steps': 12, 'img_size': 32, 'batch_split': 6, 'gen_lr': 5e-5, 'disc_img_lr': 2e-4, 'disc_seg_lr': 1e-4},
    int(10e3): {'batch_size': 12, 'num_steps': 12, 'img_size': 64, 'batch_split': 2, 'gen_lr': 2e-5, 'disc_img_lr': 1e-4, 'disc_seg_lr': 5e-5},
    int(50e3): {'batch_size': 4, 'num_steps': 24, 'img_size': 128, 'batch_split': 4, 'gen_lr': 5e-6, 'disc_img_lr': 5e-5, 'disc_seg_lr': 2e-5},
    int(500e3): {},
    **_CAMERA, **_OPTIM, **_DOUBLE,
    'model': 'SIRENBASELINESEMANTICDISENTANGLE',
}

# curriculums.py:132-177 -- FENeRF w/ latent grid: the configuration BASELINE.json's metric is quoted on
CelebA_double_semantic_texture_embedding_256_dim_96 = {
    0: {'batch_size': 24, 'num_steps': 24, 'img_size': 32, 'batch_split': 4, 'gen_lr': 6e-5, 'disc_img_lr': 2e-4, 'disc_seg_lr': 2e-4},
    int(20e3): {'batch_size': 48, 'num_steps': 24, 'img_size': 64, 'batch_split': 4, 'gen_lr': 6e-5, 'disc_img_lr': 2e-4, 'disc_seg_lr': 2e-4},
    int(50e3): {'batch_size': 24, 'num_steps': 24, 'img_size': 128, 'batch_split': 4, 'gen_lr': 2e-5, 'disc_img_lr': 5e-5, 'disc_seg_lr': 2e-5},
    int(500e3): {},
    **_CAMERA, **_OPTIM, **_DOUBLE,
    'model': 'TextureEmbeddingPiGAN256SEMANTICDISENTANGLE_DIM_96',
}
