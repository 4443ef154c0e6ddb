/*
 * fenerf.h -- C-ABI of the MI355X-native volumetric rendering core (libfenerf_hip.so).
 *
 * The reference (MrTornado24/FENeRF) has NO FFI on this path: the boundary today is plain Python
 * method calls on torch tensors.  Each entry point below names the reference code it replaces
 * (file:line under /root/reference).  Conventions:
 *   - plain C, int status return: 0 = ok, <0 = FENERF_E_* ; fenerf_last_error() gives a message
 *     (thread-local).  No exceptions cross the boundary, no torch types in any signature.
 *   - every pointer marked [dev] is a DEVICE pointer owned by the caller (e.g. a torch allocation),
 *     contiguous fp32 unless noted; [host] pointers are host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on it,
 *     nothing synchronises unless stated.
 *   - all randomness (stratified jitter, importance-sampling u, sigma noise, camera pose) is drawn by the
 *     caller and passed in: the library is deterministic.
 *   - a FenerfModel is immutable after create/update and may be used from several threads/streams.
 */
#ifndef FENERF_H_
#define FENERF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FENERF_ABI_VERSION 2   /* 2: FenerfModelDesc.wgrad_bf16_min_points (round 3) */

enum {
  FENERF_OK = 0,
  FENERF_E_INVALID = -1,     /* bad argument / unsupported shape */
  FENERF_E_HIP = -2,         /* a HIP runtime call failed (message has the hipError string) */
  FENERF_E_NOMEM = -3,
  FENERF_E_UNSUPPORTED = -4, /* model variant not built (hidden_dim not in {32,64,96,128,192,256}, ...).  Any other width up to 256 (the
                              * reference constructs any, siren/siren.py:1451) runs EXACTLY at the next instantiated one with zero padding:
                              * zero rows / trailing hidden columns of every weight matrix, zero biases and zero FiLM phase shifts make a
                              * padded feature sin(f' 0 + 0) = 0; the host packs that way (fenerf_amd/native.py padded_hidden_dim pads on
                              * the way in and slices gradients on the way out: pixels bit-identical to the padded network's) */
  FENERF_E_CLAMP_MODE = -5   /* reference raises TypeError("Need to choose clamp mode"), volumetric_rendering.py:34 */
};

/* arithmetic of the dense layers inside the SIREN kernel */
enum { FENERF_PREC_F32 = 0, FENERF_PREC_F16X3 = 1 };

/* clamp_mode of fancy_integration (volumetric_rendering.py:29-34) */
enum { FENERF_CLAMP_RELU = 1, FENERF_CLAMP_SOFTPLUS = 2 };

/* fill_mode of fancy_integration (volumetric_rendering.py:52-104) */
enum {
  FENERF_FILL_NONE = 0,
  FENERF_FILL_WEIGHT = 1,                      /* 'weight'                       -> third = weights_sum */
  FENERF_FILL_SEG_PADDING_BACKGROUND = 2,      /* 'seg_padding_background'       -> third = weights     */
  FENERF_FILL_EVAL_SEG_PADDING_BACKGROUND = 3, /* 'eval_seg_padding_background'  -> third = weights_sum */
  FENERF_FILL_EVAL_WHITE_BACK = 4              /* 'eval_white_back' (3-channel models)                 */
};

/* Radiance-field description: the per-point network of siren/siren.py.  All weight pointers are
 * [host], fp32, in torch nn.Linear layout weight[out][in], bias[out].
 *   TextureEmbeddingPiGAN{128,256}SEMANTICDISENTANGLE(_DIM_96)  siren.py:1451-1546 : grid_ch=32, n_color=3, n_label_layers=3
 *   SIRENBASELINESEMANTICDISENTANGLE                             siren.py:1163-1229 : grid_ch=0,  n_color=3, n_label_layers=2
 *   SPATIALSIRENBASELINE                                         siren.py:189-244   : grid_ch=0,  n_color=1, n_label_layers=0, output_dim=4
 */
#define FENERF_MAX_GEO 8
#define FENERF_MAX_COLOR 4
#define FENERF_MAX_LABEL_LAYERS 3
/* Samples per ray one wavefront composites (lane = sample, up to 16 samples per lane): 512 coarse + 512 fine of a hierarchical render, or
 * 1024 samples of a single pass.  The reference has no limit (its tensors are [B, R, M, C]); its curricula use 12 ... 48 + 48
 * (curriculums.py), the inversion / video scripts up to 48 + 48.  More than this returns FENERF_E_INVALID. */
#define FENERF_MAX_RAY_SAMPLES 1024
/* Format of the tape fenerf_siren_forward_save_fmt writes and the backward entry points read (FENERF_PREC_F16X3 models; exact-fp32
 * models keep FENERF_TAPE_F32):  F32 = the pre-FiLM accumulators, 4 bytes per (point, FiLM-layer feature) -- everything the backward
 * may want; U16 = frac(theta) as 16-bit fixed point, 2 bytes -- all that sin / cos need (siren.py:113-123), +-4.8e-5 rad per recomputed
 * activation, gradients within ~1.2e-4 (max-norm relative) of fp64 autograd instead of ~4e-5: a tier between the fp32 class and the
 * AMP class.  Not for the FiLM-only backward (fenerf_siren_backward_film: the frequency gradient needs the accumulator). */
#define FENERF_TAPE_F32 0
#define FENERF_TAPE_U16 1
/* FENERF_TAPE_F32_W: the fp32 tape of FENERF_TAPE_F32, bit for bit -- what changes is who forms the FiLM FREQUENCY gradient sum_p d theta
 * (W x + b): not the chain kernel from the tape's accumulators (a multiply and a 16-lane reduction per row tile: a fifth of its VALU
 * instructions) but the weight-gradient stage from its partial sums and the weights, as for U16 (fenerf_siren_param_grads_fmt takes
 * `weights`).  Same fp32 class (the sums are bf16x3 products like the weight gradients themselves).  Forward-save: identical to F32.
 * Not for FiLM-only backward passes. */
#define FENERF_TAPE_F32_W 2

typedef struct FenerfModelDesc {
  int32_t abi_version;      /* FENERF_ABI_VERSION */
  int32_t hidden_dim;       /* H in {32,64,96,128,192,256}: multiples of 32 the kernel templates are instantiated for (96 / 192: round 5) */
  int32_t n_geo;            /* FiLM layers of the density trunk (8) */
  int32_t n_color;          /* FiLM layers of the colour branch (3, or 1) */
  int32_t n_label_layers;   /* plain Linear layers of the semantic head (3, 2 or 0); NO activation between them */
  int32_t output_dim;       /* 22 = 18 labels + rgb + sigma ; 4 = rgb + sigma */
  int32_t grid_ch;          /* 32 or 0 */
  int32_t grid_d, grid_h, grid_w; /* spatial_embeddings is [1, grid_ch, D, H, W] (NCDHW, as torch stores it) */
  float box_scale;          /* UniformBoxWarp scale 2/0.24 (siren.py:181-187) */
  const float* geo_w[FENERF_MAX_GEO];   const float* geo_b[FENERF_MAX_GEO];     /* network[i].layer       */
  const float* color_w[FENERF_MAX_COLOR]; const float* color_b[FENERF_MAX_COLOR]; /* color_layer_sine[i].layer ; [0] has in = 3 + grid_ch + H, columns [dir | grid feats | x] */
  const float* label_w[FENERF_MAX_LABEL_LAYERS]; const float* label_b[FENERF_MAX_LABEL_LAYERS]; /* label_layer_linear[i] */
  const float* sigma_w; const float* sigma_b;   /* final_layer         [1][H]  */
  const float* rgb_w;   const float* rgb_b;     /* color_layer_linear[0] [3][H]  */
  const float* grid;        /* [host] spatial_embeddings or NULL */
  int32_t precision;        /* FENERF_PREC_F32 (exact fp32 MFMA) or FENERF_PREC_F16X3 (error-compensated fp16 MFMA,
                               3 MFMAs per product, fp32-class accuracy, ~2^-22 relative per product) */
  int32_t differentiable;   /* != 0: also keep the backward-chain weight stream resident (fenerf_siren_backward); either
                               precision: FENERF_PREC_F32 -> exact fp32 MFMA chain, FENERF_PREC_F16X3 -> bf16x3 chain */
  int32_t wgrad_bf16_min_points; /* 0 (default): fp32-class weight gradients everywhere.  > 0 (FENERF_PREC_F16X3 models only):
                               AMP-CLASS weight gradients -- in backward chunks of at least this many points fenerf_siren_backward writes
                               d theta and the layer inputs as bf16 and fenerf_siren_param_grads multiplies them with ONE bf16 MFMA per
                               product, fp32 accumulate (half the dump bytes, no tape re-read).  Rounding is unbiased; the error of a
                               weight gradient is ~1e-3 of its per-point noise floor sqrt(sum_p (dtheta x)^2), whatever the point
                               count -- the class of the reference's own autocast training (train_double_latent_semantic.py:402-446),
                               not the fp32 class of the default.  FiLM / bias / grid / head gradients and dz are unaffected. */
} FenerfModelDesc;

typedef struct FenerfModel FenerfModel;

/* CustomMappingNetwork (siren/siren.py:82-102): z -> [Linear, LeakyReLU(0.2)] x (n_blocks + 1) -> Linear; the two halves of the output are
 * the raw FiLM frequencies / phase shifts of an image.  All pointers [dev], fp32, nn.Linear layout (weight[out][in], bias[out]): the
 * mapping networks are trained parameters that live on the device.  Layer 0 is [hidden][z_dim], layers 1 .. n_layers - 2 [hidden][hidden],
 * the last [out_dim][hidden]. */
#define FENERF_MAP_MAX_LAYERS 8
typedef struct FenerfMappingNet {
  int32_t n_layers;     /* Linear layers = n_blocks + 2 (5 for the generators' networks, 3 for SPATIALSIRENGRID's) */
  int32_t z_dim, hidden, out_dim;
  const float* W[FENERF_MAP_MAX_LAYERS];
  const float* b[FENERF_MAP_MAX_LAYERS];
} FenerfMappingNet;
/* replaces: mapping_network(z) for small batches (generators.py:458-459 and every other `*.mapping_network(z)` call; ~18 ATen launches for
 * the two networks of a DoubleImplicitGenerator3d) -- ONE launch.  z [B][z_dim] -> out [B][out_dim]; acts [n_layers - 1][B][hidden] receives
 * the post-activation vectors fenerf_mapping_backward needs. */
int fenerf_mapping_forward(const FenerfMappingNet* net, int B, const float* z, float* acts, float* out, void* stream);
/* replaces: autograd through the same (AddmmBackward / LeakyReluBackward nodes, ~30 launches per network): gradients wrt every weight and
 * bias (dW[l] / db[l]: [dev] buffers of the parameters' shapes, overwritten) from d_out [B][out_dim]; three launches.  z itself gets no
 * gradient (the reference samples its latents; inversion optimises FiLM offsets).  workspace: fenerf_mapping_workspace_floats floats. */
size_t fenerf_mapping_workspace_floats(const FenerfMappingNet* net, int B);
int fenerf_mapping_backward(const FenerfMappingNet* net, int B, const float* z, const float* acts, const float* d_out,
                            float* const* dW, float* const* db, float* workspace, void* stream);

/* replaces: autograd through label_layer_linear (siren.py:1490-1494: 2-3 nn.Linear with NO activation between them, the layers the render
 * kernels evaluate as one folded affine map) -- the gradient of every layer's weight and bias from the gradient of the fold, i.e. from rows
 * [0, n_lab) of FenerfSirenGrads.head_w / head_b (g_head_w [n_lab][H], g_head_b [n_lab]).  W / b: the n_layers layers in application order,
 * [dev] fp32 nn.Linear layout (layers 0 .. n-2 [H][H], the last [n_lab][H]); dW / db: buffers of the same shapes, overwritten.  n_lab <= 32.
 * One launch for two layers, two for three (round 5; rounds 2-4: 11 rocBLAS / ATen launches at the end of every generator step).
 * workspace: fenerf_label_head_workspace_floats(H) floats (three layers; may be NULL otherwise). */
size_t fenerf_label_head_workspace_floats(int H);
int fenerf_label_head_backward(int n_layers, int H, int n_lab, const float* const* W, const float* const* b, const float* g_head_w,
                               const float* g_head_b, float* const* dW, float* const* db, float* workspace, void* stream);

/* per-call compositing options = the kwargs fancy_integration reads (volumetric_rendering.py:18) */
typedef struct FenerfCompositeOpts {
  int32_t clamp_mode;   /* FENERF_CLAMP_* ; 0 -> FENERF_E_CLAMP_MODE */
  float noise_std;      /* multiplies the caller-drawn N(0,1) noise (`nerf_noise`) */
  int32_t last_back, white_back, black_back;
  int32_t fill_mode;    /* FENERF_FILL_* */
  float fill_value;     /* colour written to filled pixels: black 0, white 1, grey 0.5, light_grey 0.81 */
  int32_t fill_enabled; /* 0 when fill_color is none of the four names (reference then pads channel 0 but does not fill) */
} FenerfCompositeOpts;

const char* fenerf_last_error(void);
int fenerf_abi_version(void);

/* Layout of the structs of this header AS THE LIBRARY WAS COMPILED, so that a binding written in another language (the ctypes mirrors of
 * INTEGRATION.md B and fenerf_amd/_lib.py) can be checked against the library it loads instead of against a copy of this file:
 * fenerf_struct_size("FenerfModelDesc") = sizeof, fenerf_struct_field_offset("FenerfModelDesc", "precision") = offsetof; structs:
 * FenerfModelDesc, FenerfCompositeOpts, FenerfRepackMaps, FenerfLocalMapDesc, FenerfSirenGrads, FenerfMappingNet.  Unknown names return -1.
 * fenerf_struct_field_name(s, i) enumerates the fields in declaration order (NULL behind the last).  (No reference analogue: the
 * reference has no FFI.)  tests/test_host_cpu.py executes INTEGRATION.md's snippet against these. */
/* Scheduling hint for the CALLING THREAD's subsequent launches (no reference analogue): size persistent launches (the backward chain)
 * and the weight-gradient grids for at most `cus` compute units instead of all of them; 0 = the whole device (default).  Returns the
 * previous value.  The generator step uses it to run the weight gradients of backward chunk i on a second stream BESIDE the chain of
 * chunk i + 1: the chain scales with the CUs it gets (2.45 -> 2.74 -> 3.40 ms for 393,216 points on 256 / 192 / 128 CUs), the
 * weight-gradient kernels do not (1.72 / 1.91 / 1.68 ms: they are HBM-bound), so the two share the chip (profiles/r04_gstep_overlap.md).
 * Workspace sizes (fenerf_siren_grad_workspace_bytes) depend on it: query them under the same setting as the launch. */
int fenerf_set_cu_budget(int cus);
/* How the CALLING THREAD's fenerf_render_forward calls run a hierarchical render of a FENERF_PREC_F16X3 model (no reference analogue; the
 * results are the same bit for bit either way -- both routes run the same per-tile and per-ray code):
 *   FENERF_FUSION_OFF    four launches: coarse SIREN, weights + resampling, fine SIREN, merge + composite;
 *   FENERF_FUSION_FORCE  ONE launch whenever the shape allows it (2 N <= 128, rays per image a multiple of the ray group): per group of
 *        rays whose samples fill whole 128-point tile groups (16 rays at N = 24) a workgroup evaluates the coarse samples, composites and
 *        resamples them, evaluates the fine samples and composites the pixels -- generators.py:479-519 without leaving the launch;
 *   FENERF_FUSION_AUTO (default)  the faster of the two as measured on MI355X: the one launch where the shape allows it, whole ray groups
 *        per workgroup do not lengthen the critical path AND the library was built with it enabled for AUTO -- which it is not: the
 *        one launch is 6 % slower (profiles/r04_render_one_launch.md: its ray phases run at 8 waves per CU with the matrix pipe idle),
 *        so AUTO currently means the four launches.
 * Returns the previous mode. */
#define FENERF_FUSION_AUTO 0
#define FENERF_FUSION_OFF 1
#define FENERF_FUSION_FORCE 2
int fenerf_set_render_fusion(int mode);

long fenerf_struct_size(const char* struct_name);
long fenerf_struct_field_offset(const char* struct_name, const char* field);
const char* fenerf_struct_field_name(const char* struct_name, int index);

/* Packs (host side, no GPU needed) the weights of `desc` into the kernel's streaming layout; used by
 * fenerf_model_create and exposed for layout tests.  *blob is malloc'd, free with fenerf_free_host. */
int fenerf_pack_weights_host(const FenerfModelDesc* desc, float** blob, size_t* n_floats,
                             float** consts, size_t* n_consts);
void fenerf_free_host(void* p);

/* replaces: constructing the siren nn.Module + .to(device)  (generators/generators.py:440) */
int fenerf_model_create(const FenerfModelDesc* desc, FenerfModel** out);
/* replaces: optimizer.step() mutating the nn.Module in place -- re-packs weights (and the grid if non-NULL) */
int fenerf_model_update(FenerfModel* m, const FenerfModelDesc* desc, void* stream);
void fenerf_model_destroy(FenerfModel* m);

/* Opt-in reduced-precision arithmetic for the NO-GRAD forward of a FENERF_PREC_F16X3 model (round 5; no reference analogue -- the
 * reference's own low-precision mode is autocast fp16, train_double_latent_semantic.py:279).  The default evaluates every fp32 product as
 * three fp16 MFMAs (wl xh + wh xl + wh xh: fp32-class results).  FENERF_FORWARD_F16X2 drops wl xh everywhere -- the weights enter as ONE
 * fp16 value (2^-12 relative rounding each, a fixed perturbation of the network), a third of the MFMAs and half of the weight traffic
 * through L2 -> LDS go away; FENERF_FORWARD_F16X3_COLOR_X2 keeps three terms through the geometry trunk and the label / sigma
 * head (sigma -- which decides resampling bins and the 0.9 fill threshold -- and the labels stay the default's bit for bit) and two in the
 * colour layers and the rgb head.  Both are measured beside the default in bench.py and
 * against the same tests (profiles/r05_*): they do NOT meet the default's asserted bounds, hence opt-in.  fenerf_siren_forward*,
 * fenerf_render_forward only; the differentiable path always runs the default.  Returns the previous mode or a negative error. */
#define FENERF_FORWARD_F16X3 0
#define FENERF_FORWARD_F16X2 1
#define FENERF_FORWARD_F16X3_COLOR_X2 2
int fenerf_model_set_forward_mode(FenerfModel* m, int mode);

/* Training keeps the weights on the GPU, so re-packing them through the host every optimizer step (fenerf_model_update)
 * costs more than the step itself.  For FENERF_PREC_F32 models the packed streams are pure permutations (plus zero
 * padding) of the parameters (FENERF_PREC_F16X3: of their scaled fp16 hi / lo halves): the caller builds them on the device (a gather with an index map obtained ONCE by packing
 * index-valued weights with fenerf_pack_weights_host / fenerf_pack_backward_host) and hands them over here with
 * device-to-device copies.  stream_dev / consts_dev as fenerf_pack_weights_host returns them, bwd_dev as
 * fenerf_pack_backward_host (NULL unless the model is differentiable), grid_dev = spatial_embeddings [1,32,D,H,W] or NULL. */
int fenerf_pack_backward_host(const FenerfModelDesc* desc, float** blob, size_t* n_floats);
/* FENERF_PREC_F16X3 streams are not permutations (per-row power-of-two scales, fp16 hi / lo splits), but their LAYOUT is:
 * called with index-valued weights (element = 1 + its flat index) this returns, for every fp16 half of the ring stream
 * (after the fp32 layer-0 block), that index | (is_lo << 30), 0 for padding.  The caller scales rows, splits and gathers
 * on the device (fenerf_amd/native.py::NativeModel.load_from_device) and hands the result to fenerf_model_load_packed. */
int fenerf_pack_index_map_f16(const FenerfModelDesc* desc, int32_t** map, size_t* n);
/* The backward stream of a FENERF_PREC_F16X3 model holds bf16 (hi, lo) pairs of the scaled transposed weights after its fp32
 * rgb-head block (fenerf_pack_backward_host packs it when desc->precision says so).  Called with index-valued weights
 * this returns that index for every bf16 half of the ring (0 = padding); hi / lo alternate per 512-half entry
 * (hi = RNE(w), lo = RNE(w - hi)). */
int fenerf_pack_backward_index_map_bf16(const FenerfModelDesc* desc, int32_t** map, size_t* n);
int fenerf_model_load_packed(FenerfModel* m, const float* stream_dev, size_t n_stream, const float* consts_dev, size_t n_consts,
                             const float* bwd_dev, size_t n_bwd, const float* grid_dev, void* stream);

/* The same re-pack done natively, straight into the model's resident buffers (four small kernels instead of a hundred
 * framework ops; replaces what optimizer.step() on a resident nn.Module costs the reference: nothing).  flat_dev = every
 * render parameter concatenated in the caller's canonical order behind a leading 0 (so index 0 = "zero"); the maps are
 * [dev] int32 arrays built ONCE from the fenerf_pack_*_host / fenerf_pack_*index_map* results on index-valued weights:
 *   stream_f32 / consts / bwd_f32 : fp32 elements -> flat index (whole streams for FENERF_PREC_F32; the fp32 layer-0 block,
 *                                    the fp32 consts and the fp32 rgb-head block for FENERF_PREC_F16X3)
 *   stream_h16 : fp16 halves of the forward ring, flat index | is_lo << 30        (fenerf_pack_index_map_f16)
 *   bwd_b16    : bf16 halves of the backward ring, hi / lo alternate per 512       (fenerf_pack_backward_index_map_bf16)
 *   row_off / row_len / row_film : the rows that get a power-of-two scale (FiLM layers >= 1 and the heads): flat offset,
 *                                    length, 1 for FiLM-layer rows;  scale_id[flat index] = 1 + row, 0 = unscaled
 *   consts_tail: the f16x3 result scales behind the fp32 consts: k > 0 -> 1 / (16 scale[k]), 0 -> 1/16, < 0 -> 1
 * Unused maps are NULL / 0.  grid_dev = spatial_embeddings [1,32,D,H,W] or NULL (unchanged). */
typedef struct FenerfRepackMaps {
  const int32_t* stream_f32; size_t n_stream_f32;
  const int32_t* stream_h16; size_t n_stream_h16;
  const int32_t* consts;     size_t n_consts;
  const int32_t* consts_tail; size_t n_tail;
  const int32_t* bwd_f32;    size_t n_bwd_f32;
  const int32_t* bwd_b16;    size_t n_bwd_b16;
  const int32_t* row_off; const int32_t* row_len; const int32_t* row_film; int32_t n_rows;
  const int32_t* scale_id;
} FenerfRepackMaps;
int fenerf_model_repack(FenerfModel* m, const float* flat_dev, size_t n_flat, const FenerfRepackMaps* maps, const float* grid_dev,
                        void* stream);
/* Copies the resident packed buffers out (device to device): tests compare re-packs bit for bit.  NULL = skip. */
int fenerf_model_export_packed(const FenerfModel* m, float* stream_dev, size_t n_stream, float* consts_dev, size_t n_consts,
                               float* bwd_dev, size_t n_bwd, void* stream);

/* Bytes of [dev] scratch the FiLM pre-pass needs for a batch of B images. */
size_t fenerf_film_workspace_bytes(const FenerfModel* m, int B);

/* Per-point modulation -- replaces: SPATIALSIRENGRID.forward_with_frequencies_phase_shifts (siren.py:464-477), whose
 * frequencies / phase shifts come from a mapping network evaluated on a local latent PER SAMPLE POINT (:440-462, FiLMLayer
 * takes them unbroadcast, :119-122).  Like fenerf_siren_forward, but freq_geo / phase_geo are [B, P, n_geo*H] and freq_app /
 * phase_app [B, P, n_color*H] (raw mapping outputs, '*15+30' inside).  The model must have FENERF_PREC_F32 (the FiLM blocks
 * are read per lane by the exact kernel).  film_ws: fenerf_film_workspace_bytes_pointwise(m, B, P) bytes of [dev] scratch. */
size_t fenerf_film_workspace_bytes_pointwise(const FenerfModel* m, int B, int64_t P);
int fenerf_siren_forward_pointwise(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                                   const float* freq_geo, const float* phase_geo, const float* freq_app,
                                   const float* phase_app, float* out, void* film_ws, void* stream);

/* SPATIALSIRENGRID (siren.py:413-518) as ONE launch: per-point mapping network + FiLM-SIREN.
 * replaces: self.mapping_network(sampled_latent) + forward_with_frequencies_phase_shifts with per-point parameters (siren.py:455-477).
 * FenerfLocalMapDesc = the weights of CustomMappingNetwork(32, 256, 2 L H, n_blocks = 1) (siren.py:440), [host], nn.Linear layout:
 * frequencies are rows [0, L H) of w2, phase shifts rows [L H, 2 L H) (CustomMappingNetwork.forward, siren.py:98-102), FiLM layer l =
 * rows l H .. (l + 1) H of either half, the colour layer last (siren.py:468-475).  `siren` must describe the rgb + sigma model without
 * feature grid (grid_ch 0, n_label_layers 0, output_dim 4); its precision / differentiable fields are ignored: the kernel is exact fp32
 * (v_mfma_f32_32x32x2_f32), forward only.  fenerf_siren_forward_local: points [P,3] (the caller's LOCAL cell coordinates, :458-461;
 * the kernel applies box_scale like :466), ray_dirs [P,3] or NULL (lock), latents [P,32] (sample_local_latents, :479-499) -> out [P,4]
 * = [rgb | sigma].  Per point 152 B in, 16 B out; frequencies / phase shifts never exist in memory. */
typedef struct FenerfLocalMapDesc {
  int32_t latent_dim;                 /* 32 */
  int32_t map_hidden;                 /* 256 */
  const float* w0; const float* b0;   /* mapping_network.network.0: [map_hidden][latent_dim], [map_hidden] */
  const float* w1; const float* b1;   /* mapping_network.network.2: [map_hidden][map_hidden] */
  const float* w2; const float* b2;   /* mapping_network.network.4: [2 L H][map_hidden], [2 L H] */
} FenerfLocalMapDesc;
typedef struct FenerfLocalModel FenerfLocalModel;
int fenerf_local_model_create(const FenerfModelDesc* siren, const FenerfLocalMapDesc* map, FenerfLocalModel** out);
/* the packed stream / constants fenerf_local_model_create uploads (host side, no GPU needed: layout tests); free with fenerf_free_host */
int fenerf_pack_local_host(const FenerfModelDesc* siren, const FenerfLocalMapDesc* map, float** blob, size_t* n_floats, float** consts,
                           size_t* n_consts);
void fenerf_local_model_destroy(FenerfLocalModel* m);
int fenerf_siren_forward_local(const FenerfLocalModel* m, int64_t total_points, const float* points, const float* ray_dirs,
                               const float* latents, float* out, void* stream);

/* replaces: <siren>.forward_with_frequencies_phase_shifts  (siren.py:1509-1530 / :1210-1229 / :227-244),
 *           incl. UniformBoxWarp (:181-187), sample_from_3dgrid (:314-330) and FiLMLayer (:113-123).
 * points [B,P,3], ray_dirs [B,P,3] (NULL = lock_view_dependence, i.e. (0,0,-1): generators.py:474-476),
 * freq/phase = RAW mapping-network outputs (the '*15+30' is applied inside): freq_geo/phase_geo [B, n_geo*H],
 * freq_app/phase_app [B, n_color*H].  out [B,P,output_dim] = [labels | rgb | sigma]. */
int fenerf_siren_forward(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                         const float* freq_geo, const float* phase_geo, const float* freq_app,
                         const float* phase_app, float* out, void* film_ws, void* stream);

/* Same network evaluated on points generated in-kernel from rays: point (b,r,k) = origins[b,r] + dirs[b,r]*z[b,r,k]
 * (generators.py:504 for the fine pass; the coarse pass of the reference transforms camera-space points
 * instead, volumetric_rendering.py:160 -- equal up to fp32 rounding).  lock_view: use (0,0,-1) as view dir. */
int fenerf_siren_forward_rays(const FenerfModel* m, int B, int R, int N, const float* origins, const float* dirs,
                              const float* z, int lock_view, const float* freq_geo, const float* phase_geo,
                              const float* freq_app, const float* phase_app, float* out, void* film_ws,
                              void* stream);

/* replaces: get_initial_rays_trig + perturb_points + the pose / cam2world / bmm part of transform_sampled_points
 * (volumetric_rendering.py:109-168, :220-248) for square images: u_jitter [B,R,N] ~ U[0,1) and the camera angles theta (yaw),
 * phi (pitch, before the [1e-5, pi-1e-5] clamp) [B] are the caller's draws; z_cam = -1/tan(fov/2) as the caller computes it.
 * -> origins [B,R,3], dirs [B,R,3] (world space), z [B,R,N] (jittered), pitch [B] (clamped phi), yaw [B]. */
int fenerf_ray_setup(int B, int img_size, int N, float z_cam, float ray_start, float ray_end, const float* u_jitter,
                     const float* theta, const float* phi, float* origins, float* dirs, float* z, float* pitch, float* yaw,
                     void* stream);

/* Measurement hook (bench.py roofline leg; replaces nothing): runs the FiLM pre-pass once, then `iters` back-to-back
 * launches of ONLY the SIREN kernel of fenerf_siren_forward_rays, bracketed by hipEvents recorded on `stream`;
 * synchronises and returns the average kernel duration in milliseconds. */
int fenerf_siren_time_rays(const FenerfModel* m, int B, int R, int N, const float* origins, const float* dirs,
                           const float* z, const float* freq_geo, const float* phase_geo, const float* freq_app,
                           const float* phase_app, float* out, void* film_ws, int iters, float* avg_ms, void* stream);

/* Measurement hooks (bench.py; replace nothing).
 * fenerf_siren_clock_probe: like fenerf_siren_time_rays, but every workgroup of every timed launch also stamps the shader-clock
 * counter (s_memtime) and the constant-rate wall clock (s_memrealtime) at its first and last instruction.  result[0] = average
 * kernel ms (hipEvents), [1] = shader cycles per launch (average over launches of the longest workgroup), [2] = effective shader
 * clock in GHz over those workgroups (cycles / wall-clock ticks x the device's wall-clock rate), [3] = that rate in kHz.
 * fenerf_siren_executed_flop_per_point: FLOPs the model's SIREN kernel ISSUES on the matrix pipe per sample point (static count of
 * its MFMA instructions: 3 fp16 MFMAs per product at FENERF_PREC_F16X3, the label head folded to one 18-row map) -- next to the
 * ALGORITHMIC 2 x MAC count of the reference network that roofline figures are quoted on. */
int fenerf_siren_clock_probe(const FenerfModel* m, int B, int R, int N, const float* origins, const float* dirs, const float* z,
                             const float* freq_geo, const float* phase_geo, const float* freq_app, const float* phase_app,
                             float* out, void* film_ws, int iters, double* result4, void* stream);
double fenerf_siren_executed_flop_per_point(const FenerfModel* m);
/* Per-phase device times of everything the library launches (bench.py's generator-step breakdown).  fenerf_phase_timing(1)
 * makes every launch group record a hipEvent pair on its stream (off by default: no events, no cost); fenerf_phase_times
 * synchronises those events, ADDS the elapsed milliseconds / launch-group counts per phase into ms[] / calls[] (n >=
 * FENERF_N_PHASES entries, caller-zeroed) and forgets them.  fenerf_phase_name(i) names phase i ("forward_save", "chain", ...). */
#define FENERF_N_PHASES 17
int fenerf_phase_timing(int enable);
int fenerf_phase_times(double* ms, int* calls, int n);
const char* fenerf_phase_name(int phase);

/* replaces: fancy_integration (volumetric_rendering.py:18-106).
 * rgb_sigma [BR, M, C] (M <= 512), z [BR, M], noise [BR, M] N(0,1) draws or NULL.
 * out_rgb [BR, C-1] (or [BR, C] for the two seg-padding fill modes), out_depth [BR],
 * out_weights [BR, M] or NULL, out_wsum [BR] or NULL. */
int fenerf_composite(int64_t BR, int M, int C, const float* rgb_sigma, const float* z, const float* noise,
                     const FenerfCompositeOpts* opts, float* out_rgb, float* out_depth, float* out_weights,
                     float* out_wsum, void* stream);

/* replaces: the importance-resampling block generators.py:486-499 + sample_pdf (volumetric_rendering.py:259-300):
 * weights+1e-5, z_mid, sample_pdf(z_mid, w[:,1:-1], N, det=False) with the caller's u ~ U[0,1) [BR, N].
 * z_coarse [BR,N], coarse_weights [BR,N] (fancy_integration's 3rd output) -> z_fine [BR,N] (unsorted). */
int fenerf_resample(int64_t BR, int N, const float* z_coarse, const float* coarse_weights, const float* u,
                    float* z_fine, void* stream);

/* replaces: sample_pdf itself (volumetric_rendering.py:259-300) in its reference shape: bins [BR,K+1], weights [BR,K]
 * (the function adds eps=1e-5), u [BR,n_importance] -> samples [BR,n_importance]. */
int fenerf_sample_pdf(int64_t BR, int K, int n_importance, const float* bins, const float* weights, const float* u,
                      float* samples, void* stream);

/* replaces: cat([fine, coarse]) -> torch.sort(z) -> gather -> fancy_integration (generators.py:508-519):
 * merges without materialising the sorted [BR,2N,C] tensor.  fine/coarse [BR,N,C] (2 N <= FENERF_MAX_RAY_SAMPLES), z_* [BR,N], noise [BR,2N] or NULL
 * (indexed by SORTED position, like the reference's noise tensor).  out_weights [BR,2N] in sorted order or NULL;
 * out_z_sorted [BR,2N] or NULL. */
int fenerf_merge_composite(int64_t BR, int N, int C, const float* fine, const float* coarse, const float* z_fine,
                           const float* z_coarse, const float* noise, const FenerfCompositeOpts* opts,
                           float* out_rgb, float* out_depth, float* out_weights, float* out_wsum,
                           float* out_z_sorted, void* stream);

/* ---- differentiable evaluation (generator step / inversion).  replaces what torch autograd records and replays for
 * <siren>.forward_with_frequencies_phase_shifts (siren.py:1509-1530) in train_double_latent_semantic.py (g_loss.backward())
 * and inverse_render_double_semantic.py.  The model must be created with differentiable != 0; P (points per image) must
 * be a multiple of 32 (pad with any valid point and give the pads a zero gradient).
 *
 * fenerf_siren_forward_save = fenerf_siren_forward that also keeps the pre-FiLM accumulators (W_l x_{l-1}, no bias) of
 *   every FiLM layer as a tape of fenerf_siren_tape_floats(m, B*P) floats (opaque: 32-point register dumps, L*H*B*P floats
 *   plus up to three tiles of slack the kernel may scribble on,
 *   fenerf_amd/csrc/fenerf_layout.h "Tape"; for FENERF_PREC_F16X3 models in the row-scaled units of that GEMM) and the
 *   sampled grid features tape_e [B*P][32] (NULL without a grid).
 * fenerf_siren_backward: d_out [B,P,output_dim] -> d_t = dL/dtheta per FiLM layer in the tape's layout, theta = f (W x + b) + p,
 *   followed by the per-tile FiLM sums (sum over the tile's points of dtheta and dtheta * tape, which the kernel has in
 *   registers): fenerf_siren_dtheta_floats(m, B*P) floats in all; and d_e [B*P][32] = gradient wrt the sampled grid
 *   features (NULL without a grid).
 * fenerf_siren_param_grads: (tape, d_t) -> every parameter gradient, written to the buffers of FenerfSirenGrads.  With all
 *   weight / bias pointers NULL only the FiLM gradients (d_freq_*, d_phase_*) are computed -- what inversion optimises
 *   (inverse_render_double_semantic.py:324-350).
 * fenerf_grid_backward: scatters d_e into a zero-initialised channels-last gradient grid d_grid_cl [D][H][W][32]
 *   (the transpose of sample_from_3dgrid, siren.py:314-330). */
typedef struct FenerfSirenGrads {   /* [dev] outputs, nn.Linear layout ([out][in] row-major) like FenerfModelDesc */
  float* geo_w[FENERF_MAX_GEO];     float* geo_b[FENERF_MAX_GEO];
  float* color_w[FENERF_MAX_COLOR]; float* color_b[FENERF_MAX_COLOR];   /* color_w[0]: columns [dir | grid feats | x] */
  float* head_w;   /* [32][H]: rows [0, n_lab) = gradient wrt the FOLDED label head (the product of label_layer_linear),
                      row n_lab = final_layer (sigma); the caller back-propagates the fold (tiny H x H products) */
  float* head_b;   /* [32] */
  float* rgb_w;    /* [3][H] color_layer_linear[0] */
  float* rgb_b;    /* [3] */
  float* d_freq_geo;  float* d_phase_geo;   /* [B][n_geo*H]   gradient wrt the RAW mapping-network outputs */
  float* d_freq_app;  float* d_phase_app;   /* [B][n_color*H] */
} FenerfSirenGrads;

size_t fenerf_siren_tape_floats(const FenerfModel* m, int64_t total_points);
size_t fenerf_siren_dtheta_floats(const FenerfModel* m, int64_t total_points);
/* The *_fmt entry points (round 5) are the calls above for a tape in `tape_format` (FENERF_TAPE_F32 = what the calls above use;
 * FENERF_TAPE_U16 = frac(theta) as 16-bit fixed point, FENERF_TAPE_F32_W = the fp32 tape with the frequency gradients left to the
 * weight-gradient stage; see the definition of the constants): the same tape format must be given to the
 * forward-save, the chain and the weight-gradient call of one evaluation.  replaces: the same reference lines as their plain
 * counterparts -- FiLMLayer saves its input and theta for autograd (siren.py:113-123); sin / cos of theta are all its backward reads.
 * fenerf_siren_tape_bytes: bytes of a tape (fenerf_siren_tape_floats * 4 or * 2).  fenerf_siren_param_grads_fmt with FENERF_TAPE_U16
 * also takes `weights`: the FiLM layers' weights [dev], nn.Linear layout, in the geo_w / color_w fields of a FenerfSirenGrads (read
 * only; every other field ignored) -- the FiLM frequency gradient sum_p d theta (W x + b) is then formed from the weight-gradient sums
 * sum_p d theta x^T and W, because a tape of phases does not hold W x; all weight / bias outputs are required (no FiLM-only mode). */
size_t fenerf_siren_tape_bytes(const FenerfModel* m, int64_t total_points, int tape_format);
int fenerf_siren_forward_save_fmt(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                                  const float* freq_geo, const float* phase_geo, const float* freq_app, const float* phase_app,
                                  float* out, void* tape, float* tape_e, void* film_ws, int tape_format, void* stream);
int fenerf_siren_backward_fmt(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                              const float* freq_app, const float* phase_app, const float* out, const float* d_out,
                              const void* tape, int tape_format, float* d_t, float* d_e, void* film_ws, void* stream);
int fenerf_siren_backward_grid_fmt(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                                   const float* freq_app, const float* phase_app, const float* out, const float* d_out,
                                   const void* tape, int tape_format, const float* points, float* d_t, float* d_grid_cl,
                                   float* scratch_d_e, void* film_ws, void* stream);
int fenerf_siren_param_grads_fmt(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                                 const float* freq_geo, const float* phase_geo, const float* freq_app, const float* phase_app,
                                 const float* out, const float* d_out, const void* tape, int tape_format, const float* tape_e,
                                 const float* d_t, const FenerfSirenGrads* g, const FenerfSirenGrads* weights, void* workspace,
                                 void* film_ws, void* stream);
int fenerf_siren_backward_stream_bytes_fmt(const FenerfModel* m, int64_t chunk_points, int tape_format, double* out4);

/* Per-point modulation under autograd (round 6) -- replaces: what torch autograd derives for
 * SPATIALSIRENGRID.forward_with_frequencies_phase_shifts (siren.py:464-477) when frequencies / phase_shifts hold one FiLM block PER SAMPLE
 * POINT (per-point mapping network, :440-462; FiLMLayer takes them unbroadcast, :119-122): d(out) -> gradients of every SIREN weight and bias
 * and of the per-point frequencies / phase shifts themselves (which the caller's mapping network backpropagates).
 * Like fenerf_siren_forward_save / fenerf_siren_backward / fenerf_siren_param_grads with freq_geo / phase_geo [B, P, n_geo*H] and freq_app /
 * phase_app [B, P, n_color*H] (raw mapping outputs, '*15+30' inside) and g->d_freq_geo / d_phase_geo [B, P, n_geo*H], g->d_freq_app /
 * d_phase_app [B, P, n_color*H] as outputs; every weight / bias buffer of `g` is required.  FENERF_PREC_F32 models created with
 * differentiable != 0, without a feature grid; P a multiple of 32.  tape: fenerf_siren_tape_floats(m, B*P) floats, d_t:
 * fenerf_siren_dtheta_floats(m, B*P) floats, film_ws: fenerf_film_workspace_bytes_pointwise(m, B, P) bytes, workspace:
 * fenerf_siren_grad_workspace_bytes(m, B, P) bytes, all [dev].  film_ws_prepared != 0: film_ws still holds what an earlier call of this
 * family wrote for the same FiLM tensors (the 2 L H floats per point are not prepared again).  Gradients wrt sample positions / view
 * directions are not provided. */
int fenerf_siren_forward_save_pointwise(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                                        const float* freq_geo, const float* phase_geo, const float* freq_app, const float* phase_app,
                                        float* out, float* tape, void* film_ws, void* stream);
int fenerf_siren_backward_pointwise(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                                    const float* freq_app, const float* phase_app, const float* out, const float* d_out,
                                    const float* tape, float* d_t, void* film_ws, int film_ws_prepared, void* stream);
int fenerf_siren_param_grads_pointwise(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                                       const float* freq_geo, const float* phase_geo, const float* freq_app, const float* phase_app,
                                       const float* out, const float* d_out, const float* tape, const float* d_t,
                                       const FenerfSirenGrads* g, void* workspace, void* film_ws, int film_ws_prepared, void* stream);

/* The differentiable hierarchical render as TWO calls (round 5) -- replaces: DoubleImplicitGenerator3d.forward / forward_with_frequencies
 * under autograd (generators.py:468-527, :735-797) and the part of g_loss.backward() (train_double_latent_semantic.py:402-446) /
 * loss.backward() (inverse_render_double_semantic.py:397) that runs through them.
 *
 * fenerf_render_forward_save = fenerf_render_forward (hierarchical, no fill mode) that also keeps what the backward needs in `save`
 *   (fenerf_render_save_bytes bytes of [dev] memory, opaque: prepared FiLM parameters, the two passes' sample points, outputs, tapes and
 *   sampled grid features, the resampled depths): coarse SIREN pass -> coarse weights + inverse-CDF resampling (constants of the graph,
 *   generators.py:485-503 is no_grad) -> fine pass -> merged composite.  Pixels and depth are those of fenerf_render_forward bit for bit.
 * fenerf_render_backward: g_rgb [B,R,C-1] = dL/d(out_rgb) -> every gradient of the render:
 *     grads->d_freq_* / d_phase_*  [B, n*H]   wrt the raw FiLM parameters (both passes summed),
 *     grads->geo_w .. rgb_b                   the weight / bias gradients in nn.Linear layout (head_w: the folded label head, as in
 *                                             fenerf_siren_param_grads) -- all NULL = FiLM gradients only (inversion),
 *     d_grid_ncdhw [1,32,D,H,W]               wrt spatial_embeddings (models with a grid; not for FiLM-only),
 *   from the same z_coarse / noise_final / opts the forward took.  It plans its own backward chunks (whole (pass, image) pairs or point
 *   ranges of at most chunk_points points, 0 = 393,216; FiLM-only launches of FENERF_PREC_F16X3 models by film_sums_budget_bytes of
 *   per-unit FiLM sums, 0 = 1 GiB), runs composite backward, the chain and the weight-gradient kernels per chunk and sums the chunks'
 *   gradients in chunk order.  workspace: fenerf_render_backward_workspace_bytes bytes of [dev] scratch (film_only as the call will be).
 *   `weights`: only for FENERF_TAPE_U16 (see fenerf_siren_param_grads_fmt). */
size_t fenerf_render_save_bytes(const FenerfModel* m, int B, int R, int N, int tape_format, int lock_view);
int fenerf_render_forward_save(const FenerfModel* m, int B, int R, int N, int lock_view, const float* origins, const float* dirs,
                               const float* z_coarse, const float* u, const float* noise_coarse, const float* noise_final,
                               const float* freq_geo, const float* phase_geo, const float* freq_app, const float* phase_app,
                               const FenerfCompositeOpts* opts, float* out_rgb, float* out_depth, void* save, size_t save_bytes,
                               int tape_format, void* stream);
size_t fenerf_render_backward_workspace_bytes(const FenerfModel* m, int B, int R, int N, int film_only, int64_t chunk_points,
                                              int64_t film_sums_budget_bytes);
int fenerf_render_backward(const FenerfModel* m, int B, int R, int N, int lock_view, const void* save, size_t save_bytes, int tape_format,
                           const float* z_coarse, const float* noise_final, const FenerfCompositeOpts* opts, const float* g_rgb,
                           const FenerfSirenGrads* grads, float* d_grid_ncdhw, const FenerfSirenGrads* weights, int64_t chunk_points,
                           int64_t film_sums_budget_bytes, void* workspace, size_t workspace_bytes, void* stream);

/* fenerf_render_backward cut in two, for data-parallel training (train_double_latent_semantic.py:148-150: DistributedDataParallel all-reduces
 * every generator gradient; 113 of the 124 MB are spatial_embeddings.grad).  That gradient is final as soon as the last chain launch has
 * scattered into it, with the last chunks' weight-gradient kernels still to run:
 *   stage 1 -- composite backward; chain AND weight gradients of every backward chunk but the last `keep_chunks`; the chains of those last
 *              chunks, each dump in its own slot of the workspace; d_grid_ncdhw finished.  The caller hands it to its all-reduce.
 *   stage 2 -- the weight gradients of the last chunks, the sums over chunks, the FiLM fold: everything else in `grads` finished.
 * Same arguments in both calls (g_rgb / z_coarse / noise_final / opts are read by stage 1 only and may be NULL in stage 2); `grads`' buffers
 * and the workspace (fenerf_render_backward_split_workspace_bytes: + one dump per kept chunk) must stay untouched between the two calls.
 * Weight gradients are required (no FiLM-only form).  Same kernels on the same chunks, every gradient summed in the same order:
 * stage 1 + stage 2 == fenerf_render_backward bit for bit (the atomically scattered grid gradient aside). */
size_t fenerf_render_backward_split_workspace_bytes(const FenerfModel* m, int B, int R, int N, int64_t chunk_points, int keep_chunks);
int fenerf_render_backward_stage(const FenerfModel* m, int stage, int keep_chunks, int B, int R, int N, int lock_view, const void* save,
                                 size_t save_bytes, int tape_format, const float* z_coarse, const float* noise_final,
                                 const FenerfCompositeOpts* opts, const float* g_rgb, const FenerfSirenGrads* grads, float* d_grid_ncdhw,
                                 const FenerfSirenGrads* weights, int64_t chunk_points, void* workspace, size_t workspace_bytes, void* stream);
/* HBM bytes per (sample point x FiLM-layer feature) of the backward streams of a chunk of `chunk_points` points (bench.py's generator-step
 * roofline): out[0] = what the chain kernel writes into the dump, out[1] = what the square weight-gradient job reads for one of its L - 1
 * layers (the dump of layer l + the input activations of layer l), out[2] = the four thin jobs together (two dump layers + two
 * tape layers), out[3] = the tape (fp32 pre-FiLM accumulators). */
int fenerf_siren_backward_stream_bytes(const FenerfModel* m, int64_t chunk_points, double* out4);
int fenerf_siren_forward_save(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                              const float* freq_geo, const float* phase_geo, const float* freq_app, const float* phase_app,
                              float* out, float* tape, float* tape_e, void* film_ws, void* stream);
int fenerf_siren_backward(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                          const float* freq_app, const float* phase_app, const float* out, const float* d_out,
                          const float* tape, float* d_t, float* d_e, void* film_ws, void* stream);
size_t fenerf_siren_grad_workspace_bytes(const FenerfModel* m, int B, int64_t P);
int fenerf_siren_param_grads(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                             const float* freq_geo, const float* phase_geo, const float* freq_app, const float* phase_app,
                             const float* out, const float* d_out, const float* tape, const float* tape_e, const float* d_t,
                             const FenerfSirenGrads* grads, void* workspace, void* film_ws, void* stream);
int fenerf_grid_backward(const FenerfModel* m, int64_t total_points, const float* points, const float* d_e, float* d_grid_cl,
                         void* stream);
/* Inversion steps (inverse_render_double_semantic.py:324-410: the weights are frozen, only the per-image FiLM offsets take gradients;
 * what torch autograd does there is the same backward as in training with the weight-gradient products dropped).  FENERF_PREC_F16X3
 * models only (FENERF_E_UNSUPPORTED otherwise -- use fenerf_siren_backward + fenerf_siren_param_grads with NULL weight pointers):
 *   fenerf_siren_backward_film   the backward chain WITHOUT its d(theta) dump and without d(grid features): writes only the per-tile
 *                                FiLM sums, film_sums [fenerf_siren_film_sums_floats(m, B, P)] floats (opaque) -- no 11-KB-per-point
 *                                buffer, so a whole pass can be one launch;
 *   fenerf_siren_film_grads      film_sums -> d_freq_* / d_phase_* of `grads` (its weight / bias pointers are ignored); workspace as for
 *                                fenerf_siren_param_grads. */
size_t fenerf_siren_film_sums_floats(const FenerfModel* m, int B, int64_t P);
int fenerf_siren_backward_film(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                               const float* freq_app, const float* phase_app, const float* out, const float* d_out,
                               const float* tape, float* film_sums, void* film_ws, void* stream);
int fenerf_siren_film_grads(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                            const float* freq_app, const float* phase_app, const float* film_sums,
                            const FenerfSirenGrads* grads, void* workspace, void* film_ws, void* stream);
/* fenerf_siren_backward + fenerf_grid_backward in one call: d_t as above, and the gradient wrt the sampled grid features is
 * scattered (accumulated) into d_grid_cl [D][H][W][32] -- grid_sample's backward (siren.py:314-330) -- instead of being returned.
 * points [B*P][3] as given to the forward.  Models whose chain kernel scatters in place (fenerf_siren_backward_fuses_grid != 0:
 * FENERF_PREC_F16X3 with a grid) need no scratch; otherwise scratch_d_e [B*P][32] is required and the call runs the two steps. */
int fenerf_siren_backward_fuses_grid(const FenerfModel* m);
int fenerf_siren_backward_grid(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                               const float* freq_app, const float* phase_app, const float* out, const float* d_out,
                               const float* tape, const float* points, float* d_t, float* d_grid_cl, float* scratch_d_e,
                               void* film_ws, void* stream);
/* channels-last gradient grid [D][H][W][32] -> the parameter's layout [1,32,D,H,W] (spatial_embeddings.grad) */
int fenerf_grid_gradient_ncdhw(const FenerfModel* m, const float* d_grid_cl, float* d_grid_ncdhw, void* stream);
/* replaces: what torch autograd leaves in `input.grad` / `ray_directions.grad` of forward_with_frequencies_phase_shifts
 * (siren.py:1509-1530) for a caller that asked for them: d input = W_0^T dz_0 (:1517-1520) + grid_sample's backward wrt its coordinates
 * (sample_from_3dgrid, siren.py:314-330, applied to d shared_features = W_c0[:, 3:35]^T dz_c0), times UniformBoxWarp's 2 / 0.24
 * (siren.py:181-187, :1513); d ray_directions = W_c0[:, 0:3]^T dz_c0 (the cat of :1522).  The generator API never needs it: the
 * reference builds its rays under torch.no_grad() (generators.py:465, :483); this is for callers of the bare SIREN module.
 * Call it between fenerf_siren_backward* (which leaves d_t) and fenerf_siren_param_grads, with the same (B, P), points and FiLM
 * parameters.  d_t must be an fp32 dump (FENERF_E_UNSUPPORTED for the bf16 dump of AMP-class chunks).  w_geo0 [H][3] and w_color0
 * [H][w_color0_ld] are the nn.Linear weights of layer 0 and of colour layer 0 (columns [dirs 3 | grid features | x]; only the first
 * 3 + grid channels are read) on the device, at the model's hidden width (zero rows for a padded width).  d_points / d_dirs [B*P][3],
 * either may be NULL.  film_ws as for fenerf_siren_backward. */
int fenerf_siren_input_grads(const FenerfModel* m, int B, int64_t P, const float* points, const float* freq_geo, const float* phase_geo,
                             const float* freq_app, const float* phase_app, const float* d_t, const float* w_geo0,
                             const float* w_color0, int w_color0_ld, float* d_points, float* d_dirs, void* film_ws, void* stream);

/* replaces: what torch autograd derives for the final fancy_integration of a differentiable render
 * (generators.py:519 / :790; G-step and inversion): gradient wrt rgb_final g_rgb [BR, C-1] -> gradients wrt the SIREN
 * outputs.  merge = 0: rows_a [BR,N,C], z_a [BR,N] -> d_rows_a [BR,N,C].  merge = 1: fine rows_a / coarse rows_b with
 * z_a / z_b as in fenerf_merge_composite -> d_rows_a (fine), d_rows_b (coarse), each in its own (unsorted) order.
 * noise is indexed by sorted position like the forward.  fill_mode must be FENERF_FILL_NONE; depth is not differentiated. */
int fenerf_composite_backward(int64_t BR, int N, int C, int merge, const float* rows_a, const float* rows_b,
                              const float* z_a, const float* z_b, const float* noise, const FenerfCompositeOpts* opts,
                              const float* g_rgb, float* d_rows_a, float* d_rows_b, void* stream);

/* Bytes of [dev] scratch fenerf_sparse_select needs for B images of P = R * N samples per pass. */
size_t fenerf_sparse_select_workspace_bytes(int B, int64_t P);

/* replaces: nothing the reference writes down -- its autograd (train_double_latent_semantic.py:402-446, the backward of
 * generators.py:479-519) multiplies the all-zero gradient rows of empty-space samples through the whole SIREN; this is the selection
 * step of a backward that does not (exact: a sample whose row of d loss / d SIREN outputs is all zero contributes exact zeros to every
 * gradient; under the relu clamp of volumetric_rendering.py:36-47 that is every sample with sigma + noise <= 0).
 * d_coarse / d_fine [B][P][C] are fenerf_composite_backward's outputs (merge = 1: d_rows_b, d_rows_a), P = R * N; z_coarse / z_fine
 * [B*R][N]; origins / dirs [B][R][3].  Per image, the samples with a non-zero row (NaN counts as non-zero) in sample order (coarse pass
 * first) fill slots 0 .. counts[b] - 1 of pts / rd [B][cap][3] (origins + dirs * z, generators.py:504; rd may be NULL) and d_sel
 * [B][cap][C]; the remaining slots repeat the image's first sample with a zero row.  counts [B + 1] (int32, device): kept samples per
 * image, then a flag that is 1 when some image kept more than cap (the excess is dropped: treat as an error).  Feed pts / rd / d_sel to
 * fenerf_siren_forward_save + fenerf_siren_backward + fenerf_siren_param_grads with P = cap (a multiple of 32).
 * d_fine = z_fine = NULL: a render without importance resampling (generators.py:479-519 with hierarchical_sample False, the
 * reference's inversion renders): one pass, P samples per image.
 * images (device, may be NULL = 0 .. B - 1): image b of this call is image images[b] of the input arrays -- a batch whose images keep
 * very different numbers of samples is walked in groups of similar images, each with its own cap, instead of padding every image to
 * the fullest one. */
int fenerf_sparse_select(int B, int R, int N, int C, int64_t cap, const float* d_coarse, const float* d_fine, const float* z_coarse,
                         const float* z_fine, const float* origins, const float* dirs, const int64_t* images, float* pts, float* rd,
                         float* d_sel, int32_t* counts, void* workspace, size_t workspace_bytes, void* stream);

/* Bytes of [dev] scratch fenerf_render_forward needs. */
size_t fenerf_render_workspace_bytes(const FenerfModel* m, int B, int R, int N, int hierarchical);

/* replaces: the body of DoubleImplicitGenerator3d.forward / staged_forward / *_with_frequencies after the rays
 * exist (generators.py:479-519, :583-637, :666-724, :753-791) = coarse SIREN -> composite -> resample ->
 * fine SIREN -> merge -> composite, for B images of R rays with N coarse (+N fine) samples.
 * origins/dirs [B,R,3] world space, z_coarse [B,R,N] (already jittered), u [B*R,N], noise_coarse [B,R,N] / noise_final
 * [B,R,2N or N] (NULL when nerf_noise == 0).  Outputs as fenerf_merge_composite.  `opts` applies to the FINAL
 * composite; the coarse one uses clamp_mode + noise_std only, like the reference. */
int fenerf_render_forward(const FenerfModel* m, int B, int R, int N, int hierarchical, int lock_view,
                          const float* origins, const float* dirs, const float* z_coarse, const float* u,
                          const float* noise_coarse, const float* noise_final, const float* freq_geo,
                          const float* phase_geo, const float* freq_app, const float* phase_app,
                          const FenerfCompositeOpts* opts, float* out_rgb, float* out_depth, float* out_weights,
                          float* out_wsum, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FENERF_H_ */
