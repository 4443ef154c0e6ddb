// Internal declarations shared by the host API, the packer and the kernel launchers.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fenerf.h"
#include "fenerf_layout.h"

namespace fenerf {

void set_error(const std::string& msg);
// Opt-in to > 64 KiB of dynamic LDS for kernel `kfn`.  hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute:
// what has been granted is remembered per (kernel, current device) under a lock, so a process that renders on several GPUs or
// from several host threads always launches with the opt-in in place (include/fenerf.h: thread-safe per handle).
// Returns FENERF_OK or FENERF_E_HIP (error string set).
int ensure_dynamic_lds(const void* kfn, size_t bytes);
// Per-phase device timing (fenerf_phase_timing / fenerf_phase_times): a PhaseScope around a launch group records a hipEvent pair on
// the stream when timing is on, and is two loads of a flag when it is off.
enum Phase { PH_FILM_PREP = 0, PH_SIREN, PH_SIREN_SAVE, PH_CHAIN, PH_WGRAD_FILM, PH_WGRAD_SQ, PH_WGRAD_SQ_REDUCE, PH_WGRAD_THIN,
             PH_WGRAD_THIN_REDUCE, PH_COMPOSITE, PH_RESAMPLE, PH_COMPOSITE_BWD, PH_REPACK, PH_GRID, PH_RAY_SETUP, PH_OTHER, PH_RENDER_FUSED, PH_COUNT };
static_assert(PH_COUNT == FENERF_N_PHASES, "include/fenerf.h FENERF_N_PHASES");
struct PhaseScope {
  PhaseScope(int phase, void* stream);
  ~PhaseScope();
  void* rec;
};
// compute units the calling thread's launches are sized for: the device's, or fewer under fenerf_set_cu_budget
int launch_cus(const FenerfModel* m);
// once per device: v_sin_f32 / v_cos_f32 reduce arguments far beyond +-256 revolutions on this device (fenerf_trig.h); FENERF_OK, or
// FENERF_E_UNSUPPORTED / FENERF_E_HIP with the error string set
int check_trig_domain();
int validate_desc(const FenerfModelDesc* d, std::string& err);
int pack_weights(const FenerfModelDesc* d, std::vector<float>& blob, std::vector<float>& consts, std::string& err);
int pack_weights_f16(const FenerfModelDesc* d, std::vector<float>& blob, std::vector<float>& consts, std::string& err);
int pack_weights_bwd(const FenerfModelDesc* d, std::vector<float>& blob, std::string& err);
int pack_weights_bwd16(const FenerfModelDesc* d, std::vector<float>& blob, std::string& err, std::vector<int32_t>* index);
int pack_local_weights(const FenerfModelDesc* d, const FenerfLocalMapDesc* mp, std::vector<float>& blob, std::vector<float>& consts,
                       std::string& err);   // SPATIALSIRENGRID: per-point mapping network + SIREN in one stream (fenerf_siren_local.hip)

}  // namespace fenerf

// Opaque model handle: device-resident packed weights + shape info.
struct FenerfModel {
  int H, n_geo, n_color, n_lab, C, L;
  int grid_ch, gd, gh, gw;
  float box_scale;
  fenerf::StreamShape sh;
  float* d_stream;  // [l0 entries | ring stream] * 256 floats
  float* d_consts;  // head bias | rgb bias | film biases
  float* d_grid;    // channels-last [D][H][W][32] or nullptr
  int num_cus;
  int precision;    // FENERF_PREC_*
  int differentiable;       // desc->differentiable: the backward-chain stream is resident too
  int forward_mode;         // FENERF_FORWARD_*: the no-grad forward's arithmetic (fenerf_model_set_forward_mode; FENERF_PREC_F16X3 models)
  int wgrad_bf16_min_points; // desc->wgrad_bf16_min_points: 0 = fp32-class weight gradients; > 0 = bf16 dump for chunks of at least that many points
  fenerf::BwdShape bsh;
  float* d_bwd_stream;      // [rgb-head^T entries | backward ring] * 256 floats, or nullptr
  float* d_row_scale;       // fenerf_model_repack scratch: [2][L*H + 64] row scales (forward | backward), lazily allocated
  size_t n_stream, n_consts, n_bwd;   // floats resident in d_stream / d_consts / d_bwd_stream
  // FENERF_PREC_F16X3 models: the exact-fp32 stream / consts too (host-packed at create / update), for fenerf_siren_forward_pointwise --
  // per-point FiLM blocks are read per lane by the fp32 kernel; a device-side re-pack does not refresh them (stream32_valid = 0)
  float* d_stream32;
  float* d_consts32;
  int stream32_valid;
  // AMP-class models (wgrad_bf16_min_points > 0) only: the d(theta) dumps fenerf_siren_backward* wrote, buffer -> points of that chunk.
  // The dump's format (fp32 | bf16) is a function of the chunk's point count and nothing in the buffer records it, so
  // fenerf_siren_param_grads looks the buffer up and refuses to read it as the other format (note_dump / check_dump, fenerf_api.cpp)
  mutable std::mutex dump_mu;
  mutable std::map<const void*, long long> dump_points;
};

namespace fenerf {

// Kernel parameter blocks (plain structs passed by value).
struct SirenParams {
  const float* stream;     // packed weights
  const float* consts;     // head bias [32] | rgb bias [4] | ...
  const float* fp;         // [B][L][H]  f' = (15 f + 30) / 2pi
  const float* pp;         // [B][L][H]  p' = (f b + p) / 2pi
  const float* grid;       // channels-last grid or nullptr
  int gd, gh, gw;
  float box_scale;
  // inputs: explicit points (points != nullptr) or rays
  const float* points;     // [P][3]
  const float* pdirs;      // [P][3] or nullptr (lock)
  const float* origins;    // [B*R][3]
  const float* dirs;       // [B*R][3]
  const float* z;          // [B*R][N]
  int n_per_ray;
  int lock_view;
  long long P;             // total points = B * pts_per_image
  long long pts_per_image;
  float* out;              // [P][C]
  long long ring_offset_floats;  // offset of the ring stream inside `stream`
  // differentiable evaluation (fenerf_siren_forward_save): pre-FiLM accumulators W x (no bias) of every FiLM layer as
  // register dumps of the 32-point tiles (fenerf_layout.h "Tape"), and the sampled grid features [P][32]
  float* tape;
  float* tape_e;
  int tape_format;         // FENERF_TAPE_F32 | FENERF_TAPE_U16 (FENERF_PREC_F16X3 models; fenerf_layout.h "16-bit tape")
  // SPATIALSIRENGRID (siren.py:413-518): fp / pp hold one [L][H] block per POINT instead of per image (fp32 kernel only)
  int film_per_point;
  // fenerf_siren_clock_probe: [gridDim.x][4] = s_memtime / s_memrealtime at a workgroup's first and last instruction, or nullptr
  unsigned long long* clk;
  // FiLM pre-pass inside the launch (fenerf_film.h): with raw_fg != nullptr fp / pp are OUTPUTS first -- every workgroup computes
  // the blocks of the images its tiles belong to from the raw mapping-network outputs [n_images][n_geo*H] / [n_images][n_color*H]
  const float* raw_fg; const float* raw_pg; const float* raw_fa; const float* raw_pa;
  const float* film_bias;        // [L][H] FiLM-layer biases
  const float* film_inv_scale;   // [L][H] result scale of the layer's GEMM (FENERF_PREC_F16X3) or nullptr
  int n_images;
};

struct SirenBwdParams {
  const float* stream;     // backward stream
  long long ring_offset_floats;
  const float* fp;         // [B][L][H] as the forward
  const float* pp;
  long long P, pts_per_image;
  const float* out;        // [P][C] forward outputs (sigmoid' of the rgb head)
  const float* d_out;      // [P][C] gradient wrt the outputs
  const float* tape;       // from the forward (tape layout)
  int tape_format;         // FENERF_TAPE_F32 | FENERF_TAPE_U16 (siren_bwd16w_kernel; the FiLM sums then carry no sum of d theta * tape)
  float* d_t;              // out, tape layout: dL/d(theta_l) = dx_l * cos(theta_l), theta = f (W x + b) + p
  int bf16_dump;           // siren_bwd16w_kernel: write the dump as bf16 [d theta | x = sin(2 pi theta)] halves instead (fenerf_layout.h "bf16 dump")
  float* d_e;              // [P][32] out: gradient wrt the sampled grid features (nullptr without a grid)
  float* film_tiles;       // [tiles][L][2][H] out: per-tile FiLM sums (fenerf_layout.h "FiLM sums")
  // fused grid scatter (siren_bwd16w_kernel only): when d_grid_cl != nullptr the gradient wrt the sampled grid features is not
  // written to d_e but scattered straight into the channels-last gradient grid (the transpose of sample_from_3dgrid)
  const float* points;     // [P][3]
  float* d_grid_cl;        // [gd][gh][gw][32], accumulated into
  float box_scale;
  int gd, gh, gw;
  // SPATIALSIRENGRID (siren.py:413-518) under autograd: fp / pp hold one [L][H] block per POINT (siren_bwd_kernel only; round 6)
  int film_per_point;
};

struct CompositeParams {
  long long BR;
  int M, C, N;             // M samples composited; merge: M = 2N
  const float* rows_a;     // non-merge: rgb_sigma [BR][M][C]; merge: fine [BR][N][C]
  const float* rows_b;     // merge: coarse [BR][N][C]
  const float* z_a;        // non-merge: z [BR][M]; merge: z_fine [BR][N]
  const float* z_b;        // merge: z_coarse [BR][N]
  const float* noise;      // [BR][M] or nullptr
  FenerfCompositeOpts o;
  float* out_rgb; float* out_depth; float* out_weights; float* out_wsum; float* out_z;
  int out_ch;              // channels written per ray (C-1 or C)
  int sigma_only;          // coarse pass: only weights wanted, skip colour accumulation
  // fused inverse-CDF resampling (non-merge, generators.py:486-499): with z_fine != nullptr the wave that has just computed the
  // ray's coarse weights also draws the N fine depths from them (u [BR][N]) -- the weights never leave LDS
  const float* u;
  float* z_fine;
  // backward (fenerf_composite_backward): gradient of the loss wrt out_rgb [BR][C-1] in, wrt the rows out
  const float* g_rgb;
  float* d_rows_a;         // non-merge: [BR][M][C]; merge: d fine [BR][N][C]
  float* d_rows_b;         // merge: d coarse [BR][N][C]
};

int launch_film_prep(const FenerfModel* m, long long B, const float* fg, const float* pg, const float* fa, const float* pa,
                     float* fp, float* pp, void* stream, bool for_f32_kernel = false,    // for_f32_kernel: biases of d_consts32, no GEMM result scale
                     bool twice = false);                                                // twice: rows [B, 2B) of fp / pp receive a copy of rows [0, B)
int launch_siren(const FenerfModel* m, const SirenParams& p, void* stream);     // dispatches on m->precision
int launch_siren_f32(const FenerfModel* m, const SirenParams& p, void* stream); // the exact-fp32 kernel on whatever stream `p` points at
int launch_siren_backward(const FenerfModel* m, const SirenBwdParams& p, void* stream);
int launch_siren_backward16w(const FenerfModel* m, const SirenBwdParams& p, void* stream);  // FENERF_PREC_F16X3 models: bf16x3 chain on 16-point waves (fenerf_siren_bwd16w.hip)
bool use_bf16_dump(const FenerfModel* m, long long total_points);   // the dump format of a backward chunk (fenerf_layout.h "bf16 dump"); chain and weight-gradient launches ask the same question
int bwd16w_film_unit(long long total_points, long long pts_per_image);   // points per FiLM-sum unit of that kernel: 128 (workgroup) or 16 (wave)
size_t wgrad_workspace_bytes(const FenerfModel* m, int B, long long P);
// tape_format FENERF_TAPE_U16: `weights` = the FiLM layers' weights [dev], nn.Linear layout, in the geo_w / color_w fields (the FiLM
// frequency gradients are then derived from the weight-gradient partial sums: fenerf_siren_wgrad.hip "frequency gradients without the tape")
int launch_param_grads(const FenerfModel* m, int B, long long P, const float* points, const float* dirs, const float* fp, const float* pp,
                       const float* out, const float* d_out, const float* tape, const float* tape_e, const float* d_t,
                       const FenerfSirenGrads& g, bool film_only, void* workspace, void* stream, const float* film_tiles = nullptr,
                       int tape_format = 0, const FenerfSirenGrads* weights = nullptr,
                       int film_per_point = 0);    // film_per_point: fp / pp are [B*P][L][H]; g.d_freq_* / d_phase_* are [B*P][n*H] (FENERF_PREC_F32 models)
int launch_grid_backward(const FenerfModel* m, long long P, const float* points, const float* d_e, float* d_grid_cl, void* stream);
// fenerf_siren_inputgrad.hip: d points / d dirs from the fp32 d(theta) dump (layer 0 and colour layer 0) + grid_sample's coordinate gradient
int launch_siren_input_grads(const FenerfModel* m, int B, long long P, const float* points, const float* fp, const float* d_t, const float* w_geo0,
                             const float* w_color0, int w_color0_ld, float* d_points, float* d_dirs, void* stream);
int launch_siren16w(const FenerfModel* m, const SirenParams& p, void* stream);   // f16x3 forward / forward-save, 16-point waves (fenerf_siren_f16w.hip)
// fenerf_render_forward as ONE launch (fenerf_siren_f16w.hip, FUSED): ray groups of whole octs, see there
struct FusedRenderPlan { int rays_per_group, octs_per_group, blocks; long long groups; };
bool fused_render_plan(const FenerfModel* m, long long B, long long R, int N, bool balanced_only, FusedRenderPlan* plan);
int launch_render16w_fused(const FenerfModel* m, const SirenParams& p, const FusedRenderPlan& plan, float* z_fine, float* out_fine,
                           const CompositeParams& coarse, const CompositeParams& final_, void* stream);
int launch_composite(const CompositeParams& p, bool merge, void* stream);
int launch_composite_backward(const CompositeParams& p, bool merge, void* stream);
int launch_resample(long long BR, int N, const float* z, const float* w, const float* u, float* zf, void* stream);
int launch_sample_pdf(long long BR, int K, int NS, const float* bins, const float* w, const float* u, float* out, void* stream);
int launch_ray_setup(int B, int S, int N, float z_cam, float ray_start, float ray_end, const float* u_jitter,
                     const float* theta, const float* phi, float* origins, float* dirs, float* z, float* pitch, float* yaw,
                     void* stream);
// fenerf_render_grad.hip: the small kernels of fenerf_render_forward_save / fenerf_render_backward
#define FENERF_MULTI_ADD_MAX 64
struct MultiAdd { float* dst[FENERF_MULTI_ADD_MAX]; const float* src[FENERF_MULTI_ADD_MAX]; long long n[FENERF_MULTI_ADD_MAX]; int count; };
struct FilmFold { float* out[4]; const float* in[4]; int row[4]; int B; };
int launch_render_points(int B, int R, int N, long long Pp, const float* origins, const float* dirs, const float* z, float* pts, float* rd, void* stream);
size_t sparse_select_workspace_bytes(int B, long long P);
int launch_sparse_select(int B, int R, int N, int C, long long cap, const float* d_coarse, const float* d_fine, const float* z_coarse,
                         const float* z_fine, const float* origins, const float* dirs, const long long* images, float* pts, float* rd, float* d_sel,
                         int* counts, void* workspace, void* stream);
int launch_pad_rows(const float* src, float* dst, long long nb, long long P, long long Pp, int C, bool to_padded, void* stream);
int launch_multi_add(const MultiAdd& J, void* stream);
int launch_film_fold(const FilmFold& J, void* stream);
int launch_repack(FenerfModel* m, const float* flat, const FenerfRepackMaps* r, float* scale_fwd, float* scale_bwd, void* stream);
int launch_grid_relayout(const float* src_ncdhw, float* dst_cl, int C, int D, int Hh, int W, void* stream);
int launch_grid_unlayout(const float* src_cl, float* dst_ncdhw, int D, int Hh, int W, void* stream);

}  // namespace fenerf
