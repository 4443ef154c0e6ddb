// C-ABI entry points (include/fenerf.h): argument validation, model handle, stage orchestration.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "fenerf_internal.h"

namespace fenerf {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

static int fail(int code, const std::string& msg) {
  set_error(msg);
  return code;
}
static int hip_fail(hipError_t e, const char* what) {
  return fail(FENERF_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t _e = (expr);                             \
    if (_e != hipSuccess) return hip_fail(_e, #expr);   \
  } while (0)

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int ensure_dynamic_lds(const void* kfn, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> granted;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = granted[{kfn, dev}];
  if (bytes > have) {
    e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(max dynamic LDS)");
    have = bytes;
  }
  return FENERF_OK;
}

// ---- per-phase device timing (include/fenerf.h: fenerf_phase_timing / fenerf_phase_times)
namespace {
std::atomic<int> g_phase_on{0};
struct PhaseRec { int phase; hipEvent_t e0, e1; hipStream_t st; };
std::mutex g_phase_mu;
std::vector<PhaseRec*> g_phase_recs;
const char* const kPhaseNames[PH_COUNT] = {"film_prep", "siren_forward", "forward_save", "chain", "wgrad_film_sums", "wgrad_square",
                                           "wgrad_square_reduce", "wgrad_thin", "wgrad_thin_reduce", "composite", "resample",
                                           "composite_backward", "repack", "grid", "ray_setup", "other", "render_fused"};
}  // namespace
PhaseScope::PhaseScope(int phase, void* stream) : rec(nullptr) {
  if (!g_phase_on.load(std::memory_order_relaxed)) return;
  PhaseRec* r = new (std::nothrow) PhaseRec{phase, nullptr, nullptr, (hipStream_t)stream};
  if (!r) return;
  if (hipEventCreate(&r->e0) != hipSuccess) { delete r; return; }
  if (hipEventCreate(&r->e1) != hipSuccess) { (void)hipEventDestroy(r->e0); delete r; return; }
  (void)hipEventRecord(r->e0, r->st);
  rec = r;
}
PhaseScope::~PhaseScope() {
  if (!rec) return;
  PhaseRec* r = static_cast<PhaseRec*>(rec);
  (void)hipEventRecord(r->e1, r->st);
  std::lock_guard<std::mutex> lock(g_phase_mu);
  if (g_phase_recs.size() >= 65536) {      // timing left on and never drained (fenerf_phase_times): keep the newest, bounded
    PhaseRec* old = g_phase_recs.front();
    g_phase_recs.erase(g_phase_recs.begin());
    (void)hipEventDestroy(old->e0);
    (void)hipEventDestroy(old->e1);
    delete old;
  }
  g_phase_recs.push_back(r);
}

static thread_local int g_cu_budget = 0;
int launch_cus(const FenerfModel* m) { return (g_cu_budget > 0 && g_cu_budget < m->num_cus) ? g_cu_budget : m->num_cus; }

// the dump format of a backward chunk (fenerf_layout.h "bf16 dump"): opt-in per model (FenerfModelDesc.wgrad_bf16_min_points)
bool use_bf16_dump(const FenerfModel* m, long long total_points) {
  return m && m->precision == FENERF_PREC_F16X3 && m->wgrad_bf16_min_points > 0 && total_points >= m->wgrad_bf16_min_points;
}

// AMP-class models: remember which chunk size (hence which dump format) a d(theta) buffer was written with ...
static void note_dump(const FenerfModel* m, const void* d_t, long long points) {
  if (m->wgrad_bf16_min_points <= 0) return;
  std::lock_guard<std::mutex> lock(m->dump_mu);
  if (m->dump_points.size() >= 256) m->dump_points.clear();     // buffers of long-gone steps; a miss only skips the check
  m->dump_points[d_t] = points;
}
// ... and refuse to read it as the other one (a caller that splits the backward and the weight-gradient call differently)
static int check_dump(const FenerfModel* m, const void* d_t, long long points) {
  if (m->wgrad_bf16_min_points <= 0) return FENERF_OK;
  std::lock_guard<std::mutex> lock(m->dump_mu);
  auto it = m->dump_points.find(d_t);
  if (it == m->dump_points.end() || use_bf16_dump(m, it->second) == use_bf16_dump(m, points)) return FENERF_OK;
  return fail(FENERF_E_INVALID, "fenerf_siren_param_grads: this d(theta) dump was written by a backward call over " + std::to_string(it->second) +
                                " points (" + (use_bf16_dump(m, it->second) ? "bf16" : "fp32") + " format, wgrad_bf16_min_points " +
                                std::to_string(m->wgrad_bf16_min_points) + ") and is read here as a chunk of " + std::to_string(points) +
                                " points (the other format): call both with the same (B, P)");
}

static int check_opts(const FenerfCompositeOpts* o) {
  if (!o) return fail(FENERF_E_INVALID, "opts is NULL");
  if (o->clamp_mode != FENERF_CLAMP_RELU && o->clamp_mode != FENERF_CLAMP_SOFTPLUS)
    return fail(FENERF_E_CLAMP_MODE, "Need to choose clamp mode");  // volumetric_rendering.py:34
  if (o->fill_mode < FENERF_FILL_NONE || o->fill_mode > FENERF_FILL_EVAL_WHITE_BACK)
    return fail(FENERF_E_INVALID, "unknown fill_mode");
  return FENERF_OK;
}

static int out_channels(int C, const FenerfCompositeOpts* o) {
  const bool pad = o->fill_mode == FENERF_FILL_SEG_PADDING_BACKGROUND || o->fill_mode == FENERF_FILL_EVAL_SEG_PADDING_BACKGROUND;
  return pad ? C : C - 1;
}

static int upload_model(FenerfModel* m, const FenerfModelDesc* d, hipStream_t stream, bool allocate) {
  std::vector<float> blob, consts;
  std::string err;
  int rc = d->precision == FENERF_PREC_F16X3 ? pack_weights_f16(d, blob, consts, err) : pack_weights(d, blob, consts, err);
  if (rc) return fail(rc, err);
  if (allocate) {
    HIP_TRY(hipMalloc((void**)&m->d_stream, blob.size() * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&m->d_consts, consts.size() * sizeof(float)));
  }
  m->n_stream = blob.size(); m->n_consts = consts.size();
  HIP_TRY(hipMemcpyAsync(m->d_stream, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemcpyAsync(m->d_consts, consts.data(), consts.size() * sizeof(float), hipMemcpyHostToDevice, stream));
  if (d->precision == FENERF_PREC_F16X3) {   // the exact-fp32 stream too: fenerf_siren_forward_pointwise (per-point FiLM blocks)
    std::vector<float> blob32, consts32;
    rc = pack_weights(d, blob32, consts32, err);
    if (rc) return fail(rc, err);
    if (allocate) {
      HIP_TRY(hipMalloc((void**)&m->d_stream32, blob32.size() * sizeof(float)));
      HIP_TRY(hipMalloc((void**)&m->d_consts32, consts32.size() * sizeof(float)));
    }
    HIP_TRY(hipMemcpyAsync(m->d_stream32, blob32.data(), blob32.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(m->d_consts32, consts32.data(), consts32.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));   // host vectors go out of scope
    m->stream32_valid = 1;
  }
  std::vector<float> bwd;
  if (m->differentiable) {
    rc = d->precision == FENERF_PREC_F16X3 ? pack_weights_bwd16(d, bwd, err, nullptr) : pack_weights_bwd(d, bwd, err);
    if (rc) return fail(rc, err);
    m->n_bwd = bwd.size();
    if (allocate) HIP_TRY(hipMalloc((void**)&m->d_bwd_stream, bwd.size() * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(m->d_bwd_stream, bwd.data(), bwd.size() * sizeof(float), hipMemcpyHostToDevice, stream));
  }
  if (d->grid_ch && d->grid) {
    const size_t n = (size_t)d->grid_ch * d->grid_d * d->grid_h * d->grid_w;
    if (allocate) HIP_TRY(hipMalloc((void**)&m->d_grid, n * sizeof(float)));
    float* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&tmp, n * sizeof(float)));
    hipError_t e = hipMemcpyAsync(tmp, d->grid, n * sizeof(float), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
      rc = [&] { PhaseScope ph(PH_GRID, stream); return launch_grid_relayout(tmp, m->d_grid, d->grid_ch, d->grid_d, d->grid_h, d->grid_w, stream); }();
      if (rc == FENERF_OK) e = hipStreamSynchronize(stream);
    }
    (void)hipFree(tmp);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(e, "grid upload");
  } else if (d->grid_ch && allocate) {
    return fail(FENERF_E_INVALID, "grid_ch > 0 but grid pointer is NULL");
  }
  HIP_TRY(hipStreamSynchronize(stream));  // blob/consts are stack-owned host vectors
  return FENERF_OK;
}
}  // namespace fenerf

using namespace fenerf;

extern "C" const char* fenerf_last_error(void) { return g_err.c_str(); }

extern "C" int fenerf_phase_timing(int enable) { return g_phase_on.exchange(enable ? 1 : 0); }
extern "C" const char* fenerf_phase_name(int phase) { return phase >= 0 && phase < PH_COUNT ? kPhaseNames[phase] : ""; }
extern "C" int fenerf_phase_times(double* ms, int* calls, int n) {
  if (!ms || !calls || n < PH_COUNT) return fail(FENERF_E_INVALID, "fenerf_phase_times: need ms / calls arrays of FENERF_N_PHASES entries");
  std::vector<PhaseRec*> recs;
  {
    std::lock_guard<std::mutex> lock(g_phase_mu);
    recs.swap(g_phase_recs);
  }
  int rc = FENERF_OK;
  for (PhaseRec* r : recs) {
    float t = 0.f;
    hipError_t e = hipEventSynchronize(r->e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&t, r->e0, r->e1);
    if (e == hipSuccess) { ms[r->phase] += (double)t; calls[r->phase] += 1; }
    else rc = hip_fail(e, "fenerf_phase_times");
    (void)hipEventDestroy(r->e0);
    (void)hipEventDestroy(r->e1);
    delete r;
  }
  return rc;
}
extern "C" int fenerf_abi_version(void) { return FENERF_ABI_VERSION; }
static thread_local int g_render_fusion = FENERF_FUSION_AUTO;
// AUTO takes the one-launch route only if it has been measured to be at least as fast as the four launches (profiles/r04_render_one_launch.md)
static constexpr bool kFusionAutoEnabled = false;
extern "C" int fenerf_set_render_fusion(int mode) {
  const int prev = g_render_fusion;
  g_render_fusion = (mode == FENERF_FUSION_OFF || mode == FENERF_FUSION_FORCE) ? mode : FENERF_FUSION_AUTO;
  return prev;
}
extern "C" int fenerf_set_cu_budget(int cus) {
  const int prev = g_cu_budget;
  g_cu_budget = cus > 0 ? cus : 0;
  return prev;
}

// ---- struct layouts as compiled (include/fenerf.h fenerf_struct_*): bindings check themselves against the loaded library
namespace {
struct FieldInfo { const char* name; long offset; };
struct StructInfo { const char* name; long size; const FieldInfo* fields; int n; };
#define FLD(S, f) {#f, (long)offsetof(S, f)}
const FieldInfo kDescFields[] = {
    FLD(FenerfModelDesc, abi_version), FLD(FenerfModelDesc, hidden_dim), FLD(FenerfModelDesc, n_geo), FLD(FenerfModelDesc, n_color),
    FLD(FenerfModelDesc, n_label_layers), FLD(FenerfModelDesc, output_dim), FLD(FenerfModelDesc, grid_ch), FLD(FenerfModelDesc, grid_d),
    FLD(FenerfModelDesc, grid_h), FLD(FenerfModelDesc, grid_w), FLD(FenerfModelDesc, box_scale), FLD(FenerfModelDesc, geo_w),
    FLD(FenerfModelDesc, geo_b), FLD(FenerfModelDesc, color_w), FLD(FenerfModelDesc, color_b), FLD(FenerfModelDesc, label_w),
    FLD(FenerfModelDesc, label_b), FLD(FenerfModelDesc, sigma_w), FLD(FenerfModelDesc, sigma_b), FLD(FenerfModelDesc, rgb_w),
    FLD(FenerfModelDesc, rgb_b), FLD(FenerfModelDesc, grid), FLD(FenerfModelDesc, precision), FLD(FenerfModelDesc, differentiable),
    FLD(FenerfModelDesc, wgrad_bf16_min_points)};
const FieldInfo kOptsFields[] = {
    FLD(FenerfCompositeOpts, clamp_mode), FLD(FenerfCompositeOpts, noise_std), FLD(FenerfCompositeOpts, last_back),
    FLD(FenerfCompositeOpts, white_back), FLD(FenerfCompositeOpts, black_back), FLD(FenerfCompositeOpts, fill_mode),
    FLD(FenerfCompositeOpts, fill_value), FLD(FenerfCompositeOpts, fill_enabled)};
const FieldInfo kRepackFields[] = {
    FLD(FenerfRepackMaps, stream_f32), FLD(FenerfRepackMaps, n_stream_f32), FLD(FenerfRepackMaps, stream_h16), FLD(FenerfRepackMaps, n_stream_h16),
    FLD(FenerfRepackMaps, consts), FLD(FenerfRepackMaps, n_consts), FLD(FenerfRepackMaps, consts_tail), FLD(FenerfRepackMaps, n_tail),
    FLD(FenerfRepackMaps, bwd_f32), FLD(FenerfRepackMaps, n_bwd_f32), FLD(FenerfRepackMaps, bwd_b16), FLD(FenerfRepackMaps, n_bwd_b16),
    FLD(FenerfRepackMaps, row_off), FLD(FenerfRepackMaps, row_len), FLD(FenerfRepackMaps, row_film), FLD(FenerfRepackMaps, n_rows),
    FLD(FenerfRepackMaps, scale_id)};
const FieldInfo kLocalFields[] = {
    FLD(FenerfLocalMapDesc, latent_dim), FLD(FenerfLocalMapDesc, map_hidden), FLD(FenerfLocalMapDesc, w0), FLD(FenerfLocalMapDesc, b0),
    FLD(FenerfLocalMapDesc, w1), FLD(FenerfLocalMapDesc, b1), FLD(FenerfLocalMapDesc, w2), FLD(FenerfLocalMapDesc, b2)};
const FieldInfo kGradFields[] = {
    FLD(FenerfSirenGrads, geo_w), FLD(FenerfSirenGrads, geo_b), FLD(FenerfSirenGrads, color_w), FLD(FenerfSirenGrads, color_b),
    FLD(FenerfSirenGrads, head_w), FLD(FenerfSirenGrads, head_b), FLD(FenerfSirenGrads, rgb_w), FLD(FenerfSirenGrads, rgb_b),
    FLD(FenerfSirenGrads, d_freq_geo), FLD(FenerfSirenGrads, d_phase_geo), FLD(FenerfSirenGrads, d_freq_app), FLD(FenerfSirenGrads, d_phase_app)};
const FieldInfo kMapFields[] = {FLD(FenerfMappingNet, n_layers), FLD(FenerfMappingNet, z_dim), FLD(FenerfMappingNet, hidden),
                                FLD(FenerfMappingNet, out_dim), FLD(FenerfMappingNet, W), FLD(FenerfMappingNet, b)};
#undef FLD
#define STRUCT(S, F) {#S, (long)sizeof(S), F, (int)(sizeof(F) / sizeof(F[0]))}
const StructInfo kStructs[] = {STRUCT(FenerfModelDesc, kDescFields), STRUCT(FenerfCompositeOpts, kOptsFields), STRUCT(FenerfRepackMaps, kRepackFields),
                               STRUCT(FenerfLocalMapDesc, kLocalFields), STRUCT(FenerfSirenGrads, kGradFields), STRUCT(FenerfMappingNet, kMapFields)};
#undef STRUCT
const StructInfo* find_struct(const char* name) {
  if (!name) return nullptr;
  for (const StructInfo& s : kStructs) if (!strcmp(s.name, name)) return &s;
  return nullptr;
}
}  // namespace
extern "C" long fenerf_struct_size(const char* struct_name) {
  const StructInfo* s = find_struct(struct_name);
  return s ? s->size : -1;
}
extern "C" long fenerf_struct_field_offset(const char* struct_name, const char* field) {
  const StructInfo* s = find_struct(struct_name);
  if (!s || !field) return -1;
  for (int i = 0; i < s->n; ++i) if (!strcmp(s->fields[i].name, field)) return s->fields[i].offset;
  return -1;
}
extern "C" const char* fenerf_struct_field_name(const char* struct_name, int index) {
  const StructInfo* s = find_struct(struct_name);
  return (s && index >= 0 && index < s->n) ? s->fields[index].name : nullptr;
}

extern "C" int fenerf_model_create(const FenerfModelDesc* d, FenerfModel** out) {
  if (!out) return fail(FENERF_E_INVALID, "out is NULL");
  *out = nullptr;
  std::string err;
  int rc = validate_desc(d, err);
  if (rc) return fail(rc, err);
  if ((rc = check_trig_domain())) return rc;        // fenerf_trig.h: the sine / cosine argument domain this build relies on
  FenerfModel* m = new (std::nothrow) FenerfModel();
  if (!m) return fail(FENERF_E_NOMEM, "out of host memory");
  // (value-initialised by `new FenerfModel()`: every scalar / pointer member is zero, the dump registry is an empty map)
  m->H = d->hidden_dim; m->n_geo = d->n_geo; m->n_color = d->n_color; m->C = d->output_dim;
  m->n_lab = d->output_dim - 4; m->L = d->n_geo + d->n_color;
  m->grid_ch = d->grid_ch; m->gd = d->grid_d; m->gh = d->grid_h; m->gw = d->grid_w;
  m->box_scale = d->box_scale;
  m->precision = d->precision;
  m->differentiable = d->differentiable != 0;
  m->wgrad_bf16_min_points = d->wgrad_bf16_min_points > 0 ? d->wgrad_bf16_min_points : 0;
  if (d->precision != FENERF_PREC_F32 && d->precision != FENERF_PREC_F16X3) { delete m; return fail(FENERF_E_INVALID, "unknown precision"); }
  m->bsh = bwd_stream_shape(m->H, m->n_geo, m->n_color, m->grid_ch != 0);
  m->sh = stream_shape(m->H, m->n_geo, m->n_color, m->grid_ch != 0);
  int dev = 0;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) { delete m; return hip_fail(e, "hipGetDeviceProperties"); }
  m->num_cus = prop.multiProcessorCount;
#ifdef FENERF_EXPERIMENT_SWITCHES
  // experiment switch (profiles/r04_gstep_overlap_why_not.md): size every persistent launch for fewer CUs than the device has.
  // Compiled in only by `make EXPERIMENTS=1` (round 6): the shipped library reads no tuning variable from the environment.
  if (const char* e = getenv("FENERF_EXP_NUM_CUS")) {
    const int n = atoi(e);
    if (n > 0 && n < m->num_cus) {
      static bool said = false;
      if (!said) { fprintf(stderr, "libfenerf_hip: FENERF_EXP_NUM_CUS=%d: persistent launches sized for %d of %d CUs (experiment switch)\n", n, n, m->num_cus); said = true; }
      m->num_cus = n;
    }
  }
#endif
  rc = upload_model(m, d, nullptr, true);
  if (rc) { fenerf_model_destroy(m); return rc; }
  *out = m;
  return FENERF_OK;
}

extern "C" int fenerf_model_set_forward_mode(FenerfModel* m, int mode) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (mode != FENERF_FORWARD_F16X3 && mode != FENERF_FORWARD_F16X2 && mode != FENERF_FORWARD_F16X3_COLOR_X2) return fail(FENERF_E_INVALID, "unknown forward mode");
  if (mode != FENERF_FORWARD_F16X3 && m->precision != FENERF_PREC_F16X3) return fail(FENERF_E_UNSUPPORTED, "reduced-precision forward modes: FENERF_PREC_F16X3 models only");
  const int prev = m->forward_mode;
  m->forward_mode = mode;
  return prev;
}

extern "C" int fenerf_model_update(FenerfModel* m, const FenerfModelDesc* d, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  std::string err;
  int rc = validate_desc(d, err);
  if (rc) return fail(rc, err);
  if (d->hidden_dim != m->H || d->n_geo != m->n_geo || d->n_color != m->n_color || d->output_dim != m->C ||
      d->grid_ch != m->grid_ch || d->precision != m->precision || (d->differentiable != 0) != (m->differentiable != 0) ||
      (d->wgrad_bf16_min_points > 0 ? d->wgrad_bf16_min_points : 0) != m->wgrad_bf16_min_points || (d->grid && (d->grid_d != m->gd || d->grid_h != m->gh || d->grid_w != m->gw)))
    return fail(FENERF_E_INVALID, "fenerf_model_update: architecture differs from the created model");
  m->box_scale = d->box_scale;
  return upload_model(m, d, (hipStream_t)stream, false);
}

extern "C" int fenerf_model_load_packed(FenerfModel* m, const float* stream_dev, size_t n_stream, const float* consts_dev,
                                        size_t n_consts, const float* bwd_dev, size_t n_bwd, const float* grid_dev, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  size_t want_s = (size_t)(m->sh.l0_entries + m->sh.ring_entries) * 256, want_c = (size_t)CONST_FILM_BIAS + (size_t)m->L * m->H;
  if (m->precision == FENERF_PREC_F16X3) {
    const StreamShape16 s16 = stream_shape16(m->H, m->n_geo, m->n_color, m->grid_ch != 0);
    want_s = (size_t)s16.l0_entries * 256 + (size_t)s16.ring_entries * 256;   // an f16 entry is 64 x 8 halves = 256 floats' worth
    want_c = (size_t)CONST_FILM_BIAS + (size_t)2 * m->L * m->H + 36;
  }
  size_t want_b = (size_t)(m->bsh.ht_entries + m->bsh.ring_entries) * 256;
  if (m->precision == FENERF_PREC_F16X3) {
    const BwdShape16 b16 = bwd_stream_shape16(m->H, m->n_geo, m->n_color, m->grid_ch != 0);
    want_b = (size_t)(b16.ht_entries + b16.ring_entries) * 256;
  }
  if (!stream_dev || !consts_dev || n_stream != want_s || n_consts != want_c) return fail(FENERF_E_INVALID, "packed stream / consts size mismatch");
  if (m->differentiable && (!bwd_dev || n_bwd != want_b)) return fail(FENERF_E_INVALID, "backward stream size mismatch");
  hipStream_t st = (hipStream_t)stream;
  m->stream32_valid = 0;
  HIP_TRY(hipMemcpyAsync(m->d_stream, stream_dev, n_stream * sizeof(float), hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(m->d_consts, consts_dev, n_consts * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (m->differentiable) HIP_TRY(hipMemcpyAsync(m->d_bwd_stream, bwd_dev, n_bwd * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (grid_dev) {
    if (!m->grid_ch) return fail(FENERF_E_INVALID, "model has no feature grid");
    { PhaseScope ph(PH_GRID, stream); return launch_grid_relayout(grid_dev, m->d_grid, m->grid_ch, m->gd, m->gh, m->gw, stream); }
  }
  return FENERF_OK;
}

extern "C" int fenerf_model_repack(FenerfModel* m, const float* flat_dev, size_t n_flat, const FenerfRepackMaps* r, const float* grid_dev,
                                   void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (!flat_dev || !r || n_flat == 0) return fail(FENERF_E_INVALID, "NULL pointer");
  const bool f16 = m->precision == FENERF_PREC_F16X3;
  const size_t tail = f16 ? (size_t)m->L * m->H + 36 : 0;
  if (!r->stream_f32 || !r->consts || r->n_stream_f32 + r->n_stream_h16 / 2 != m->n_stream || (r->n_stream_h16 & 1) ||
      r->n_consts + r->n_tail != m->n_consts || r->n_tail != tail || (f16 != (r->stream_h16 != nullptr)) || (f16 && !r->consts_tail))
    return fail(FENERF_E_INVALID, "repack maps do not match the model's forward stream / consts");
  if (m->differentiable &&
      (!r->bwd_f32 || r->n_bwd_f32 + r->n_bwd_b16 / 2 != m->n_bwd || (r->n_bwd_b16 & 1) || (f16 != (r->bwd_b16 != nullptr))))
    return fail(FENERF_E_INVALID, "repack maps do not match the model's backward stream");
  const size_t cap = (size_t)m->L * m->H + 64;
  if (f16) {
    if (!r->row_off || !r->row_len || !r->row_film || !r->scale_id || r->n_rows < 0 || (size_t)r->n_rows + 1 > cap)
      return fail(FENERF_E_INVALID, "repack maps: row table missing or too long");
    if (!m->d_row_scale) HIP_TRY(hipMalloc((void**)&m->d_row_scale, 2 * cap * sizeof(float)));
  }
  m->stream32_valid = 0;
  int rc = [&] { PhaseScope ph(PH_REPACK, stream); return launch_repack(m, flat_dev, r, m->d_row_scale, m->d_row_scale ? m->d_row_scale + cap : nullptr, stream); }();
  if (rc) return rc;
  if (grid_dev) {
    if (!m->grid_ch) return fail(FENERF_E_INVALID, "model has no feature grid");
    { PhaseScope ph(PH_GRID, stream); return launch_grid_relayout(grid_dev, m->d_grid, m->grid_ch, m->gd, m->gh, m->gw, stream); }
  }
  return FENERF_OK;
}

extern "C" int fenerf_model_export_packed(const FenerfModel* m, float* stream_dev, size_t n_stream, float* consts_dev, size_t n_consts,
                                          float* bwd_dev, size_t n_bwd, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  hipStream_t st = (hipStream_t)stream;
  if ((stream_dev && n_stream != m->n_stream) || (consts_dev && n_consts != m->n_consts) || (bwd_dev && (!m->differentiable || n_bwd != m->n_bwd)))
    return fail(FENERF_E_INVALID, "export: buffer size mismatch");
  if (stream_dev) HIP_TRY(hipMemcpyAsync(stream_dev, m->d_stream, n_stream * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (consts_dev) HIP_TRY(hipMemcpyAsync(consts_dev, m->d_consts, n_consts * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (bwd_dev) HIP_TRY(hipMemcpyAsync(bwd_dev, m->d_bwd_stream, n_bwd * sizeof(float), hipMemcpyDeviceToDevice, st));
  return FENERF_OK;
}

extern "C" void fenerf_model_destroy(FenerfModel* m) {
  if (!m) return;
  if (m->d_row_scale) (void)hipFree(m->d_row_scale);
  if (m->d_stream32) (void)hipFree(m->d_stream32);
  if (m->d_consts32) (void)hipFree(m->d_consts32);
  if (m->d_stream) (void)hipFree(m->d_stream);
  if (m->d_consts) (void)hipFree(m->d_consts);
  if (m->d_grid) (void)hipFree(m->d_grid);
  if (m->d_bwd_stream) (void)hipFree(m->d_bwd_stream);
  delete m;
}

extern "C" size_t fenerf_film_workspace_bytes(const FenerfModel* m, int B) {
  if (!m || B <= 0) return 0;
  // + 1 KiB: the shared-stream kernel fetches FiLM parameters with whole-KiB LDS-DMA transfers
  return align_up((size_t)2 * B * m->L * m->H * sizeof(float) + 1024, 256);
}

static int film_prep(const FenerfModel* m, long long B, const float* fg, const float* pg, const float* fa, const float* pa,
                     void* film_ws, const float** fp, const float** pp, void* stream) {
  if (!fg || !pg) return fail(FENERF_E_INVALID, "freq_geo / phase_geo is NULL");
  if (!fa || !pa) return fail(FENERF_E_INVALID, "freq_app / phase_app is NULL");
  if (!film_ws) return fail(FENERF_E_INVALID, "film workspace is NULL");
  float* f = (float*)film_ws;
  float* p = f + (size_t)B * m->L * m->H;
  *fp = f; *pp = p;
  PhaseScope ph(PH_FILM_PREP, stream);
  return launch_film_prep(m, B, fg, pg, fa, pa, f, p, stream);
}

static int run_siren(const FenerfModel* m, const SirenParams& sp, void* stream) {
  PhaseScope ph(sp.tape ? PH_SIREN_SAVE : PH_SIREN, stream);
  return launch_siren(m, sp, stream);
}
static int run_composite(const CompositeParams& p, bool merge, void* stream) {
  PhaseScope ph(PH_COMPOSITE, stream);
  return launch_composite(p, merge, stream);
}

static void fill_common(const FenerfModel* m, SirenParams& sp, const float* fp, const float* pp) {
  memset(&sp, 0, sizeof(sp));
  sp.stream = m->d_stream;
  sp.consts = m->d_consts;
  sp.fp = fp; sp.pp = pp;
  sp.grid = m->d_grid; sp.gd = m->gd; sp.gh = m->gh; sp.gw = m->gw;
  sp.box_scale = m->box_scale;
  sp.ring_offset_floats = (long long)m->sh.l0_entries * 256;
}

extern "C" int fenerf_siren_forward(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                                    const float* freq_geo, const float* phase_geo, const float* freq_app,
                                    const float* phase_app, float* out, void* film_ws, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (B <= 0 || P < 0) return fail(FENERF_E_INVALID, "B must be > 0 and P >= 0");
  if (P == 0) return FENERF_OK;
  if (!points || !out) return fail(FENERF_E_INVALID, "points / out is NULL");
  const float *fp, *pp;
  int rc = film_prep(m, B, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream);
  if (rc) return rc;
  SirenParams sp;
  fill_common(m, sp, fp, pp);
  sp.points = points; sp.pdirs = ray_dirs;
  sp.P = (long long)B * P; sp.pts_per_image = P; sp.n_per_ray = 1;
  sp.out = out;
  return run_siren(m, sp, stream);
}

extern "C" size_t fenerf_film_workspace_bytes_pointwise(const FenerfModel* m, int B, int64_t P) {
  if (!m || B <= 0 || P <= 0) return 0;
  return align_up((size_t)2 * (size_t)B * (size_t)P * m->L * m->H * sizeof(float) + 1024, 256);
}

extern "C" int fenerf_siren_forward_pointwise(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                                              const float* freq_geo, const float* phase_geo, const float* freq_app,
                                              const float* phase_app, float* out, void* film_ws, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  const bool f16 = m->precision == FENERF_PREC_F16X3;
  if (f16 && !m->stream32_valid)
    return fail(FENERF_E_UNSUPPORTED, "per-point FiLM parameters run on the exact-fp32 stream, which a device-side re-pack "
                                      "(fenerf_model_repack / fenerf_model_load_packed) does not refresh: fenerf_model_update first");
  if (B <= 0 || P < 0) return fail(FENERF_E_INVALID, "B must be > 0 and P >= 0");
  if (P == 0) return FENERF_OK;
  if (!points || !out) return fail(FENERF_E_INVALID, "points / out is NULL");
  if (!freq_geo || !phase_geo || !freq_app || !phase_app || !film_ws) return fail(FENERF_E_INVALID, "film parameter / workspace pointer is NULL");
  // every lane reads its own point's FiLM block: the exact-fp32 kernel, for FENERF_PREC_F16X3 models too (their fp32-class promise holds)
  float* fp = (float*)film_ws;
  float* pp = fp + (size_t)B * (size_t)P * m->L * m->H;
  int rc = [&] { PhaseScope ph(PH_FILM_PREP, stream); return launch_film_prep(m, (long long)B * P, freq_geo, phase_geo, freq_app, phase_app, fp, pp, stream, true); }();
  if (rc) return rc;
  SirenParams sp;
  fill_common(m, sp, fp, pp);
  if (f16) { sp.stream = m->d_stream32; sp.consts = m->d_consts32; }
  sp.points = points; sp.pdirs = ray_dirs;
  sp.P = (long long)B * P; sp.pts_per_image = P; sp.n_per_ray = 1;
  sp.out = out;
  sp.film_per_point = 1;
  PhaseScope ph(PH_SIREN, stream);
  return launch_siren_f32(m, sp, stream);
}

extern "C" int fenerf_siren_forward_rays(const FenerfModel* m, int B, int R, int N, const float* origins,
                                         const float* dirs, const float* z, int lock_view, const float* freq_geo,
                                         const float* phase_geo, const float* freq_app, const float* phase_app,
                                         float* out, void* film_ws, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (B <= 0 || R < 0 || N <= 0) return fail(FENERF_E_INVALID, "B, N must be > 0 and R >= 0");
  if (R == 0) return FENERF_OK;
  if (!origins || !dirs || !z || !out) return fail(FENERF_E_INVALID, "origins / dirs / z / out is NULL");
  const float *fp, *pp;
  int rc = film_prep(m, B, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream);
  if (rc) return rc;
  SirenParams sp;
  fill_common(m, sp, fp, pp);
  sp.origins = origins; sp.dirs = dirs; sp.z = z; sp.n_per_ray = N; sp.lock_view = lock_view;
  sp.P = (long long)B * R * N; sp.pts_per_image = (long long)R * N;
  sp.out = out;
  return run_siren(m, sp, stream);
}

extern "C" int fenerf_siren_time_rays(const FenerfModel* m, int B, int R, int N, const float* origins, const float* dirs,
                                      const float* z, const float* freq_geo, const float* phase_geo, const float* freq_app,
                                      const float* phase_app, float* out, void* film_ws, int iters, float* avg_ms,
                                      void* stream) {
  if (!m || !avg_ms || iters < 1) return fail(FENERF_E_INVALID, "model / avg_ms is NULL or iters < 1");
  if (B <= 0 || R <= 0 || N <= 0 || !origins || !dirs || !z || !out) return fail(FENERF_E_INVALID, "bad shape or NULL pointer");
  const float *fp, *pp;
  int rc = film_prep(m, B, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream);
  if (rc) return rc;
  SirenParams sp;
  fill_common(m, sp, fp, pp);
  sp.origins = origins; sp.dirs = dirs; sp.z = z; sp.n_per_ray = N;
  sp.P = (long long)B * R * N; sp.pts_per_image = (long long)R * N;
  sp.out = out;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  rc = launch_siren(m, sp, stream);  // warm-up (also sets the LDS attribute)
  if (rc == FENERF_OK) {
    hipError_t e = hipEventRecord(e0, (hipStream_t)stream);
    for (int i = 0; i < iters && rc == FENERF_OK; ++i) rc = launch_siren(m, sp, stream);
    if (e == hipSuccess) e = hipEventRecord(e1, (hipStream_t)stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e != hipSuccess && rc == FENERF_OK) rc = hip_fail(e, "event timing");
    *avg_ms = ms / (float)iters;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

extern "C" int fenerf_siren_clock_probe(const FenerfModel* m, int B, int R, int N, const float* origins, const float* dirs,
                                        const float* z, const float* freq_geo, const float* phase_geo, const float* freq_app,
                                        const float* phase_app, float* out, void* film_ws, int iters, double* result4, void* stream) {
  if (!m || !result4 || iters < 1) return fail(FENERF_E_INVALID, "model / result is NULL or iters < 1");
  if (B <= 0 || R <= 0 || N <= 0 || !origins || !dirs || !z || !out) return fail(FENERF_E_INVALID, "bad shape or NULL pointer");
  const float *fp, *pp;
  int rc = film_prep(m, B, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream);
  if (rc) return rc;
  SirenParams sp;
  fill_common(m, sp, fp, pp);
  sp.origins = origins; sp.dirs = dirs; sp.z = z; sp.n_per_ray = N;
  sp.P = (long long)B * R * N; sp.pts_per_image = (long long)R * N;
  sp.out = out;
  if (sp.pts_per_image % 32 != 0 && B > 1) return fail(FENERF_E_INVALID, "clock probe: one launch only (R * N a multiple of 32, or B = 1)");
  int dev = 0, khz = 0;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
  const size_t per = (size_t)m->num_cus * 4;            // a launch has at most one workgroup per CU
  unsigned long long* d_clk = nullptr;
  HIP_TRY(hipMalloc((void**)&d_clk, per * iters * sizeof(unsigned long long)));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMemsetAsync(d_clk, 0, per * iters * sizeof(unsigned long long), (hipStream_t)stream);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  std::vector<unsigned long long> h(per * iters);
  float ms = 0.f;
  if (e == hipSuccess) {
    rc = launch_siren(m, sp, stream);   // warm-up (also sets the LDS attribute)
    if (rc == FENERF_OK) e = hipEventRecord(e0, (hipStream_t)stream);
    for (int i = 0; i < iters && rc == FENERF_OK; ++i) {
      sp.clk = d_clk + per * i;
      rc = launch_siren(m, sp, stream);
    }
    if (e == hipSuccess) e = hipEventRecord(e1, (hipStream_t)stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess) e = hipMemcpy(h.data(), d_clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(d_clk);
  if (rc) return rc;
  if (e != hipSuccess) return hip_fail(e, "clock probe");
  double cyc_launch = 0.0, cyc_all = 0.0, ticks_all = 0.0;
  for (int i = 0; i < iters; ++i) {
    double longest = 0.0;
    for (int b = 0; b < m->num_cus; ++b) {
      const unsigned long long* c = h.data() + per * i + (size_t)b * 4;
      if (c[2] <= c[0] || c[3] <= c[1]) continue;      // workgroup slot not used by this launch
      const double cyc = (double)(c[2] - c[0]), tk = (double)(c[3] - c[1]);
      if (cyc > longest) longest = cyc;
      cyc_all += cyc; ticks_all += tk;
    }
    cyc_launch += longest;
  }
  result4[0] = ms / (double)iters;
  result4[1] = cyc_launch / (double)iters;
  result4[2] = ticks_all > 0 ? cyc_all / ticks_all * (double)khz * 1e-6 : 0.0;   // cycles per tick x ticks per second -> GHz
  result4[3] = (double)khz;
  return FENERF_OK;
}

extern "C" double fenerf_siren_executed_flop_per_point(const FenerfModel* m) {
  if (!m) return 0.0;
  const int H = m->H, NB = H / 32, G = m->grid_ch ? 1 : 0;
  const int n_sq = (m->n_geo - 1) + (m->n_color - 1);                    // H -> H FiLM layers
  if (m->precision == FENERF_PREC_F16X3) {
    // siren16w_kernel, per 16-point tile: v_mfma_f32_16x16x32_f16 (16,384 FLOP), 6 per k32-step and 32-row n-block (two row tiles x
    // [wl xh, wh xl, wh xh]); colour layer 0 has one k32-step of grid features and one of view direction more; the folded head and
    // the rgb head are one n-block each; layer 0 (K = 3) runs on v_mfma_f32_16x16x4_f32 (2,048 FLOP), 2 per n-block
    const int KS = H / 32;
    // (fenerf_model_set_forward_mode: 4 instead of 6 where the weight lo halves are not multiplied -- everywhere, or in the colour layers and the rgb head)
    const double geo = (double)(m->n_geo - 1) * NB * KS + KS /* label / sigma head */, col = (double)(m->n_color - 1) * NB * KS + (double)NB * (KS + G + 1) + KS /* rgb head */;
    const double per_geo = m->forward_mode == FENERF_FORWARD_F16X2 ? 4.0 : 6.0, per_col = m->forward_mode == FENERF_FORWARD_F16X3 ? 6.0 : 4.0;
    const double mf16 = per_geo * geo + per_col * col;
    return (mf16 * 16384.0 + 2.0 * NB * 2048.0) / 16.0;
  }
  // siren_kernel, per 32-point tile: v_mfma_f32_32x32x2_f32 (4,096 FLOP), ceil(K / 2) per 32-row n-block
  const int K0 = H + 3 + m->grid_ch;
  const double mf32 = (double)n_sq * NB * (H / 2) + (double)NB * ((K0 + 1) / 2) + (double)NB * 2 + 2.0 * (H / 2);
  return mf32 * 4096.0 / 32.0;
}

extern "C" int fenerf_ray_setup(int B, int img_size, int N, float z_cam, float ray_start, float ray_end, const float* u_jitter,
                                const float* theta, const float* phi, float* origins, float* dirs, float* z, float* pitch,
                                float* yaw, void* stream) {
  if (B < 0 || img_size < 0 || N < 1) return fail(FENERF_E_INVALID, "need B, img_size >= 0 and N >= 1");
  if (B == 0 || img_size == 0) return FENERF_OK;
  if (!u_jitter || !theta || !phi || !origins || !dirs || !z || !pitch || !yaw) return fail(FENERF_E_INVALID, "NULL pointer");
  { PhaseScope ph(PH_RAY_SETUP, stream); return launch_ray_setup(B, img_size, N, z_cam, ray_start, ray_end, u_jitter, theta, phi, origins, dirs, z, pitch, yaw, stream); }
}

extern "C" int fenerf_composite(int64_t BR, int M, int C, const float* rgb_sigma, const float* z, const float* noise,
                                const FenerfCompositeOpts* opts, float* out_rgb, float* out_depth, float* out_weights,
                                float* out_wsum, void* stream) {
  int rc = check_opts(opts);
  if (rc) return rc;
  if (BR < 0 || M < 1 || M > FENERF_MAX_RAY_SAMPLES || C < 2) return fail(FENERF_E_INVALID, "need BR >= 0, 1 <= M <= 1024 (FENERF_MAX_RAY_SAMPLES), C >= 2");
  if (BR == 0) return FENERF_OK;
  if (!rgb_sigma || !z) return fail(FENERF_E_INVALID, "rgb_sigma / z is NULL");
  if (opts->fill_mode == FENERF_FILL_EVAL_WHITE_BACK && C != 4) return fail(FENERF_E_INVALID, "eval_white_back needs a 3-channel model");
  CompositeParams p;
  memset(&p, 0, sizeof(p));
  p.BR = BR; p.M = M; p.C = C; p.N = M;
  p.rows_a = rgb_sigma; p.z_a = z; p.noise = noise; p.o = *opts;
  p.out_rgb = out_rgb; p.out_depth = out_depth; p.out_weights = out_weights; p.out_wsum = out_wsum;
  p.out_ch = out_channels(C, opts);
  p.sigma_only = out_rgb ? 0 : 1;
  return run_composite(p, false, stream);
}

extern "C" int fenerf_resample(int64_t BR, int N, const float* z_coarse, const float* coarse_weights, const float* u,
                               float* z_fine, void* stream) {
  if (BR < 0 || N < 3 || 2 * N > FENERF_MAX_RAY_SAMPLES) return fail(FENERF_E_INVALID, "need BR >= 0 and 3 <= N <= 512 (FENERF_MAX_RAY_SAMPLES / 2)");
  if (BR == 0) return FENERF_OK;
  if (!z_coarse || !coarse_weights || !u || !z_fine) return fail(FENERF_E_INVALID, "NULL pointer");
  { PhaseScope ph(PH_RESAMPLE, stream); return launch_resample(BR, N, z_coarse, coarse_weights, u, z_fine, stream); }
}

extern "C" int fenerf_sample_pdf(int64_t BR, int K, int n_importance, const float* bins, const float* weights,
                                 const float* u, float* samples, void* stream) {
  if (BR < 0 || K < 1 || 2 * (K + 1) > FENERF_MAX_RAY_SAMPLES || n_importance < 1 || 2 * n_importance > FENERF_MAX_RAY_SAMPLES)
    return fail(FENERF_E_INVALID, "need BR >= 0, 1 <= K <= 511, 1 <= n_importance <= 512 (FENERF_MAX_RAY_SAMPLES / 2)");
  if (BR == 0) return FENERF_OK;
  if (!bins || !weights || !u || !samples) return fail(FENERF_E_INVALID, "NULL pointer");
  { PhaseScope ph(PH_RESAMPLE, stream); return launch_sample_pdf(BR, K, n_importance, bins, weights, u, samples, stream); }
}

extern "C" int fenerf_merge_composite(int64_t BR, int N, int C, const float* fine, const float* coarse,
                                      const float* z_fine, const float* z_coarse, const float* noise,
                                      const FenerfCompositeOpts* opts, float* out_rgb, float* out_depth,
                                      float* out_weights, float* out_wsum, float* out_z_sorted, void* stream) {
  int rc = check_opts(opts);
  if (rc) return rc;
  if (BR < 0 || N < 1 || 2 * N > FENERF_MAX_RAY_SAMPLES || C < 2) return fail(FENERF_E_INVALID, "need BR >= 0, 1 <= N <= 512 (FENERF_MAX_RAY_SAMPLES / 2), C >= 2");
  if (BR == 0) return FENERF_OK;
  if (!fine || !coarse || !z_fine || !z_coarse) return fail(FENERF_E_INVALID, "NULL pointer");
  if (opts->fill_mode == FENERF_FILL_EVAL_WHITE_BACK && C != 4) return fail(FENERF_E_INVALID, "eval_white_back needs a 3-channel model");
  CompositeParams p;
  memset(&p, 0, sizeof(p));
  p.BR = BR; p.M = 2 * N; p.C = C; p.N = N;
  p.rows_a = fine; p.rows_b = coarse; p.z_a = z_fine; p.z_b = z_coarse; p.noise = noise; p.o = *opts;
  p.out_rgb = out_rgb; p.out_depth = out_depth; p.out_weights = out_weights; p.out_wsum = out_wsum; p.out_z = out_z_sorted;
  p.out_ch = out_channels(C, opts);
  p.sigma_only = out_rgb ? 0 : 1;
  return run_composite(p, true, stream);
}

extern "C" size_t fenerf_siren_tape_floats(const FenerfModel* m, int64_t total_points) {
  if (!m || total_points <= 0) return 0;
  // whole 32-point tiles, rounded up to the 4 tiles a workgroup of the shared-stream kernel walks together: the phantom
  // tiles that pad its last quad dump their registers too (a guard would put a branch into the MFMA stream)
  const size_t tiles = ((size_t)total_points + 31) / 32, quads = (tiles + 3) / 4;
  return (size_t)m->L * m->H * 32 * quads * 4;
}

extern "C" size_t fenerf_siren_tape_bytes(const FenerfModel* m, int64_t total_points, int tape_format) {
  const size_t f = fenerf_siren_tape_floats(m, total_points);
  return tape_format == FENERF_TAPE_U16 ? f * 2 : f * 4;      // same tiles, same slack, 2 instead of 4 bytes per (point, feature)
}

static int check_tape_format(const FenerfModel* m, int tape_format) {
  if (tape_format != FENERF_TAPE_F32 && tape_format != FENERF_TAPE_U16 && tape_format != FENERF_TAPE_F32_W) return fail(FENERF_E_INVALID, "unknown tape format");
  if (tape_format != FENERF_TAPE_F32 && m->precision != FENERF_PREC_F16X3)
    return fail(FENERF_E_UNSUPPORTED, "FENERF_TAPE_U16 / FENERF_TAPE_F32_W: FENERF_PREC_F16X3 models only (the exact-fp32 kernels keep the fp32 tape)");
  return FENERF_OK;
}

extern "C" size_t fenerf_siren_dtheta_floats(const FenerfModel* m, int64_t total_points) {
  if (!m || total_points <= 0) return 0;
  const long long tiles = (total_points + 31) / 32;
  // d theta dump + the chain kernel's FiLM sums, one [L][2][H] block per 16-point tile (the 32-point kernels use every other one's worth)
  return (size_t)m->L * m->H * (size_t)tiles * 32 + (size_t)film_tile_floats(2 * tiles, m->L, m->H);
}

extern "C" int fenerf_siren_backward_stream_bytes(const FenerfModel* m, int64_t chunk_points, double* out4) {
  if (!m || !out4 || chunk_points <= 0) return fail(FENERF_E_INVALID, "model / out is NULL or chunk_points <= 0");
  if (use_bf16_dump(m, chunk_points)) {
    // bf16 dump (fenerf_layout.h): d(theta) and x = sin(2 pi theta) 2 B each, written by the chain and read once by the square job
    // (the last layer's x is not written: 1 / L less); the thin jobs read two bf16 d(theta) layers and two fp32 tape layers
    out4[0] = 2.0 + 2.0 * (m->L - 1) / m->L; out4[1] = 2.0 + 2.0; out4[2] = 2 * 2.0 + 2 * 4.0; out4[3] = 4.0;
  } else {
    // fp32 dump: d(theta) 4 B written and read once; the weight gradients recompute their input activations from the fp32 tape
    out4[0] = 4.0; out4[1] = 4.0 + 4.0; out4[2] = 2 * 4.0 + 2 * 4.0; out4[3] = 4.0;
  }
  return FENERF_OK;
}

extern "C" int fenerf_siren_backward_stream_bytes_fmt(const FenerfModel* m, int64_t chunk_points, int tape_format, double* out4) {
  int rc = fenerf_siren_backward_stream_bytes(m, chunk_points, out4);
  if (rc) return rc;
  if (tape_format == FENERF_TAPE_U16) {     // every tape read shrinks from 4 to 2 bytes: the square job's, the thin jobs' two layers, the tape itself
    out4[1] -= 2.0; out4[2] -= 2 * 2.0; out4[3] = 2.0;
  }
  return FENERF_OK;
}

extern "C" int fenerf_siren_forward_save(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                                         const float* freq_geo, const float* phase_geo, const float* freq_app,
                                         const float* phase_app, float* out, float* tape, float* tape_e, void* film_ws,
                                         void* stream) {
  return fenerf_siren_forward_save_fmt(m, B, P, points, ray_dirs, freq_geo, phase_geo, freq_app, phase_app, out, tape, tape_e, film_ws,
                                       FENERF_TAPE_F32, stream);
}

extern "C" int fenerf_siren_forward_save_fmt(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                                             const float* freq_geo, const float* phase_geo, const float* freq_app,
                                             const float* phase_app, float* out, void* tape, float* tape_e, void* film_ws,
                                             int tape_format, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (int rcf = check_tape_format(m, tape_format)) return rcf;
  if (!m->differentiable) return fail(FENERF_E_UNSUPPORTED, "model was not created with differentiable != 0");
  if (B <= 0 || P < 0) return fail(FENERF_E_INVALID, "B must be > 0 and P >= 0");
  if (P % 32) return fail(FENERF_E_INVALID, "differentiable path: points per image must be a multiple of 32");
  if (P == 0) return FENERF_OK;
  if (!points || !out || !tape || (m->grid_ch && !tape_e)) return fail(FENERF_E_INVALID, "points / out / tape is NULL");
  const float *fp, *pp;
  int rc = film_prep(m, B, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream);
  if (rc) return rc;
  SirenParams sp;
  fill_common(m, sp, fp, pp);
  sp.points = points; sp.pdirs = ray_dirs;
  sp.P = (long long)B * P; sp.pts_per_image = P; sp.n_per_ray = 1;
  sp.out = out; sp.tape = (float*)tape; sp.tape_e = tape_e; sp.tape_format = tape_format;
  return run_siren(m, sp, stream);
}

extern "C" int fenerf_siren_backward(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                                     const float* freq_app, const float* phase_app, const float* out, const float* d_out,
                                     const float* tape, float* d_t, float* d_e, void* film_ws, void* stream) {
  return fenerf_siren_backward_fmt(m, B, P, freq_geo, phase_geo, freq_app, phase_app, out, d_out, tape, FENERF_TAPE_F32, d_t, d_e, film_ws, stream);
}

extern "C" int fenerf_siren_backward_fmt(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                                         const float* freq_app, const float* phase_app, const float* out, const float* d_out,
                                         const void* tape, int tape_format, float* d_t, float* d_e, void* film_ws, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (int rcf = check_tape_format(m, tape_format)) return rcf;
  if (!m->differentiable || !m->d_bwd_stream) return fail(FENERF_E_UNSUPPORTED, "model was not created with differentiable != 0");
  if (B <= 0 || P < 0) return fail(FENERF_E_INVALID, "B must be > 0 and P >= 0");
  if (P % 32) return fail(FENERF_E_INVALID, "differentiable path: points per image must be a multiple of 32");
  if (P == 0) return FENERF_OK;
  if (!out || !d_out || !tape || !d_t || (m->grid_ch && !d_e)) return fail(FENERF_E_INVALID, "NULL pointer");
  const float *fp, *pp;
  int rc = film_prep(m, B, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream);
  if (rc) return rc;
  SirenBwdParams bp;
  memset(&bp, 0, sizeof(bp));
  bp.stream = m->d_bwd_stream;
  bp.ring_offset_floats = (long long)m->bsh.ht_entries * 256;
  bp.fp = fp; bp.pp = pp;
  bp.P = (long long)B * P; bp.pts_per_image = P;
  bp.out = out; bp.d_out = d_out; bp.tape = (const float*)tape; bp.tape_format = tape_format; bp.d_t = d_t; bp.d_e = d_e;
  bp.film_tiles = d_t + (size_t)m->L * m->H * (size_t)B * (size_t)P;    // appended to the dtheta dump
  bp.bf16_dump = use_bf16_dump(m, (long long)B * P);
  note_dump(m, d_t, (long long)B * P);
  PhaseScope ph(PH_CHAIN, stream);
  return m->precision == FENERF_PREC_F16X3 ? launch_siren_backward16w(m, bp, stream) : launch_siren_backward(m, bp, stream);
}

extern "C" size_t fenerf_siren_film_sums_floats(const FenerfModel* m, int B, int64_t P) {
  if (!m || B <= 0 || P <= 0) return 0;
  // one [L][2][H] block per unit of the chain kernel's FiLM sums: its workgroup's 128 points where an oct cannot straddle images, else a wave's 16
  const long long total = (long long)B * P, unit = bwd16w_film_unit(total, P);
  return (size_t)film_tile_floats((total + unit - 1) / unit, m->L, m->H);
}

// Inversion (inverse_render_double_semantic.py:324-410 optimises only the FiLM offsets): the chain without its d(theta) dump and without
// d(grid features) -- the per-tile FiLM sums are all fenerf_siren_film_grads needs.
extern "C" int fenerf_siren_backward_film(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                                          const float* freq_app, const float* phase_app, const float* out, const float* d_out,
                                          const float* tape, float* film_sums, void* film_ws, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (!m->differentiable || !m->d_bwd_stream) return fail(FENERF_E_UNSUPPORTED, "model was not created with differentiable != 0");
  if (m->precision != FENERF_PREC_F16X3)
    return fail(FENERF_E_UNSUPPORTED, "fenerf_siren_backward_film: FENERF_PREC_F16X3 models only (the exact-fp32 jobs take their FiLM sums from the dump)");
  if (B <= 0 || P < 0) return fail(FENERF_E_INVALID, "B must be > 0 and P >= 0");
  if (P % 32) return fail(FENERF_E_INVALID, "differentiable path: points per image must be a multiple of 32");
  if (P == 0) return FENERF_OK;
  if (!out || !d_out || !tape || !film_sums) return fail(FENERF_E_INVALID, "NULL pointer");
  const float *fp, *pp;
  int rc = film_prep(m, B, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream);
  if (rc) return rc;
  SirenBwdParams bp;
  memset(&bp, 0, sizeof(bp));
  bp.stream = m->d_bwd_stream;
  bp.ring_offset_floats = (long long)m->bsh.ht_entries * 256;
  bp.fp = fp; bp.pp = pp;
  bp.P = (long long)B * P; bp.pts_per_image = P;
  bp.out = out; bp.d_out = d_out; bp.tape = tape;
  bp.d_t = nullptr; bp.d_e = nullptr;          // no dump, no d(grid features)
  bp.film_tiles = film_sums;
  PhaseScope ph(PH_CHAIN, stream);
  return launch_siren_backward16w(m, bp, stream);
}

extern "C" int fenerf_siren_film_grads(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                                       const float* freq_app, const float* phase_app, const float* film_sums,
                                       const FenerfSirenGrads* g, void* workspace, void* film_ws, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (!m->differentiable) return fail(FENERF_E_UNSUPPORTED, "model was not created with differentiable != 0");
  if (m->precision != FENERF_PREC_F16X3) return fail(FENERF_E_UNSUPPORTED, "fenerf_siren_film_grads: FENERF_PREC_F16X3 models only");
  if (B <= 0 || P < 0) return fail(FENERF_E_INVALID, "B must be > 0 and P >= 0");
  if (P % 32) return fail(FENERF_E_INVALID, "differentiable path: points per image must be a multiple of 32");
  if (P == 0) return FENERF_OK;
  if (!film_sums || !g || !workspace) return fail(FENERF_E_INVALID, "NULL pointer");
  if (!g->d_freq_geo || !g->d_phase_geo || !g->d_freq_app || !g->d_phase_app) return fail(FENERF_E_INVALID, "grads: film pointer is NULL");
  const float *fp, *pp;
  int rc = film_prep(m, B, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream);
  if (rc) return rc;
  return launch_param_grads(m, B, P, nullptr, nullptr, fp, pp, nullptr, nullptr, nullptr, nullptr, nullptr, *g, true, workspace, stream, film_sums);
}

extern "C" int fenerf_siren_backward_fuses_grid(const FenerfModel* m) {
  return m && m->differentiable && m->grid_ch && m->precision == FENERF_PREC_F16X3;
}

extern "C" int fenerf_siren_backward_grid(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                                          const float* freq_app, const float* phase_app, const float* out, const float* d_out,
                                          const float* tape, const float* points, float* d_t, float* d_grid_cl, float* scratch_d_e,
                                          void* film_ws, void* stream) {
  return fenerf_siren_backward_grid_fmt(m, B, P, freq_geo, phase_geo, freq_app, phase_app, out, d_out, tape, FENERF_TAPE_F32, points, d_t, d_grid_cl,
                                        scratch_d_e, film_ws, stream);
}

extern "C" int fenerf_siren_backward_grid_fmt(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                                              const float* freq_app, const float* phase_app, const float* out, const float* d_out,
                                              const void* tape, int tape_format, const float* points, float* d_t, float* d_grid_cl,
                                              float* scratch_d_e, void* film_ws, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (!m->grid_ch) return fail(FENERF_E_UNSUPPORTED, "model has no feature grid");
  if (int rcf = check_tape_format(m, tape_format)) return rcf;
  if (!points || !d_grid_cl) return fail(FENERF_E_INVALID, "points / d_grid_cl is NULL");
  if (!fenerf_siren_backward_fuses_grid(m)) {
    if (!scratch_d_e) return fail(FENERF_E_INVALID, "this model's chain kernel does not scatter in place: scratch_d_e is required");
    int rc = fenerf_siren_backward_fmt(m, B, P, freq_geo, phase_geo, freq_app, phase_app, out, d_out, tape, tape_format, d_t, scratch_d_e, film_ws, stream);
    if (rc || P == 0) return rc;
    { PhaseScope ph(PH_GRID, stream); return launch_grid_backward(m, (long long)B * P, points, scratch_d_e, d_grid_cl, stream); }
  }
  if (!m->differentiable || !m->d_bwd_stream) return fail(FENERF_E_UNSUPPORTED, "model was not created with differentiable != 0");
  if (B <= 0 || P < 0) return fail(FENERF_E_INVALID, "B must be > 0 and P >= 0");
  if (P % 32) return fail(FENERF_E_INVALID, "differentiable path: points per image must be a multiple of 32");
  if (P == 0) return FENERF_OK;
  if (!out || !d_out || !tape || !d_t) return fail(FENERF_E_INVALID, "NULL pointer");
  const float *fp, *pp;
  int rc = film_prep(m, B, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream);
  if (rc) return rc;
  SirenBwdParams bp;
  memset(&bp, 0, sizeof(bp));
  bp.stream = m->d_bwd_stream;
  bp.ring_offset_floats = (long long)m->bsh.ht_entries * 256;
  bp.fp = fp; bp.pp = pp;
  bp.P = (long long)B * P; bp.pts_per_image = P;
  bp.out = out; bp.d_out = d_out; bp.tape = (const float*)tape; bp.tape_format = tape_format; bp.d_t = d_t; bp.d_e = nullptr;
  bp.film_tiles = d_t + (size_t)m->L * m->H * (size_t)B * (size_t)P;
  bp.points = points; bp.d_grid_cl = d_grid_cl; bp.box_scale = m->box_scale; bp.gd = m->gd; bp.gh = m->gh; bp.gw = m->gw;
  bp.bf16_dump = use_bf16_dump(m, (long long)B * P);
  note_dump(m, d_t, (long long)B * P);
  { PhaseScope ph(PH_CHAIN, stream); return launch_siren_backward16w(m, bp, stream); }
}

extern "C" size_t fenerf_siren_grad_workspace_bytes(const FenerfModel* m, int B, int64_t P) {
  if (!m || B <= 0 || P <= 0) return 0;
  return align_up(wgrad_workspace_bytes(m, B, P), 256);
}

extern "C" int fenerf_siren_param_grads(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                                        const float* freq_geo, const float* phase_geo, const float* freq_app,
                                        const float* phase_app, const float* out, const float* d_out, const float* tape,
                                        const float* tape_e, const float* d_t, const FenerfSirenGrads* g, void* workspace,
                                        void* film_ws, void* stream) {
  return fenerf_siren_param_grads_fmt(m, B, P, points, ray_dirs, freq_geo, phase_geo, freq_app, phase_app, out, d_out, tape, FENERF_TAPE_F32, tape_e,
                                      d_t, g, nullptr, workspace, film_ws, stream);
}

extern "C" int fenerf_siren_param_grads_fmt(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                                            const float* freq_geo, const float* phase_geo, const float* freq_app,
                                            const float* phase_app, const float* out, const float* d_out, const void* tape,
                                            int tape_format, const float* tape_e, const float* d_t, const FenerfSirenGrads* g,
                                            const FenerfSirenGrads* weights, void* workspace, void* film_ws, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (int rcf = check_tape_format(m, tape_format)) return rcf;
  if (tape_format != FENERF_TAPE_F32) {
    if (!weights) return fail(FENERF_E_INVALID, "FENERF_TAPE_U16 / _F32_W: the FiLM layers' weights are required (the frequency gradients are derived from the weight-gradient sums)");
    for (int i = 0; i < m->n_geo; ++i) if (!weights->geo_w[i]) return fail(FENERF_E_INVALID, "FENERF_TAPE_U16 / _F32_W: weights->geo_w has a NULL entry");
    for (int i = 0; i < m->n_color; ++i) if (!weights->color_w[i]) return fail(FENERF_E_INVALID, "FENERF_TAPE_U16 / _F32_W: weights->color_w has a NULL entry");
  }
  if (!m->differentiable) return fail(FENERF_E_UNSUPPORTED, "model was not created with differentiable != 0");
  if (B <= 0 || P < 0) return fail(FENERF_E_INVALID, "B must be > 0 and P >= 0");
  if (P % 32) return fail(FENERF_E_INVALID, "differentiable path: points per image must be a multiple of 32");
  if (P == 0) return FENERF_OK;
  if (!points || !out || !d_out || !tape || !d_t || !g || !workspace || (m->grid_ch && !tape_e)) return fail(FENERF_E_INVALID, "NULL pointer");
  if (!g->d_freq_geo || !g->d_phase_geo || !g->d_freq_app || !g->d_phase_app) return fail(FENERF_E_INVALID, "grads: film pointer is NULL");
  // all weight / bias pointers NULL = FiLM gradients only (inversion); otherwise every one of them is required
  int have = 0, want = 0;
  for (int i = 0; i < m->n_geo; ++i) { want += 2; have += (g->geo_w[i] != nullptr) + (g->geo_b[i] != nullptr); }
  for (int i = 0; i < m->n_color; ++i) { want += 2; have += (g->color_w[i] != nullptr) + (g->color_b[i] != nullptr); }
  want += 4; have += (g->head_w != nullptr) + (g->head_b != nullptr) + (g->rgb_w != nullptr) + (g->rgb_b != nullptr);
  if (have != 0 && have != want) return fail(FENERF_E_INVALID, "grads: give every weight / bias buffer or none (FiLM gradients only)");
  const bool film_only = have == 0;
  if (film_only && tape_format != FENERF_TAPE_F32)
    return fail(FENERF_E_INVALID, "FiLM-only gradients need FENERF_TAPE_F32 (the chain kernel's second FiLM sum)");
  int rc = check_dump(m, d_t, (long long)B * P);
  if (rc) return rc;
  const float *fp, *pp;
  rc = film_prep(m, B, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream);
  if (rc) return rc;
  return launch_param_grads(m, B, P, points, ray_dirs, fp, pp, out, d_out, (const float*)tape, tape_e, d_t, *g, film_only, workspace, stream, nullptr,
                            tape_format, weights);
}

// Gradients wrt the SIREN's inputs (sample positions, view directions) from the fp32 d(theta) dump a fenerf_siren_backward* call left:
// one extra pass over two layers of the dump (fenerf_siren_inputgrad.hip).  include/fenerf.h says what it replaces.
extern "C" int fenerf_siren_input_grads(const FenerfModel* m, int B, int64_t P, const float* points, const float* freq_geo, const float* phase_geo,
                                        const float* freq_app, const float* phase_app, const float* d_t, const float* w_geo0,
                                        const float* w_color0, int w_color0_ld, float* d_points, float* d_dirs, void* film_ws, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (!m->differentiable) return fail(FENERF_E_UNSUPPORTED, "model was not created with differentiable != 0");
  if (B <= 0 || P < 0) return fail(FENERF_E_INVALID, "B must be > 0 and P >= 0");
  if (P % 32) return fail(FENERF_E_INVALID, "differentiable path: points per image must be a multiple of 32");
  if (P == 0) return FENERF_OK;
  if (!d_t || !w_geo0 || !w_color0) return fail(FENERF_E_INVALID, "d_t / w_geo0 / w_color0 is NULL");
  if (!d_points && !d_dirs) return fail(FENERF_E_INVALID, "d_points and d_dirs are both NULL");
  if (m->grid_ch != 0 && m->grid_ch != 32) return fail(FENERF_E_UNSUPPORTED, "fenerf_siren_input_grads: feature grids of 32 channels only");
  if (m->grid_ch && d_points && !points) return fail(FENERF_E_INVALID, "points is NULL (the grid's coordinate gradient needs the sample positions)");
  if (w_color0_ld < 3 + m->grid_ch) return fail(FENERF_E_INVALID, "w_color0_ld < 3 + grid channels");
  if (B > 65535) return fail(FENERF_E_UNSUPPORTED, "fenerf_siren_input_grads: at most 65535 images per call (one grid row per image)");
  if (use_bf16_dump(m, (long long)B * P))
    return fail(FENERF_E_UNSUPPORTED, "fenerf_siren_input_grads reads the fp32 d(theta) dump; this chunk's dump is bf16 (wgrad_bf16_min_points)");
  int rc = check_dump(m, d_t, (long long)B * P);
  if (rc) return rc;
  const float *fp, *pp;
  rc = film_prep(m, B, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream);
  if (rc) return rc;
  PhaseScope ph(PH_OTHER, stream);
  return launch_siren_input_grads(m, B, P, points, fp, d_t, w_geo0, w_color0, w_color0_ld, d_points, d_dirs, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Per-point modulation under autograd (round 6): SPATIALSIRENGRID.forward_with_frequencies_phase_shifts with one FiLM
// block per sample point (siren.py:464-477; FiLMLayer takes them unbroadcast, :119-122), differentiable.  Three calls in the shape of
// fenerf_siren_forward_save / _backward / _param_grads, on the exact-fp32 kernels (FENERF_PREC_F32 models): the forward keeps the tape,
// the chain reads every lane's own FiLM block, the weight-gradient jobs scale d(theta) by the point's own frequency while they stage it
// (the factorisation sum_images diag(f_image) sum_points d(theta) x^T of the per-image path does not exist here) and the gradients
// wrt the per-point frequencies / phase shifts leave as [B, P, n*H] tensors for the mapping network's own backward.
// ---------------------------------------------------------------------------------------------------------------------------------
static int pointwise_prep(const FenerfModel* m, int B, int64_t P, const float* fg, const float* pg, const float* fa, const float* pa, void* film_ws,
                          const float** fp, const float** pp, void* stream, int prepared = 0) {
  if (m->precision != FENERF_PREC_F32) return fail(FENERF_E_UNSUPPORTED, "differentiable per-point FiLM parameters: FENERF_PREC_F32 models only");
  if (!m->differentiable || !m->d_bwd_stream) return fail(FENERF_E_UNSUPPORTED, "model was not created with differentiable != 0");
  if (B <= 0 || P < 0) return fail(FENERF_E_INVALID, "B must be > 0 and P >= 0");
  if (P % 32) return fail(FENERF_E_INVALID, "differentiable path: points per image must be a multiple of 32");
  if (!fg || !pg || !fa || !pa || !film_ws) return fail(FENERF_E_INVALID, "film parameter / workspace pointer is NULL");
  float* f = (float*)film_ws;
  float* q = f + (size_t)B * (size_t)P * m->L * m->H;
  *fp = f; *pp = q;
  if (P == 0 || prepared) return FENERF_OK;     // prepared: film_ws still holds the blocks an earlier call of this family wrote for the same arguments
  PhaseScope ph(PH_FILM_PREP, stream);
  return launch_film_prep(m, (long long)B * P, fg, pg, fa, pa, f, q, stream, true);
}

extern "C" int fenerf_siren_forward_save_pointwise(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                                                   const float* freq_geo, const float* phase_geo, const float* freq_app,
                                                   const float* phase_app, float* out, float* tape, void* film_ws, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (m->grid_ch) return fail(FENERF_E_UNSUPPORTED, "per-point FiLM parameters: models without a feature grid (the reference's SPATIALSIRENGRID has none)");
  const float *fp, *pp;
  int rc = pointwise_prep(m, B, P, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream);
  if (rc || P == 0) return rc;
  if (!points || !out || !tape) return fail(FENERF_E_INVALID, "points / out / tape is NULL");
  SirenParams sp;
  fill_common(m, sp, fp, pp);
  sp.points = points; sp.pdirs = ray_dirs;
  sp.P = (long long)B * P; sp.pts_per_image = P; sp.n_per_ray = 1;
  sp.out = out; sp.tape = tape; sp.tape_format = FENERF_TAPE_F32;
  sp.film_per_point = 1;
  PhaseScope ph(PH_SIREN, stream);
  return launch_siren_f32(m, sp, stream);
}

extern "C" int fenerf_siren_backward_pointwise(const FenerfModel* m, int B, int64_t P, const float* freq_geo, const float* phase_geo,
                                               const float* freq_app, const float* phase_app, const float* out, const float* d_out,
                                               const float* tape, float* d_t, void* film_ws, int film_ws_prepared, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (m->grid_ch) return fail(FENERF_E_UNSUPPORTED, "per-point FiLM parameters: models without a feature grid");
  const float *fp, *pp;
  int rc = pointwise_prep(m, B, P, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream, film_ws_prepared);
  if (rc || P == 0) return rc;
  if (!out || !d_out || !tape || !d_t) return fail(FENERF_E_INVALID, "NULL pointer");
  SirenBwdParams bp;
  memset(&bp, 0, sizeof(bp));
  bp.stream = m->d_bwd_stream;
  bp.ring_offset_floats = (long long)m->bsh.ht_entries * 256;
  bp.fp = fp; bp.pp = pp;
  bp.P = (long long)B * P; bp.pts_per_image = P;
  bp.out = out; bp.d_out = d_out; bp.tape = tape; bp.tape_format = FENERF_TAPE_F32; bp.d_t = d_t; bp.d_e = nullptr;
  bp.film_tiles = d_t + (size_t)m->L * m->H * (size_t)B * (size_t)P;    // (written, not used: the sums mix points of different frequencies)
  bp.film_per_point = 1;
  note_dump(m, d_t, (long long)B * P);
  PhaseScope ph(PH_CHAIN, stream);
  return launch_siren_backward(m, bp, stream);
}

extern "C" int fenerf_siren_param_grads_pointwise(const FenerfModel* m, int B, int64_t P, const float* points, const float* ray_dirs,
                                                  const float* freq_geo, const float* phase_geo, const float* freq_app,
                                                  const float* phase_app, const float* out, const float* d_out, const float* tape,
                                                  const float* d_t, const FenerfSirenGrads* g, void* workspace, void* film_ws, int film_ws_prepared,
                                                  void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (m->grid_ch) return fail(FENERF_E_UNSUPPORTED, "per-point FiLM parameters: models without a feature grid");
  const float *fp, *pp;
  int rc = pointwise_prep(m, B, P, freq_geo, phase_geo, freq_app, phase_app, film_ws, &fp, &pp, stream, film_ws_prepared);
  if (rc || P == 0) return rc;
  if (!points || !out || !d_out || !tape || !d_t || !g || !workspace) return fail(FENERF_E_INVALID, "NULL pointer");
  if (!g->d_freq_geo || !g->d_phase_geo || !g->d_freq_app || !g->d_phase_app) return fail(FENERF_E_INVALID, "grads: film pointer is NULL");
  for (int i = 0; i < m->n_geo; ++i) if (!g->geo_w[i] || !g->geo_b[i]) return fail(FENERF_E_INVALID, "grads: every weight / bias buffer is required");
  for (int i = 0; i < m->n_color; ++i) if (!g->color_w[i] || !g->color_b[i]) return fail(FENERF_E_INVALID, "grads: every weight / bias buffer is required");
  if (!g->head_w || !g->head_b || !g->rgb_w || !g->rgb_b) return fail(FENERF_E_INVALID, "grads: every weight / bias buffer is required");
  rc = check_dump(m, d_t, (long long)B * P);
  if (rc) return rc;
  return launch_param_grads(m, B, P, points, ray_dirs, fp, pp, out, d_out, tape, nullptr, d_t, *g, false, workspace, stream, nullptr, FENERF_TAPE_F32, nullptr, 1);
}

extern "C" int fenerf_grid_backward(const FenerfModel* m, int64_t total_points, const float* points, const float* d_e,
                                    float* d_grid_cl, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (!m->grid_ch) return fail(FENERF_E_UNSUPPORTED, "model has no feature grid");
  if (total_points < 0) return fail(FENERF_E_INVALID, "total_points < 0");
  if (total_points == 0) return FENERF_OK;
  if (!points || !d_e || !d_grid_cl) return fail(FENERF_E_INVALID, "NULL pointer");
  { PhaseScope ph(PH_GRID, stream); return launch_grid_backward(m, total_points, points, d_e, d_grid_cl, stream); }
}

extern "C" int fenerf_grid_gradient_ncdhw(const FenerfModel* m, const float* d_grid_cl, float* d_grid_ncdhw, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (!m->grid_ch) return fail(FENERF_E_UNSUPPORTED, "model has no feature grid");
  if (!d_grid_cl || !d_grid_ncdhw) return fail(FENERF_E_INVALID, "NULL pointer");
  { PhaseScope ph(PH_GRID, stream); return launch_grid_unlayout(d_grid_cl, d_grid_ncdhw, m->gd, m->gh, m->gw, stream); }
}

extern "C" int fenerf_composite_backward(int64_t BR, int N, int C, int merge, const float* rows_a, const float* rows_b,
                                         const float* z_a, const float* z_b, const float* noise, const FenerfCompositeOpts* opts,
                                         const float* g_rgb, float* d_rows_a, float* d_rows_b, void* stream) {
  int rc = check_opts(opts);
  if (rc) return rc;
  const int M = merge ? 2 * N : N;
  if (BR < 0 || N < 1 || M > FENERF_MAX_RAY_SAMPLES || C < 2) return fail(FENERF_E_INVALID, "need BR >= 0, 1 <= samples <= 1024 (FENERF_MAX_RAY_SAMPLES), C >= 2");
  if (opts->fill_mode != FENERF_FILL_NONE) return fail(FENERF_E_UNSUPPORTED, "fill modes are not differentiated (generator.forward does not use them)");
  if (BR == 0) return FENERF_OK;
  if (!rows_a || !z_a || !g_rgb || !d_rows_a || (merge && (!rows_b || !z_b || !d_rows_b))) return fail(FENERF_E_INVALID, "NULL pointer");
  CompositeParams p;
  memset(&p, 0, sizeof(p));
  p.BR = BR; p.M = M; p.C = C; p.N = N;
  p.rows_a = rows_a; p.rows_b = rows_b; p.z_a = z_a; p.z_b = z_b; p.noise = noise; p.o = *opts;
  p.g_rgb = g_rgb; p.d_rows_a = d_rows_a; p.d_rows_b = d_rows_b;
  { PhaseScope ph(PH_COMPOSITE_BWD, stream); return launch_composite_backward(p, merge != 0, stream); }
}

extern "C" size_t fenerf_sparse_select_workspace_bytes(int B, int64_t P) {
  if (B < 1 || P < 1) return 0;
  return sparse_select_workspace_bytes(B, P);
}

extern "C" int fenerf_sparse_select(int B, int R, int N, int C, int64_t cap, const float* d_coarse, const float* d_fine, const float* z_coarse,
                                    const float* z_fine, const float* origins, const float* dirs, const int64_t* images, float* pts, float* rd,
                                    float* d_sel, int32_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 1 || B > 65535 || R < 1 || N < 1 || C < 1 || cap < 1) return fail(FENERF_E_INVALID, "need 1 <= B <= 65535, R, N, C, cap >= 1");
  if ((int64_t)R * N > (int64_t)1 << 36) return fail(FENERF_E_INVALID, "too many samples per image");
  if (!d_coarse || !z_coarse || !origins || !dirs || !pts || !d_sel || !counts || !workspace || (!d_fine) != (!z_fine))
    return fail(FENERF_E_INVALID, "NULL pointer (d_fine and z_fine may be NULL together: one pass)");
  if (workspace_bytes < sparse_select_workspace_bytes(B, (int64_t)R * N)) return fail(FENERF_E_INVALID, "workspace too small (fenerf_sparse_select_workspace_bytes)");
  { PhaseScope ph(PH_OTHER, stream); return launch_sparse_select(B, R, N, C, cap, d_coarse, d_fine, z_coarse, z_fine, origins, dirs, (const long long*)images, pts, rd, d_sel, counts, workspace, stream); }
}

// workspace layout of fenerf_render_forward
namespace {
struct RenderWs {
  size_t film, coarse, fine, wts, zf, total;
};
RenderWs render_ws(const FenerfModel* m, int B, int R, int N, int hier) {
  RenderWs w;
  const size_t pts = (size_t)B * R * N;
  size_t off = 0;
  w.film = off; off += fenerf_film_workspace_bytes(m, B);
  w.coarse = off; off += align_up(pts * m->C * sizeof(float), 256);
  w.fine = off; off += hier ? align_up(pts * m->C * sizeof(float), 256) : 0;
  w.wts = off; off += hier ? align_up(pts * sizeof(float), 256) : 0;
  w.zf = off; off += hier ? align_up(pts * sizeof(float), 256) : 0;
  w.total = off;
  return w;
}
}  // namespace

extern "C" size_t fenerf_render_workspace_bytes(const FenerfModel* m, int B, int R, int N, int hierarchical) {
  if (!m || B <= 0 || R <= 0 || N <= 0) return 0;
  return render_ws(m, B, R, N, hierarchical).total;
}

extern "C" int fenerf_render_forward(const FenerfModel* m, int B, int R, int N, int hierarchical, int lock_view,
                                     const float* origins, const float* dirs, const float* z_coarse, const float* u,
                                     const float* noise_coarse, const float* noise_final, const float* freq_geo,
                                     const float* phase_geo, const float* freq_app, const float* phase_app,
                                     const FenerfCompositeOpts* opts, float* out_rgb, float* out_depth,
                                     float* out_weights, float* out_wsum, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  int rc = check_opts(opts);
  if (rc) return rc;
  if (B <= 0 || R <= 0 || N <= 0) return fail(FENERF_E_INVALID, "B, R, N must be > 0");
  const int M = hierarchical ? 2 * N : N;
  if (M > FENERF_MAX_RAY_SAMPLES) return fail(FENERF_E_INVALID, "at most 1024 samples per ray (512 + 512 hierarchical; FENERF_MAX_RAY_SAMPLES)");
  if (hierarchical && N < 3) return fail(FENERF_E_INVALID, "hierarchical sampling needs num_steps >= 3");
  if (!origins || !dirs || !z_coarse || !out_rgb) return fail(FENERF_E_INVALID, "origins / dirs / z_coarse / out_rgb is NULL");
  if (hierarchical && !u) return fail(FENERF_E_INVALID, "hierarchical sampling needs u");
  if (opts->fill_mode == FENERF_FILL_EVAL_WHITE_BACK && m->C != 4) return fail(FENERF_E_INVALID, "eval_white_back needs a 3-channel model");
  const RenderWs ws = render_ws(m, B, R, N, hierarchical);
  if (!workspace || workspace_bytes < ws.total) return fail(FENERF_E_INVALID, "workspace too small (see fenerf_render_workspace_bytes)");
  char* base = (char*)workspace;
  if (!freq_geo || !phase_geo || !freq_app || !phase_app) return fail(FENERF_E_INVALID, "film parameter pointer is NULL");
  // FiLM pre-pass (f' / p' of every image, [B][L][H] each): computed by the coarse SIREN launch itself -- every workgroup prepares the
  // blocks of the images its tiles belong to before its first tile (fenerf_film.h) -- and reused by the fine launch
  float* fp = (float*)(base + ws.film);
  float* pp = fp + (size_t)B * m->L * m->H;
  float* coarse = (float*)(base + ws.coarse);
  const long long BR = (long long)B * R;

  SirenParams sp;
  fill_common(m, sp, fp, pp);
  sp.origins = origins; sp.dirs = dirs; sp.n_per_ray = N; sp.lock_view = lock_view;
  sp.P = BR * N; sp.pts_per_image = (long long)R * N;
  sp.z = z_coarse; sp.out = coarse;
  sp.raw_fg = freq_geo; sp.raw_pg = phase_geo; sp.raw_fa = freq_app; sp.raw_pa = phase_app;
  sp.film_bias = m->d_consts + CONST_FILM_BIAS;
  sp.film_inv_scale = m->precision == FENERF_PREC_F16X3 ? m->d_consts + CONST_FILM_BIAS + (size_t)m->L * m->H : nullptr;
  sp.n_images = B;
  CompositeParams cp;
  if (!hierarchical) {
    rc = run_siren(m, sp, stream);                     // the only pass     (generators.py:479)
    if (rc) return rc;
    memset(&cp, 0, sizeof(cp));
    cp.BR = BR; cp.M = N; cp.C = m->C; cp.N = N;
    cp.rows_a = coarse; cp.z_a = z_coarse; cp.noise = noise_final; cp.o = *opts;
    cp.out_rgb = out_rgb; cp.out_depth = out_depth; cp.out_weights = out_weights; cp.out_wsum = out_wsum;
    cp.out_ch = out_channels(m->C, opts);
    return run_composite(cp, false, stream);           // (generators.py:519)
  }
  float* fine = (float*)(base + ws.fine);
  float* zf = (float*)(base + ws.zf);
  FusedRenderPlan plan;
  if ((g_render_fusion == FENERF_FUSION_FORCE || (g_render_fusion == FENERF_FUSION_AUTO && kFusionAutoEnabled)) &&
      fused_render_plan(m, B, R, N, g_render_fusion == FENERF_FUSION_AUTO, &plan)) {
    // ONE launch (fenerf_siren_f16w.hip, FUSED): per ray group coarse octs -> weights + resampling -> fine octs -> merge + composite
    CompositeParams c1, c2;
    memset(&c1, 0, sizeof(c1));
    c1.BR = BR; c1.M = N; c1.C = m->C; c1.N = N;
    c1.rows_a = coarse; c1.z_a = z_coarse; c1.noise = noise_coarse;
    c1.o.clamp_mode = opts->clamp_mode; c1.o.noise_std = opts->noise_std;
    c1.sigma_only = 1; c1.out_ch = m->C - 1;
    c1.u = u; c1.z_fine = zf;
    memset(&c2, 0, sizeof(c2));
    c2.BR = BR; c2.M = 2 * N; c2.C = m->C; c2.N = N;
    c2.rows_a = fine; c2.rows_b = coarse; c2.z_a = zf; c2.z_b = z_coarse; c2.noise = noise_final; c2.o = *opts;
    c2.out_rgb = out_rgb; c2.out_depth = out_depth; c2.out_weights = out_weights; c2.out_wsum = out_wsum;
    c2.out_ch = out_channels(m->C, opts);
    sp.out = coarse;
    PhaseScope ph(PH_RENDER_FUSED, stream);
    return launch_render16w_fused(m, sp, plan, zf, fine, c1, c2, stream);
  }
  rc = run_siren(m, sp, stream);                       // coarse pass   (generators.py:479)
  if (rc) return rc;
  sp.raw_fg = nullptr;                                    // the fine pass finds f' / p' in the workspace
  memset(&cp, 0, sizeof(cp));                             // coarse weights (generators.py:487) and, in the same wave per ray, the
  cp.BR = BR; cp.M = N; cp.C = m->C; cp.N = N;            // inverse-CDF resampling from them (generators.py:489-499): one launch
  cp.rows_a = coarse; cp.z_a = z_coarse; cp.noise = noise_coarse;
  cp.o.clamp_mode = opts->clamp_mode; cp.o.noise_std = opts->noise_std;
  cp.sigma_only = 1; cp.out_ch = m->C - 1;
  cp.u = u; cp.z_fine = zf;
  rc = run_composite(cp, false, stream);
  if (rc) return rc;
  sp.z = zf; sp.out = fine;
  rc = run_siren(m, sp, stream);                       // fine pass     (generators.py:505)
  if (rc) return rc;
  memset(&cp, 0, sizeof(cp));                             // merge + final composite (generators.py:508-519)
  cp.BR = BR; cp.M = 2 * N; cp.C = m->C; cp.N = N;
  cp.rows_a = fine; cp.rows_b = coarse; cp.z_a = zf; cp.z_b = z_coarse; cp.noise = noise_final; cp.o = *opts;
  cp.out_rgb = out_rgb; cp.out_depth = out_depth; cp.out_weights = out_weights; cp.out_wsum = out_wsum;
  cp.out_ch = out_channels(m->C, opts);
  return run_composite(cp, true, stream);
}

// =====================================================================================================================================
// The differentiable hierarchical render as TWO calls (round 5; SURVEY.md 8b "fenerf_render_backward"): what rounds 1-4 orchestrated in
// Python (fenerf_amd/generators/autograd.py::_hierarchical_forward / HierarchicalRenderFunction.backward, siren/autograd.py::
// chunked_backward: ~600 lines of chunk planning, workspace carving, launch order and gradient sums) lives here, so that a host that is
// not Python can take a generator step (generators.py:479-527 + g_loss.backward(), train_double_latent_semantic.py:402-446) with two
// entry points, and so that the Python host's step has no glue launches left between the library's kernels.
// =====================================================================================================================================
namespace {
const long long kDefaultChunkPoints = 393216;          // one pass of a 128 x 128 x 24 image: profiles/r04_gstep_overlap.md (chunk-size sweep)
const long long kDefaultFilmSumsBudget = 1LL << 30;    // bytes of per-unit FiLM sums one FiLM-only chain launch may write

struct RenderSave {     // byte offsets into the caller's `save` buffer (everything the backward needs that the caller does not hold)
  long long P, Pp;
  size_t fp2, pp2, pts2, rd, out2, tape2, tape_half, tape_e2, zf, rows_c, rows_f, total;
};
RenderSave render_save(const FenerfModel* m, int B, int R, int N, int tape_format, int lock_view) {
  RenderSave s;
  s.P = (long long)R * N;
  s.Pp = (s.P + 31) / 32 * 32;
  const size_t Pp = (size_t)s.Pp, P = (size_t)s.P;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
  s.fp2 = take((size_t)2 * B * m->L * m->H * sizeof(float));
  s.pp2 = take((size_t)2 * B * m->L * m->H * sizeof(float));
  s.pts2 = take((size_t)2 * B * Pp * 3 * sizeof(float));
  s.rd = take(lock_view ? 0 : (size_t)2 * B * Pp * 3 * sizeof(float));     // per-point view directions, once per pass (a backward chunk may span both)
  s.out2 = take((size_t)2 * B * Pp * m->C * sizeof(float));
  s.tape_half = (size_t)(tape_format == FENERF_TAPE_U16 ? 2 : 4) * m->L * m->H * B * Pp;     // pass 1 | pass 2 + the slack behind the last tile
  s.tape2 = take(s.tape_half + fenerf_siren_tape_bytes(m, (int64_t)B * s.Pp, tape_format));
  s.tape_e2 = take(m->grid_ch ? (size_t)2 * B * Pp * 32 * sizeof(float) : 0);
  s.zf = take((size_t)B * P * sizeof(float));
  s.rows_c = take(Pp != P ? (size_t)B * P * m->C * sizeof(float) : 0);      // unpadded row copies for the composite kernels (ragged shapes only)
  s.rows_f = take(Pp != P ? (size_t)B * P * m->C * sizeof(float) : 0);
  s.total = off;
  return s;
}

struct Chunk { int b, nb; long long s, n; };
std::vector<Chunk> plan_chunks(int nB, long long Pp, long long max_points) {
  max_points = max_points / 128 * 128;
  if (max_points < 128) max_points = 128;
  std::vector<Chunk> c;
  if (Pp <= max_points) {
    const int per = (int)(max_points / Pp) < 1 ? 1 : (int)(max_points / Pp);
    for (int b = 0; b < nB; b += per) c.push_back(Chunk{b, nB - b < per ? nB - b : per, 0, Pp});
  } else {
    for (int b = 0; b < nB; ++b)
      for (long long s = 0; s < Pp; s += max_points) c.push_back(Chunk{b, 1, s, Pp - s < max_points ? Pp - s : max_points});
  }
  return c;
}

// per-model sizes of the weight-gradient outputs, in FenerfSirenGrads order
struct GradSizes { long long geo_w[FENERF_MAX_GEO], geo_b[FENERF_MAX_GEO], color_w[FENERF_MAX_COLOR], color_b[FENERF_MAX_COLOR], head_w, head_b, rgb_w, rgb_b, total; };
GradSizes grad_sizes(const FenerfModel* m) {
  GradSizes z;
  memset(&z, 0, sizeof(z));
  const long long H = m->H;
  for (int i = 0; i < m->n_geo; ++i) { z.geo_w[i] = i == 0 ? H * 3 : H * H; z.geo_b[i] = H; }
  for (int i = 0; i < m->n_color; ++i) { z.color_w[i] = i == 0 ? H * (3 + m->grid_ch + H) : H * H; z.color_b[i] = H; }
  z.head_w = 32 * H; z.head_b = 32; z.rgb_w = 3 * H; z.rgb_b = 3;
  for (int i = 0; i < m->n_geo; ++i) z.total += z.geo_w[i] + z.geo_b[i];
  for (int i = 0; i < m->n_color; ++i) z.total += z.color_w[i] + z.color_b[i];
  z.total += z.head_w + z.head_b + z.rgb_w + z.rgb_b;
  return z;
}
// carve a FenerfSirenGrads' weight buffers out of `base` (floats), in grad_sizes order
void carve_grads(const FenerfModel* m, const GradSizes& z, float* base, FenerfSirenGrads* g) {
  for (int i = 0; i < m->n_geo; ++i) { g->geo_w[i] = base; base += z.geo_w[i]; g->geo_b[i] = base; base += z.geo_b[i]; }
  for (int i = 0; i < m->n_color; ++i) { g->color_w[i] = base; base += z.color_w[i]; g->color_b[i] = base; base += z.color_b[i]; }
  g->head_w = base; base += z.head_w; g->head_b = base; base += z.head_b; g->rgb_w = base; base += z.rgb_w; g->rgb_b = base; base += z.rgb_b;
}

// points one backward launch may take: the d(theta) dump's budget, or -- FiLM-only chain launches of f16x3 models write no dump -- what
// their per-unit FiLM sums may occupy (and the kernel's 32-bit tile arithmetic: 2^24 points)
long long backward_max_points(const FenerfModel* m, long long Pp, bool film16, long long chunk_points, long long film_budget) {
  if (!film16) return chunk_points > 0 ? chunk_points : kDefaultChunkPoints;
  const double bytes_pp = 4.0 * (double)fenerf_siren_film_sums_floats(m, 1, Pp) / (double)Pp;
  double cap = (double)(film_budget > 0 ? film_budget : kDefaultFilmSumsBudget) / bytes_pp;
  if (cap > (double)(1 << 24)) cap = (double)(1 << 24);
  return (long long)cap;
}

struct BackwardWs {
  size_t d_out2, d_fc, dump, dump_stride, d_grid_cl, d_e, wgrad, film2, scratch, total;      // dump_stride: bytes of one chunk's dump (split backward: several slots)
  long long max_chunk_points, film_row;   // film_row = floats of one image's four FiLM gradient rows
  int max_nb;
};
BackwardWs backward_ws(const FenerfModel* m, int B, int R, int N, int film_only, long long chunk_points, long long film_budget, int dump_slots = 1) {
  BackwardWs w;
  const long long P = (long long)R * N, Pp = (P + 31) / 32 * 32;
  const bool film16 = film_only && m->precision == FENERF_PREC_F16X3;
  const std::vector<Chunk> chunks = plan_chunks(2 * B, Pp, backward_max_points(m, Pp, film16, chunk_points, film_budget));
  w.max_chunk_points = 0; w.max_nb = 0;
  size_t wg = 0, sums = 0;
  for (const Chunk& c : chunks) {
    if ((long long)c.nb * c.n > w.max_chunk_points) w.max_chunk_points = (long long)c.nb * c.n;
    if (c.nb > w.max_nb) w.max_nb = c.nb;
    const size_t b = wgrad_workspace_bytes(m, c.nb, c.n);
    if (b > wg) wg = b;
    const size_t f = fenerf_siren_film_sums_floats(m, c.nb, c.n) * sizeof(float);
    if (f > sums) sums = f;
  }
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
  w.d_out2 = take((size_t)2 * B * Pp * m->C * sizeof(float));
  w.d_fc = take(Pp != P ? (size_t)2 * B * P * m->C * sizeof(float) : 0);
  w.dump_stride = align_up(film16 ? sums : fenerf_siren_dtheta_floats(m, w.max_chunk_points) * sizeof(float), 256);
  {
    const size_t slots = (size_t)(dump_slots < 1 ? 1 : (dump_slots > (int)chunks.size() ? (int)chunks.size() : dump_slots));
    w.dump = take(w.dump_stride * (slots < 1 ? 1 : slots));
  }
  w.d_grid_cl = take(m->grid_ch && !film_only ? (size_t)m->gd * m->gh * m->gw * 32 * sizeof(float) : 0);
  w.d_e = take(m->grid_ch && !fenerf_siren_backward_fuses_grid(m) ? (size_t)w.max_chunk_points * 32 * sizeof(float) : 0);
  w.wgrad = take(align_up(wg, 256));
  w.film_row = (long long)2 * (m->n_geo + m->n_color) * m->H;
  w.film2 = take((size_t)2 * B * w.film_row * sizeof(float));
  // a later chunk's gradients before they are added to the first one's: every weight tensor + the FiLM rows of the chunk's images
  w.scratch = take(chunks.size() > 1 ? ((size_t)(film_only ? 0 : grad_sizes(m).total) + (size_t)w.max_nb * w.film_row) * sizeof(float) : 0);
  w.total = off;
  return w;
}
}  // namespace

extern "C" size_t fenerf_render_save_bytes(const FenerfModel* m, int B, int R, int N, int tape_format, int lock_view) {
  if (!m || B <= 0 || R <= 0 || N <= 0) return 0;
  return render_save(m, B, R, N, tape_format, lock_view).total;
}

extern "C" int fenerf_render_forward_save(const FenerfModel* m, int B, int R, int N, int lock_view, const float* origins, const float* dirs,
                                          const float* z_coarse, const float* u, const float* noise_coarse, const float* noise_final,
                                          const float* freq_geo, const float* phase_geo, const float* freq_app, const float* phase_app,
                                          const FenerfCompositeOpts* opts, float* out_rgb, float* out_depth, void* save, size_t save_bytes,
                                          int tape_format, void* stream) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (!m->differentiable) return fail(FENERF_E_UNSUPPORTED, "model was not created with differentiable != 0");
  int rc = check_opts(opts);
  if (rc) return rc;
  if ((rc = check_tape_format(m, tape_format))) return rc;
  if (B <= 0 || R <= 0 || N < 3 || 2 * N > FENERF_MAX_RAY_SAMPLES) return fail(FENERF_E_INVALID, "need B, R > 0 and 3 <= num_steps <= 512 (hierarchical render; FENERF_MAX_RAY_SAMPLES / 2)");
  if (opts->fill_mode != FENERF_FILL_NONE) return fail(FENERF_E_UNSUPPORTED, "fill modes are not differentiated (generator.forward does not use them)");
  if (!origins || !dirs || !z_coarse || !u || !out_rgb || !out_depth) return fail(FENERF_E_INVALID, "origins / dirs / z_coarse / u / out_rgb / out_depth is NULL");
  if (!freq_geo || !phase_geo || !freq_app || !phase_app) return fail(FENERF_E_INVALID, "film parameter pointer is NULL");
  const RenderSave sv = render_save(m, B, R, N, tape_format, lock_view);
  if (!save || save_bytes < sv.total) return fail(FENERF_E_INVALID, "save buffer too small (see fenerf_render_save_bytes)");
  char* base = (char*)save;
  const long long P = sv.P, Pp = sv.Pp, BR = (long long)B * R;
  const int C = m->C;
  float* fp2 = (float*)(base + sv.fp2);
  float* pp2 = (float*)(base + sv.pp2);
  float* pts2 = (float*)(base + sv.pts2);
  float* rd = lock_view ? nullptr : (float*)(base + sv.rd);
  float* out2 = (float*)(base + sv.out2);
  char* tape2 = base + sv.tape2;
  float* tape_e2 = m->grid_ch ? (float*)(base + sv.tape_e2) : nullptr;
  float* zf = (float*)(base + sv.zf);
  // FiLM pre-pass once, for both passes and for the backward: f' / p' of image b at rows b and B + b (pass-major "images")
  { PhaseScope ph(PH_FILM_PREP, stream); if ((rc = launch_film_prep(m, B, freq_geo, phase_geo, freq_app, phase_app, fp2, pp2, stream, false, true))) return rc; }
  // ---- coarse pass (generators.py:468-479)
  { PhaseScope ph(PH_OTHER, stream); if ((rc = launch_render_points(B, R, N, Pp, origins, dirs, z_coarse, pts2, rd, stream))) return rc; }
  SirenParams sp;
  fill_common(m, sp, fp2, pp2);
  sp.points = pts2; sp.pdirs = rd;
  sp.P = (long long)B * Pp; sp.pts_per_image = Pp; sp.n_per_ray = 1;
  sp.out = out2; sp.tape = (float*)tape2; sp.tape_e = tape_e2; sp.tape_format = tape_format;
  if ((rc = run_siren(m, sp, stream))) return rc;
  const float* coarse = out2;
  const float* fine = out2 + (size_t)B * Pp * C;
  if (Pp != P) {
    PhaseScope ph(PH_OTHER, stream);
    if ((rc = launch_pad_rows(out2, (float*)(base + sv.rows_c), B, P, Pp, C, false, stream))) return rc;
    coarse = (const float*)(base + sv.rows_c);
  }
  // ---- coarse weights and, in the same wave per ray, the inverse-CDF resampling (generators.py:485-499; constants of the graph)
  CompositeParams cp;
  memset(&cp, 0, sizeof(cp));
  cp.BR = BR; cp.M = N; cp.C = C; cp.N = N;
  cp.rows_a = coarse; cp.z_a = z_coarse; cp.noise = noise_coarse;
  cp.o.clamp_mode = opts->clamp_mode; cp.o.noise_std = opts->noise_std;
  cp.sigma_only = 1; cp.out_ch = C - 1;
  cp.u = u; cp.z_fine = zf;
  if ((rc = run_composite(cp, false, stream))) return rc;
  // ---- fine pass (generators.py:504-505)
  { PhaseScope ph(PH_OTHER, stream); if ((rc = launch_render_points(B, R, N, Pp, origins, dirs, zf, pts2 + (size_t)B * Pp * 3, rd ? rd + (size_t)B * Pp * 3 : nullptr, stream))) return rc; }
  sp.points = pts2 + (size_t)B * Pp * 3;
  sp.out = out2 + (size_t)B * Pp * C;
  sp.tape = (float*)(tape2 + sv.tape_half);
  sp.tape_e = tape_e2 ? tape_e2 + (size_t)B * Pp * 32 : nullptr;
  if ((rc = run_siren(m, sp, stream))) return rc;
  if (Pp != P) {
    PhaseScope ph(PH_OTHER, stream);
    if ((rc = launch_pad_rows(out2 + (size_t)B * Pp * C, (float*)(base + sv.rows_f), B, P, Pp, C, false, stream))) return rc;
    fine = (const float*)(base + sv.rows_f);
  }
  // ---- merge + final composite (generators.py:508-519)
  memset(&cp, 0, sizeof(cp));
  cp.BR = BR; cp.M = 2 * N; cp.C = C; cp.N = N;
  cp.rows_a = fine; cp.rows_b = coarse; cp.z_a = zf; cp.z_b = z_coarse; cp.noise = noise_final; cp.o = *opts;
  cp.out_rgb = out_rgb; cp.out_depth = out_depth;
  cp.out_ch = out_channels(C, opts);
  return run_composite(cp, true, stream);
}

extern "C" size_t fenerf_render_backward_workspace_bytes(const FenerfModel* m, int B, int R, int N, int film_only, int64_t chunk_points,
                                                         int64_t film_sums_budget_bytes) {
  if (!m || B <= 0 || R <= 0 || N <= 0) return 0;
  return backward_ws(m, B, R, N, film_only, chunk_points, film_sums_budget_bytes).total;
}

// stage 0: the whole backward in one call (fenerf_render_backward).  stage 1 / 2: the same launches cut in two (fenerf_render_backward_stage):
// 1 = composite backward, chain AND weight gradients of every chunk but the last `keep_chunks`, the chains of those last chunks -- each
// into its own dump slot -- and the finished grid gradient; 2 = the weight gradients of the last chunks, the sums, the FiLM fold.  Same
// kernels on the same chunks in the same order per gradient tensor: stage 1 + stage 2 = stage 0 bit for bit (atomics of the grid aside).
static int render_backward_impl(const FenerfModel* m, int B, int R, int N, int lock_view, const void* save, size_t save_bytes,
                                int tape_format, const float* z_coarse, const float* noise_final, const FenerfCompositeOpts* opts,
                                const float* g_rgb, const FenerfSirenGrads* grads, float* d_grid_ncdhw, const FenerfSirenGrads* weights,
                                int64_t chunk_points, int64_t film_sums_budget_bytes, void* workspace, size_t workspace_bytes,
                                void* stream, int keep_chunks, int stage) {
  if (!m) return fail(FENERF_E_INVALID, "model is NULL");
  if (!m->differentiable || !m->d_bwd_stream) return fail(FENERF_E_UNSUPPORTED, "model was not created with differentiable != 0");
  int rc = check_opts(opts);
  if (rc) return rc;
  if ((rc = check_tape_format(m, tape_format))) return rc;
  if (B <= 0 || R <= 0 || N < 3 || 2 * N > FENERF_MAX_RAY_SAMPLES) return fail(FENERF_E_INVALID, "need B, R > 0 and 3 <= num_steps <= 512 (hierarchical render; FENERF_MAX_RAY_SAMPLES / 2)");
  if (!save || !grads || !workspace || (stage != 2 && (!z_coarse || !g_rgb))) return fail(FENERF_E_INVALID, "NULL pointer");
  if (!grads->d_freq_geo || !grads->d_phase_geo || !grads->d_freq_app || !grads->d_phase_app) return fail(FENERF_E_INVALID, "grads: film pointer is NULL");
  int have = 0, want = 0;
  for (int i = 0; i < m->n_geo; ++i) { want += 2; have += (grads->geo_w[i] != nullptr) + (grads->geo_b[i] != nullptr); }
  for (int i = 0; i < m->n_color; ++i) { want += 2; have += (grads->color_w[i] != nullptr) + (grads->color_b[i] != nullptr); }
  want += 4; have += (grads->head_w != nullptr) + (grads->head_b != nullptr) + (grads->rgb_w != nullptr) + (grads->rgb_b != nullptr);
  if (have != 0 && have != want) return fail(FENERF_E_INVALID, "grads: give every weight / bias buffer or none (FiLM gradients only)");
  const bool film_only = have == 0;
  if (stage != 0 && (film_only || keep_chunks < 1)) return fail(FENERF_E_INVALID, "the two-stage backward takes weight gradients and keeps at least one chunk");
  if (film_only && tape_format != FENERF_TAPE_F32) return fail(FENERF_E_INVALID, "FiLM-only gradients need FENERF_TAPE_F32 (the chain kernel's second FiLM sum)");
  if (tape_format != FENERF_TAPE_F32 && !weights) return fail(FENERF_E_INVALID, "FENERF_TAPE_U16 / _F32_W: the FiLM layers' weights are required");
  if (!film_only && m->grid_ch && !d_grid_ncdhw && stage != 2) return fail(FENERF_E_INVALID, "d_grid_ncdhw is NULL");
  const RenderSave sv = render_save(m, B, R, N, tape_format, lock_view);
  if (save_bytes < sv.total) return fail(FENERF_E_INVALID, "save buffer too small (see fenerf_render_save_bytes)");
  const BackwardWs wsz = backward_ws(m, B, R, N, film_only, chunk_points, film_sums_budget_bytes, stage == 0 ? 1 : keep_chunks);
  if (workspace_bytes < wsz.total) return fail(FENERF_E_INVALID, "workspace too small (see fenerf_render_backward_workspace_bytes / _split_workspace_bytes)");
  const char* sb = (const char*)save;
  char* wb = (char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  const long long P = sv.P, Pp = sv.Pp, BR = (long long)B * R;
  const int C = m->C, H = m->H, L = m->L, ng = m->n_geo, nc = m->n_color, nB = 2 * B;
  const float* fp2 = (const float*)(sb + sv.fp2);
  const float* pp2 = (const float*)(sb + sv.pp2);
  const float* pts2 = (const float*)(sb + sv.pts2);
  const float* rd = lock_view ? nullptr : (const float*)(sb + sv.rd);
  const float* out2 = (const float*)(sb + sv.out2);
  const char* tape2 = sb + sv.tape2;
  const float* tape_e2 = m->grid_ch ? (const float*)(sb + sv.tape_e2) : nullptr;
  const float* zf = (const float*)(sb + sv.zf);
  float* d_out2 = (float*)(wb + wsz.d_out2);
  // ---- composite backward: the gradient wrt the two passes' rows, written where the chain reads it (coarse | fine halves, pass-major)
  if (stage != 2) {
    CompositeParams cp;
    memset(&cp, 0, sizeof(cp));
    cp.BR = BR; cp.M = 2 * N; cp.C = C; cp.N = N;
    cp.rows_a = Pp == P ? out2 + (size_t)B * Pp * C : (const float*)(sb + sv.rows_f);
    cp.rows_b = Pp == P ? out2 : (const float*)(sb + sv.rows_c);
    cp.z_a = zf; cp.z_b = z_coarse; cp.noise = noise_final; cp.o = *opts;
    cp.g_rgb = g_rgb;
    float* d_f = Pp == P ? d_out2 + (size_t)B * Pp * C : (float*)(wb + wsz.d_fc);
    float* d_c = Pp == P ? d_out2 : (float*)(wb + wsz.d_fc) + (size_t)B * P * C;
    cp.d_rows_a = d_f; cp.d_rows_b = d_c;
    { PhaseScope ph(PH_COMPOSITE_BWD, stream); if ((rc = launch_composite_backward(cp, true, stream))) return rc; }
    if (Pp != P) {
      PhaseScope ph(PH_OTHER, stream);
      if ((rc = launch_pad_rows(d_c, d_out2, B, P, Pp, C, true, stream))) return rc;
      if ((rc = launch_pad_rows(d_f, d_out2 + (size_t)B * Pp * C, B, P, Pp, C, true, stream))) return rc;
    }
  }
  // ---- chunks of whole (pass, image) pairs -- or point ranges of one -- : chain, then the weight / FiLM gradients of the chunk
  const bool film16 = film_only && m->precision == FENERF_PREC_F16X3;
  const std::vector<Chunk> chunks = plan_chunks(nB, Pp, backward_max_points(m, Pp, film16, chunk_points, film_sums_budget_bytes));
  float* const dump0 = (float*)(wb + wsz.dump);
  float* d_grid_cl = (m->grid_ch && !film_only) ? (float*)(wb + wsz.d_grid_cl) : nullptr;
  float* d_e = (m->grid_ch && !fenerf_siren_backward_fuses_grid(m)) ? (float*)(wb + wsz.d_e) : nullptr;
  if (d_grid_cl && stage != 2) HIP_TRY(hipMemsetAsync(d_grid_cl, 0, (size_t)m->gd * m->gh * m->gw * 32 * sizeof(float), st));
  float* film2 = (float*)(wb + wsz.film2);      // [d_freq_geo | d_phase_geo: nB x ng H each][d_freq_app | d_phase_app: nB x nc H each]
  float* f2[4] = {film2, film2 + (size_t)nB * ng * H, film2 + (size_t)2 * nB * ng * H, film2 + (size_t)2 * nB * ng * H + (size_t)nB * nc * H};
  const long long frow[4] = {(long long)ng * H, (long long)ng * H, (long long)nc * H, (long long)nc * H};
  const GradSizes gz = grad_sizes(m);
  const size_t LHw = (size_t)(tape_format == FENERF_TAPE_U16 ? 2 : 4) * L * H;      // tape bytes per point
  const int n_chunks = (int)chunks.size();
  const int n_late = stage == 0 ? 0 : (keep_chunks < n_chunks ? keep_chunks : n_chunks), n_early = n_chunks - n_late;
  bool first = stage != 2 || n_early == 0;      // does the next weight-gradient call write the output buffers (or a scratch set that is then added)?
  for (int ci = 0; ci < n_chunks; ++ci) {
    const Chunk& c = chunks[ci];
    const bool late = ci >= n_early;
    if (stage == 2 && !late) continue;
    const bool do_chain = stage != 2, do_wgrad = stage == 0 || (stage == 1 && !late) || (stage == 2 && late);
    float* const dump = dump0 + (late ? (size_t)(ci - n_early) * (wsz.dump_stride / sizeof(float)) : 0);
    const long long g0 = (long long)c.b * Pp + c.s, npts = (long long)c.nb * c.n;
    const float* fp_c = fp2 + (size_t)c.b * L * H;
    const float* pp_c = pp2 + (size_t)c.b * L * H;
    const float* out_c = out2 + (size_t)g0 * C;
    const float* dout_c = d_out2 + (size_t)g0 * C;
    const float* pts_c = pts2 + (size_t)g0 * 3;
    const char* tape_c = tape2 + (size_t)g0 * LHw;
    SirenBwdParams bp;
    memset(&bp, 0, sizeof(bp));
    bp.stream = m->d_bwd_stream;
    bp.ring_offset_floats = (long long)m->bsh.ht_entries * 256;
    bp.fp = fp_c; bp.pp = pp_c;
    bp.P = npts; bp.pts_per_image = c.n;
    bp.out = out_c; bp.d_out = dout_c; bp.tape = (const float*)tape_c; bp.tape_format = tape_format;
    const float* film_sums = nullptr;
    if (film16) {
      bp.d_t = nullptr; bp.d_e = nullptr; bp.film_tiles = dump; film_sums = dump;
    } else {
      bp.d_t = dump; bp.film_tiles = dump + (size_t)L * H * (size_t)npts;
      bp.bf16_dump = use_bf16_dump(m, npts);
      if (m->grid_ch && !film_only) {
        if (fenerf_siren_backward_fuses_grid(m)) { bp.points = pts_c; bp.d_grid_cl = d_grid_cl; bp.box_scale = m->box_scale; bp.gd = m->gd; bp.gh = m->gh; bp.gw = m->gw; }
        else bp.d_e = d_e;
      } else if (m->grid_ch) {
        bp.d_e = d_e;      // FiLM-only on an exact-fp32 model: the chain still writes d(grid features); nobody reads them
      }
    }
    if (do_chain) {
      PhaseScope ph(PH_CHAIN, stream);
      rc = m->precision == FENERF_PREC_F16X3 ? launch_siren_backward16w(m, bp, stream) : launch_siren_backward(m, bp, stream);
      if (rc) return rc;
    }
    if (do_chain && m->grid_ch && !film_only && !fenerf_siren_backward_fuses_grid(m)) {
      PhaseScope ph(PH_GRID, stream);
      if ((rc = launch_grid_backward(m, npts, pts_c, d_e, d_grid_cl, stream))) return rc;
    }
    if (!do_wgrad) continue;
    // where this chunk's gradients go: the first chunk writes the outputs, later ones a scratch set that is then added (chunk order:
    // the sums of rounds 2-4, bit for bit); FiLM rows of an image's first point range are written in place, later ranges added
    FenerfSirenGrads gc;
    memset(&gc, 0, sizeof(gc));
    float* scratch = (float*)(wb + wsz.scratch);
    float* film_scratch = scratch + (film_only ? 0 : gz.total);
    const bool film_direct = c.s == 0;
    float* fdst[4];
    {
      size_t o = 0;
      for (int k = 0; k < 4; ++k) {
        fdst[k] = film_direct ? f2[k] + (size_t)c.b * frow[k] : film_scratch + o;
        o += (size_t)c.nb * frow[k];
      }
    }
    if (!film_only) {
      if (first) gc = *grads;
      else carve_grads(m, gz, scratch, &gc);
    }
    gc.d_freq_geo = fdst[0]; gc.d_phase_geo = fdst[1]; gc.d_freq_app = fdst[2]; gc.d_phase_app = fdst[3];
    const float* dirs_c = rd ? rd + (size_t)g0 * 3 : nullptr;
    rc = launch_param_grads(m, c.nb, c.n, pts_c, dirs_c, fp_c, pp_c, out_c, dout_c, (const float*)tape_c, tape_e2 ? tape_e2 + (size_t)g0 * 32 : nullptr,
                            film16 ? nullptr : dump, gc, film_only, wb + wsz.wgrad, stream, film_sums, tape_format, weights);
    if (rc) return rc;
    if (!film_direct || (!first && !film_only)) {
      MultiAdd J;
      memset(&J, 0, sizeof(J));
      int k = 0;
      auto add = [&](float* dst, const float* src, long long n) { if (dst && src && n > 0) { J.dst[k] = dst; J.src[k] = src; J.n[k] = n; ++k; } };
      if (!first && !film_only) {
        for (int i = 0; i < ng; ++i) { add(grads->geo_w[i], gc.geo_w[i], gz.geo_w[i]); add(grads->geo_b[i], gc.geo_b[i], gz.geo_b[i]); }
        for (int i = 0; i < nc; ++i) { add(grads->color_w[i], gc.color_w[i], gz.color_w[i]); add(grads->color_b[i], gc.color_b[i], gz.color_b[i]); }
        add(grads->head_w, gc.head_w, gz.head_w); add(grads->head_b, gc.head_b, gz.head_b); add(grads->rgb_w, gc.rgb_w, gz.rgb_w); add(grads->rgb_b, gc.rgb_b, gz.rgb_b);
      }
      if (!film_direct)
        for (int q = 0; q < 4; ++q) add(f2[q] + (size_t)c.b * frow[q], fdst[q], (long long)c.nb * frow[q]);
      J.count = k;
      PhaseScope ph(PH_OTHER, stream);
      if ((rc = launch_multi_add(J, stream))) return rc;
    }
    first = false;
  }
  // ---- the two passes of an image share its FiLM parameters: row b + row B + b
  if (stage != 1) {
    FilmFold F;
    F.B = B;
    F.out[0] = grads->d_freq_geo; F.out[1] = grads->d_phase_geo; F.out[2] = grads->d_freq_app; F.out[3] = grads->d_phase_app;
    for (int q = 0; q < 4; ++q) { F.in[q] = f2[q]; F.row[q] = (int)frow[q]; }
    PhaseScope ph(PH_OTHER, stream);
    if ((rc = launch_film_fold(F, stream))) return rc;
  }
  if (d_grid_cl && stage != 2) {
    PhaseScope ph(PH_GRID, stream);
    if ((rc = launch_grid_unlayout(d_grid_cl, d_grid_ncdhw, m->gd, m->gh, m->gw, stream))) return rc;
  }
  return FENERF_OK;
}

extern "C" int fenerf_render_backward(const FenerfModel* m, int B, int R, int N, int lock_view, const void* save, size_t save_bytes,
                                      int tape_format, const float* z_coarse, const float* noise_final, const FenerfCompositeOpts* opts,
                                      const float* g_rgb, const FenerfSirenGrads* grads, float* d_grid_ncdhw, const FenerfSirenGrads* weights,
                                      int64_t chunk_points, int64_t film_sums_budget_bytes, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  return render_backward_impl(m, B, R, N, lock_view, save, save_bytes, tape_format, z_coarse, noise_final, opts, g_rgb, grads, d_grid_ncdhw, weights,
                              chunk_points, film_sums_budget_bytes, workspace, workspace_bytes, stream, 1, 0);
}

extern "C" size_t fenerf_render_backward_split_workspace_bytes(const FenerfModel* m, int B, int R, int N, int64_t chunk_points, int keep_chunks) {
  if (!m || B <= 0 || R <= 0 || N <= 0 || keep_chunks < 1) return 0;
  return backward_ws(m, B, R, N, 0, chunk_points, 0, keep_chunks).total;
}

extern "C" int fenerf_render_backward_stage(const FenerfModel* m, int stage, int keep_chunks, int B, int R, int N, int lock_view, const void* save,
                                            size_t save_bytes, int tape_format, const float* z_coarse, const float* noise_final,
                                            const FenerfCompositeOpts* opts, const float* g_rgb, const FenerfSirenGrads* grads, float* d_grid_ncdhw,
                                            const FenerfSirenGrads* weights, int64_t chunk_points, void* workspace, size_t workspace_bytes,
                                            void* stream) {
  if (stage != 1 && stage != 2) return fail(FENERF_E_INVALID, "stage must be 1 (render stage) or 2 (weight stage)");
  return render_backward_impl(m, B, R, N, lock_view, save, save_bytes, tape_format, z_coarse, noise_final, opts, g_rgb, grads, d_grid_ncdhw, weights,
                              chunk_points, 0, workspace, workspace_bytes, stream, keep_chunks, stage);
}
