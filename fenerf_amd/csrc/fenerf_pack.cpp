// Host-side weight packer: nn.Linear-layout fp32 weights -> the SIREN kernel's streaming layout
// (fenerf_layout.h).  Pure C++ (no HIP), so layout tests can run without a GPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fenerf.h"
#include "fenerf_internal.h"
#include "fenerf_layout.h"

namespace fenerf {

// One k-step of a body: the source column each lane-half multiplies (-1 = zero).
struct KStep { int col_h0, col_h1; };

// Emits one body (one 32-row n-block): rows r0..r0+31 of W (row-major [nrows][ncols], rows >= nrows are
// zero), k-steps as listed, padded to whole entries and to a multiple of PF entries.
static void emit_body(std::vector<float>& out, const double* W, int nrows, int ncols, int r0,
                      const std::vector<KStep>& ks, int padded_entries) {
  const int n_entries = ((int)ks.size() + 3) / 4;
  for (int e = 0; e < padded_entries; ++e) {
    for (int lane = 0; lane < 64; ++lane) {
      const int row = r0 + (lane & 31), h = lane >> 5;
      for (int j = 0; j < 4; ++j) {
        float v = 0.f;
        const int s = e * 4 + j;
        if (e < n_entries && s < (int)ks.size() && row < nrows) {
          const int col = h ? ks[s].col_h1 : ks[s].col_h0;
          if (col >= 0) v = (float)W[(size_t)row * ncols + col];
        }
        out.push_back(v);
      }
    }
  }
}

static std::vector<double> to_f64(const float* p, size_t n) {
  std::vector<double> v(n);
  for (size_t i = 0; i < n; ++i) v[i] = p[i];
  return v;
}

// HEAD matrix [32][H]: rows [0, n_lab) = the label head folded to one affine map (2-3 Linear layers with no activation
// between them, siren.py:1490-1494; folded in fp64), row n_lab = sigma.  head_b = the matching biases.
static void fold_head(const FenerfModelDesc* d, std::vector<double>& Wh, std::vector<double>& head_b) {
  const int H = d->hidden_dim, n_lab = d->output_dim - 4;
  Wh.assign((size_t)32 * H, 0.0);
  head_b.assign(32, 0.0);
  if (n_lab > 0) {
    int rows = (d->n_label_layers == 1) ? n_lab : H;
    std::vector<double> A = to_f64(d->label_w[0], (size_t)rows * H), c = to_f64(d->label_b[0], rows);
    for (int i = 1; i < d->n_label_layers; ++i) {
      const int orow = (i == d->n_label_layers - 1) ? n_lab : H;
      auto Wi = to_f64(d->label_w[i], (size_t)orow * rows);
      auto bi = to_f64(d->label_b[i], orow);
      std::vector<double> A2((size_t)orow * H, 0.0), c2(orow, 0.0);
      for (int o = 0; o < orow; ++o) {
        double cb = bi[o];
        for (int k = 0; k < rows; ++k) {
          const double w = Wi[(size_t)o * rows + k];
          cb += w * c[k];
          const double* arow = &A[(size_t)k * H];
          double* drow = &A2[(size_t)o * H];
          for (int x = 0; x < H; ++x) drow[x] += w * arow[x];
        }
        c2[o] = cb;
      }
      A.swap(A2); c.swap(c2); rows = orow;
    }
    for (int o = 0; o < n_lab; ++o) {
      for (int x = 0; x < H; ++x) Wh[(size_t)o * H + x] = A[(size_t)o * H + x];
      head_b[o] = c[o];
    }
  }
  for (int x = 0; x < H; ++x) Wh[(size_t)n_lab * H + x] = d->sigma_w[x];
  head_b[n_lab] = d->sigma_b[0];
}

int validate_desc(const FenerfModelDesc* d, std::string& err) {
  if (!d) { err = "desc is NULL"; return FENERF_E_INVALID; }
  if (d->abi_version != FENERF_ABI_VERSION) { err = "abi_version mismatch"; return FENERF_E_INVALID; }
  const int H = d->hidden_dim;
  if (!(H == 32 || H == 64 || H == 96 || H == 128 || H == 192 || H == 256)) { err = "hidden_dim must be 32, 64, 96, 128, 192 or 256"; return FENERF_E_UNSUPPORTED; }
  if (d->n_geo < 2 || d->n_geo > FENERF_MAX_GEO) { err = "n_geo out of range"; return FENERF_E_INVALID; }
  if (d->n_color < 1 || d->n_color > FENERF_MAX_COLOR) { err = "n_color out of range"; return FENERF_E_INVALID; }
  if (d->n_label_layers < 0 || d->n_label_layers > FENERF_MAX_LABEL_LAYERS) { err = "n_label_layers out of range"; return FENERF_E_INVALID; }
  const int n_lab = d->output_dim - 4;
  if (n_lab < 0 || n_lab + 1 > 32) { err = "output_dim must be in [4, 35]"; return FENERF_E_INVALID; }
  if ((n_lab > 0) != (d->n_label_layers > 0)) { err = "label layers and output_dim disagree"; return FENERF_E_INVALID; }
  if (!(d->grid_ch == 0 || d->grid_ch == 32)) { err = "grid_ch must be 0 or 32"; return FENERF_E_UNSUPPORTED; }
  if (d->grid_ch && (d->grid_d < 2 || d->grid_h < 2 || d->grid_w < 2)) { err = "grid dims must be >= 2"; return FENERF_E_INVALID; }
  for (int i = 0; i < d->n_geo; ++i) if (!d->geo_w[i] || !d->geo_b[i]) { err = "geo weight pointer is NULL"; return FENERF_E_INVALID; }
  for (int i = 0; i < d->n_color; ++i) if (!d->color_w[i] || !d->color_b[i]) { err = "color weight pointer is NULL"; return FENERF_E_INVALID; }
  for (int i = 0; i < d->n_label_layers; ++i) if (!d->label_w[i] || !d->label_b[i]) { err = "label weight pointer is NULL"; return FENERF_E_INVALID; }
  if (!d->sigma_w || !d->sigma_b || !d->rgb_w || !d->rgb_b) { err = "sigma/rgb weight pointer is NULL"; return FENERF_E_INVALID; }
  return FENERF_OK;
}

int pack_weights(const FenerfModelDesc* d, std::vector<float>& blob, std::vector<float>& consts, std::string& err) {
  int rc = validate_desc(d, err);
  if (rc) return rc;
  const int H = d->hidden_dim;
  const bool grid = d->grid_ch != 0;
  const StreamShape sh = stream_shape(H, d->n_geo, d->n_color, grid);
  const int L = d->n_geo + d->n_color;
  blob.clear();
  blob.reserve((size_t)(sh.l0_entries + sh.ring_entries) * 256);

  // k-steps over an H-wide activation held in MFMA C/D order, optionally shifted by a column offset
  auto x_ksteps = [&](int col_off) {
    std::vector<KStep> ks(H / 2);
    for (int s = 0; s < H / 2; ++s) ks[s] = {col_off + feat_of(s, 0), col_off + feat_of(s, 1)};
    return ks;
  };

  // ---- layer 0 (3 -> H): k-steps (x | y), (z | 0); one entry per n-block, floats 2,3 unused
  {
    auto W0 = to_f64(d->geo_w[0], (size_t)H * 3);
    std::vector<KStep> ks = {{0, 1}, {2, -1}};
    for (int nb = 0; nb < sh.NB; ++nb) emit_body(blob, W0.data(), H, 3, nb * 32, ks, 1);
  }
  // ---- G1..G(n_geo-1)
  for (int l = 1; l < d->n_geo; ++l) {
    auto W = to_f64(d->geo_w[l], (size_t)H * H);
    auto ks = x_ksteps(0);
    for (int nb = 0; nb < sh.NB; ++nb) emit_body(blob, W.data(), H, H, nb * 32, ks, sh.KGXP);
  }
  // ---- C0: reference columns are [dir(3) | grid feats(32) | x(H)]  (siren.py:1522 / :1222)
  {
    const int cin = 3 + d->grid_ch + H;
    auto W = to_f64(d->color_w[0], (size_t)H * cin);
    auto ks = x_ksteps(3 + d->grid_ch);
    if (grid)
      for (int j = 0; j < FENERF_E_KSTEPS; ++j) ks.push_back({3 + j, 3 + 16 + j});  // half h holds channels 16h + j
    ks.push_back({0, 1});   // (dir.x | dir.y)
    ks.push_back({2, -1});  // (dir.z | 0)
    for (int nb = 0; nb < sh.NB; ++nb) emit_body(blob, W.data(), H, cin, nb * 32, ks, sh.c0_kgp);
  }
  // ---- HEAD: rows [0, n_lab) = folded label head, row n_lab = sigma.  The label head is 2-3 Linear layers with
  //      no activation between them (siren.py:1490-1494), i.e. one affine map; fold it in fp64.
  std::vector<double> head_b, Wh;
  fold_head(d, Wh, head_b);
  {
    emit_body(blob, Wh.data(), 32, H, 0, x_ksteps(0), sh.KGXP);
  }
  // ---- C1..
  for (int l = 1; l < d->n_color; ++l) {
    auto W = to_f64(d->color_w[l], (size_t)H * H);
    auto ks = x_ksteps(0);
    for (int nb = 0; nb < sh.NB; ++nb) emit_body(blob, W.data(), H, H, nb * 32, ks, sh.KGXP);
  }
  // ---- RGB (3 rows)
  {
    auto W = to_f64(d->rgb_w, (size_t)3 * H);
    emit_body(blob, W.data(), 3, H, 0, x_ksteps(0), sh.KGXP);
  }
  // ---- tail pad
  blob.insert(blob.end(), (size_t)FENERF_PF * 256, 0.f);
  if (blob.size() != (size_t)(sh.l0_entries + sh.ring_entries) * 256) { err = "internal: stream size mismatch"; return FENERF_E_INVALID; }

  consts.assign((size_t)CONST_FILM_BIAS + (size_t)L * H, 0.f);
  for (int i = 0; i < 32; ++i) consts[CONST_HEAD_BIAS + i] = (float)head_b[i];
  for (int i = 0; i < 3; ++i) consts[CONST_RGB_BIAS + i] = d->rgb_b[i];
  for (int l = 0; l < d->n_geo; ++l) memcpy(&consts[CONST_FILM_BIAS + (size_t)l * H], d->geo_b[l], sizeof(float) * H);
  for (int l = 0; l < d->n_color; ++l) memcpy(&consts[CONST_FILM_BIAS + (size_t)(d->n_geo + l) * H], d->color_b[l], sizeof(float) * H);
  return FENERF_OK;
}

// ------------------------------------------------------------------------------------------------
// SPATIALSIRENGRID in one kernel (fenerf_siren_local.hip): the per-point mapping network and the SIREN interleaved in ONE ring
// stream, in the kernel's consumption order (every body padded to whole ring revolutions):
//   M0: MH/32 bodies of network.0 [MH][32], k-steps (latent 2 s | latent 2 s + 1)
//   M1: MH/32 bodies of network.2 [MH][MH] over h1 in MFMA C/D order
//   per FiLM layer l and n-block nb: F = network.4 rows l H + 32 nb .. (frequencies), P = rows L H + l H + 32 nb .. (phase shifts), both
//       over h2; then Z = the layer's own rows over its input (layer 0: (x | y), (z | 0); first colour layer: x then (dir.x | dir.y), (dir.z | 0))
//   sigma head (one body, row 0) between the trunk and the colour layers; rgb head (rows 0..2) last; tail pad.
// consts: b0 [MH] | b1 [MH] | b2 [2 L H] | FiLM-layer biases [L H] | sigma bias [4] | rgb bias [4].
// ------------------------------------------------------------------------------------------------
int pack_local_weights(const FenerfModelDesc* d, const FenerfLocalMapDesc* mp, std::vector<float>& blob, std::vector<float>& consts,
                       std::string& err) {
  int rc = validate_desc(d, err);
  if (rc) return rc;
  if (!mp) { err = "map desc is NULL"; return FENERF_E_INVALID; }
  if (mp->latent_dim != 32 || mp->map_hidden != 256) { err = "local mapping network must be 32 -> 256 -> 256 -> 2 L H (siren.py:440)"; return FENERF_E_UNSUPPORTED; }
  if (d->grid_ch != 0 || d->n_label_layers != 0 || d->output_dim != 4) { err = "per-point modulation is defined for the rgb + sigma model without feature grid (SPATIALSIRENGRID)"; return FENERF_E_UNSUPPORTED; }
  if (!mp->w0 || !mp->b0 || !mp->w1 || !mp->b1 || !mp->w2 || !mp->b2) { err = "map desc: NULL weight pointer"; return FENERF_E_INVALID; }
  const int H = d->hidden_dim, MH = 256, ZL = 32, NB = H / 32, L = d->n_geo + d->n_color;
  const int KGM = MH / 8, KGXP = pad_pf(H / 8), KGCP = pad_pf(H / 8 + 1);
  auto ksteps = [&](int width, int col_off) {
    std::vector<KStep> ks(width / 2);
    for (int s = 0; s < width / 2; ++s) ks[s] = {col_off + feat_of(s, 0), col_off + feat_of(s, 1)};
    return ks;
  };
  blob.clear();
  {
    auto W = to_f64(mp->w0, (size_t)MH * ZL);
    std::vector<KStep> ks(ZL / 2);
    for (int s = 0; s < ZL / 2; ++s) ks[s] = {2 * s, 2 * s + 1};
    for (int nb = 0; nb < MH / 32; ++nb) emit_body(blob, W.data(), MH, ZL, nb * 32, ks, FENERF_PF);
  }
  {
    auto W = to_f64(mp->w1, (size_t)MH * MH);
    const auto ks = ksteps(MH, 0);
    for (int nb = 0; nb < MH / 32; ++nb) emit_body(blob, W.data(), MH, MH, nb * 32, ks, KGM);
  }
  const auto W2 = to_f64(mp->w2, (size_t)2 * L * H * MH);
  const auto ksm = ksteps(MH, 0);
  auto film_bodies = [&](int l, int nb) {
    emit_body(blob, W2.data(), 2 * L * H, MH, l * H + nb * 32, ksm, KGM);            // frequencies of (l, nb)
    emit_body(blob, W2.data(), 2 * L * H, MH, L * H + l * H + nb * 32, ksm, KGM);    // phase shifts
  };
  // rows r0 .. r0 + 31 of a matrix whose row count is r0 + 32 at most: emit_body zero-fills rows >= nrows, so pass the true count
  {
    auto W = to_f64(d->geo_w[0], (size_t)H * 3);
    const std::vector<KStep> ks = {{0, 1}, {2, -1}};
    for (int nb = 0; nb < NB; ++nb) { film_bodies(0, nb); emit_body(blob, W.data(), H, 3, nb * 32, ks, FENERF_PF); }
  }
  for (int l = 1; l < d->n_geo; ++l) {
    auto W = to_f64(d->geo_w[l], (size_t)H * H);
    const auto ks = ksteps(H, 0);
    for (int nb = 0; nb < NB; ++nb) { film_bodies(l, nb); emit_body(blob, W.data(), H, H, nb * 32, ks, KGXP); }
  }
  {
    std::vector<double> Wh((size_t)32 * H, 0.0);
    for (int x = 0; x < H; ++x) Wh[x] = d->sigma_w[x];
    emit_body(blob, Wh.data(), 32, H, 0, ksteps(H, 0), KGXP);
  }
  for (int c = 0; c < d->n_color; ++c) {
    const int cin = c == 0 ? 3 + H : H;
    auto W = to_f64(d->color_w[c], (size_t)H * cin);
    auto ks = ksteps(H, c == 0 ? 3 : 0);
    if (c == 0) { ks.push_back({0, 1}); ks.push_back({2, -1}); }
    for (int nb = 0; nb < NB; ++nb) { film_bodies(d->n_geo + c, nb); emit_body(blob, W.data(), H, cin, nb * 32, ks, c == 0 ? KGCP : KGXP); }
  }
  {
    auto W = to_f64(d->rgb_w, (size_t)3 * H);
    emit_body(blob, W.data(), 3, H, 0, ksteps(H, 0), KGXP);
  }
  blob.insert(blob.end(), (size_t)FENERF_PF * 256, 0.f);
  consts.assign((size_t)2 * MH + (size_t)3 * L * H + 8, 0.f);
  memcpy(&consts[0], mp->b0, sizeof(float) * MH);
  memcpy(&consts[MH], mp->b1, sizeof(float) * MH);
  memcpy(&consts[2 * MH], mp->b2, sizeof(float) * 2 * L * H);
  float* fb = &consts[(size_t)2 * MH + (size_t)2 * L * H];
  for (int l = 0; l < d->n_geo; ++l) memcpy(fb + (size_t)l * H, d->geo_b[l], sizeof(float) * H);
  for (int c = 0; c < d->n_color; ++c) memcpy(fb + (size_t)(d->n_geo + c) * H, d->color_b[c], sizeof(float) * H);
  float* hb = fb + (size_t)L * H;
  hb[0] = d->sigma_b[0];
  for (int i = 0; i < 3; ++i) hb[4 + i] = d->rgb_b[i];
  return FENERF_OK;
}

// ------------------------------------------------------------------------------------------------
// f16x3 packing (fenerf_layout.h "f16x3 mode")
// ------------------------------------------------------------------------------------------------
struct KStep16 { int col[2][8]; };   // source column per lane-half and slot, -1 = zero

static inline uint16_t f16_bits(_Float16 v) { uint16_t b; memcpy(&b, &v, 2); return b; }

// Index-map mode (fenerf_pack_index_map_f16): the desc's weights hold (1 + flat index) of themselves; instead of fp16
// halves the packer records, per half of the stream, that index | (is_lo << 30) (0 = padding) and applies no row scales.
// The caller then builds the stream on the device: scale rows, split hi / lo, gather.
static std::vector<int32_t>* g_tags = nullptr;
static std::mutex g_tags_mutex;

// row_scale[row] (power of two) is applied before the hi/lo split; rows >= nrows are zero.
static void emit_body16(std::vector<uint16_t>& out, const double* W, int nrows, int ncols, int r0,
                        const std::vector<KStep16>& ks, int padded_entries, const std::vector<double>& row_scale) {
  const int real = 2 * (int)ks.size();
  for (int e = 0; e < padded_entries; ++e) {
    const int s = e >> 1, lo = e & 1;
    for (int lane = 0; lane < 64; ++lane) {
      const int row = r0 + (lane & 31), h = lane >> 5;
      for (int t = 0; t < 8; ++t) {
        uint16_t bits = 0;
        int32_t code = 0;
        if (e < real && row < nrows) {
          const int col = ks[s].col[h][t];
          if (col >= 0) {
            const float w = (float)(W[(size_t)row * ncols + col] * row_scale[row]);   // exact: power-of-two scale
            const _Float16 hi = (_Float16)w;
            bits = lo ? f16_bits((_Float16)(w - (float)hi)) : f16_bits(hi);
            if (g_tags) code = (int32_t)w | (lo << 30);
          }
        }
        out.push_back(bits);
        if (g_tags) g_tags->push_back(code);
      }
    }
  }
}

// power of two s with max|row| * s in [0.5, 1)  (1 for an all-zero row)
static std::vector<double> row_scales(const double* W, int nrows, int ncols) {
  std::vector<double> sc(nrows, 1.0);
  if (g_tags) return sc;   // index-map mode: tags must pass through unscaled
  for (int r = 0; r < nrows; ++r) {
    double m = 0;
    for (int c = 0; c < ncols; ++c) m = std::fmax(m, std::fabs((double)(float)W[(size_t)r * ncols + c]));
    if (m > 0) { int ex; std::frexp(m, &ex); sc[r] = std::ldexp(1.0, -ex); }
  }
  return sc;
}

int pack_weights_f16(const FenerfModelDesc* d, std::vector<float>& blob, std::vector<float>& consts, std::string& err) {
  int rc = validate_desc(d, err);
  if (rc) return rc;
  const int H = d->hidden_dim;
  const bool grid = d->grid_ch != 0;
  const StreamShape16 sh = stream_shape16(H, d->n_geo, d->n_color, grid);
  const int L = d->n_geo + d->n_color;
  std::vector<float> l0;          // fp32 layer-0 block, identical to the f32 mode
  {
    auto W0 = to_f64(d->geo_w[0], (size_t)H * 3);
    std::vector<KStep> ks = {{0, 1}, {2, -1}};
    for (int nb = 0; nb < sh.NB; ++nb) emit_body(l0, W0.data(), H, 3, nb * 32, ks, 1);
  }
  std::vector<uint16_t> ring;
  ring.reserve((size_t)sh.ring_entries * 512);
  std::vector<float> inv_scale((size_t)L * H, 1.f), head_inv(32, 1.f), rgb_inv(4, 1.f);

  auto x_ksteps = [&](int col_off) {
    std::vector<KStep16> ks(H / 16);
    for (int s = 0; s < H / 16; ++s)
      for (int h = 0; h < 2; ++h)
        for (int t = 0; t < 8; ++t) ks[s].col[h][t] = col_off + feat16_of(s, h, t);
    return ks;
  };
  const double act = (double)F16_ACT_SCALE;
  auto pad_stage_to = [&](size_t stage_begin_halves, int stage_entries) {   // zero entries up to the padded stage size
    const size_t want = stage_begin_halves + (size_t)stage_entries * 512;
    if (ring.size() < want) ring.insert(ring.end(), want - ring.size(), 0);
    if (g_tags) g_tags->resize(ring.size(), 0);
  };
  for (int l = 1; l < d->n_geo; ++l) {
    const size_t begin = ring.size();
    auto W = to_f64(d->geo_w[l], (size_t)H * H);
    auto sc = row_scales(W.data(), H, H);
    for (int n = 0; n < H; ++n) inv_scale[(size_t)l * H + n] = (float)(1.0 / (sc[n] * act));
    auto ks = x_ksteps(0);
    for (int nb = 0; nb < sh.NB; ++nb) emit_body16(ring, W.data(), H, H, nb * 32, ks, sh.body_ep, sc);
    pad_stage_to(begin, sh.sq_stage_e);
  }
  {  // C0: [x | grid feats | dir]; lane-half h holds grid channels 16h..16h+15: k-step j slot t <-> channel 16h + 8j + t
    const int cin = 3 + d->grid_ch + H;
    auto W = to_f64(d->color_w[0], (size_t)H * cin);
    auto sc = row_scales(W.data(), H, cin);
    for (int n = 0; n < H; ++n) inv_scale[(size_t)d->n_geo * H + n] = (float)(1.0 / (sc[n] * act));
    auto ks = x_ksteps(3 + d->grid_ch);
    if (grid)
      for (int j = 0; j < 2; ++j) {
        KStep16 k;
        for (int h = 0; h < 2; ++h) for (int t = 0; t < 8; ++t) k.col[h][t] = 3 + 16 * h + 8 * j + t;
        ks.push_back(k);
      }
    KStep16 kd;
    for (int h = 0; h < 2; ++h) for (int t = 0; t < 8; ++t) kd.col[h][t] = (h == 0 && t < 3) ? t : -1;
    ks.push_back(kd);
    const size_t begin = ring.size();
    for (int nb = 0; nb < sh.NB; ++nb) emit_body16(ring, W.data(), H, cin, nb * 32, ks, sh.c0_ep, sc);
    pad_stage_to(begin, sh.c0_stage_e);
  }
  std::vector<double> head_b, Wh;
  fold_head(d, Wh, head_b);
  {  // HEAD: folded label rows + sigma row, per-row scaled (label and sigma magnitudes differ by orders)
    auto sc = row_scales(Wh.data(), 32, H);
    for (int r = 0; r < 32; ++r) head_inv[r] = (float)(1.0 / (sc[r] * act));
    const size_t begin = ring.size();
    emit_body16(ring, Wh.data(), 32, H, 0, x_ksteps(0), sh.body_ep, sc);
    pad_stage_to(begin, sh.head_stage_e);
  }
  for (int l = 1; l < d->n_color; ++l) {
    const size_t begin = ring.size();
    auto W = to_f64(d->color_w[l], (size_t)H * H);
    auto sc = row_scales(W.data(), H, H);
    for (int n = 0; n < H; ++n) inv_scale[(size_t)(d->n_geo + l) * H + n] = (float)(1.0 / (sc[n] * act));
    auto ks = x_ksteps(0);
    for (int nb = 0; nb < sh.NB; ++nb) emit_body16(ring, W.data(), H, H, nb * 32, ks, sh.body_ep, sc);
    pad_stage_to(begin, sh.sq_stage_e);
  }
  {
    const size_t begin = ring.size();
    auto W = to_f64(d->rgb_w, (size_t)3 * H);
    auto sc = row_scales(W.data(), 3, H);
    for (int r = 0; r < 3; ++r) rgb_inv[r] = (float)(1.0 / (sc[r] * act));
    emit_body16(ring, W.data(), 3, H, 0, x_ksteps(0), sh.body_ep, sc);
    pad_stage_to(begin, sh.head_stage_e);
  }
  if (ring.size() != (size_t)sh.tile_entries * 512) { err = "internal: f16 stream size mismatch"; return FENERF_E_INVALID; }
  // replicated head: the prefetch of the next tile's first chunks reads past the end instead of wrapping
  ring.insert(ring.end(), ring.begin(), ring.begin() + (size_t)FENERF_DPF * FENERF_CH * 512);
  if (g_tags) g_tags->insert(g_tags->end(), g_tags->begin(), g_tags->begin() + (size_t)FENERF_DPF * FENERF_CH * 512);
  blob = l0;
  const size_t off = blob.size();
  blob.resize(off + ring.size() / 2);
  memcpy(&blob[off], ring.data(), ring.size() * 2);

  consts.assign((size_t)CONST_FILM_BIAS + (size_t)2 * L * H + 36, 0.f);
  for (int i = 0; i < 32; ++i) consts[CONST_HEAD_BIAS + i] = (float)head_b[i];
  for (int i = 0; i < 3; ++i) consts[CONST_RGB_BIAS + i] = d->rgb_b[i];
  for (int l = 0; l < d->n_geo; ++l) memcpy(&consts[CONST_FILM_BIAS + (size_t)l * H], d->geo_b[l], sizeof(float) * H);
  for (int l = 0; l < d->n_color; ++l) memcpy(&consts[CONST_FILM_BIAS + (size_t)(d->n_geo + l) * H], d->color_b[l], sizeof(float) * H);
  float* sc_out = &consts[CONST_FILM_BIAS + (size_t)L * H];
  memcpy(sc_out, inv_scale.data(), sizeof(float) * (size_t)L * H);
  memcpy(sc_out + (size_t)L * H, head_inv.data(), sizeof(float) * 32);
  memcpy(sc_out + (size_t)L * H + 32, rgb_inv.data(), sizeof(float) * 4);
  return FENERF_OK;
}

// ------------------------------------------------------------------------------------------------
// backward-chain stream (fenerf_layout.h "Backward chain"): the forward format applied to transposed matrices
// ------------------------------------------------------------------------------------------------
int pack_weights_bwd(const FenerfModelDesc* d, std::vector<float>& blob, std::string& err) {
  int rc = validate_desc(d, err);
  if (rc) return rc;
  const int H = d->hidden_dim;
  const bool grid = d->grid_ch != 0;
  const BwdShape sh = bwd_stream_shape(H, d->n_geo, d->n_color, grid);
  blob.clear();
  blob.reserve((size_t)(sh.ht_entries + sh.ring_entries) * 256);
  auto x_ksteps = [&](int col_off) {
    std::vector<KStep> ks(H / 2);
    for (int s = 0; s < H / 2; ++s) ks[s] = {col_off + feat_of(s, 0), col_off + feat_of(s, 1)};
    return ks;
  };
  // FENERF_PREC_F16X3: the forward evaluates layer l with rows scaled by s_i = 2^e_i * 16 (activations x16, weights 2^e_i),
  // the FiLM frequencies f'' absorb 1/s_i, and the chain kernel forms dz' = dtheta * 2 pi f'' = dz / s_i.  Scaling row i of
  // the backward weights by the same (exact, power-of-two) s_i makes W'^T dz' = W^T dz with the kernel unchanged.
  const bool scaled = d->precision == FENERF_PREC_F16X3;
  auto film_row_scale = [&](const float* W, int rows, int cols) {
    std::vector<double> sc(rows, 1.0);
    if (scaled) {
      auto W64 = to_f64(W, (size_t)rows * cols);
      sc = row_scales(W64.data(), rows, cols);
      for (auto& v : sc) v *= (double)F16_ACT_SCALE;
    }
    return sc;
  };
  auto transposed = [&](const float* W, int rows, int cols, int col0, int ncol, const std::vector<double>* sc = nullptr) {
    std::vector<double> T((size_t)ncol * rows);   // -> [ncol][rows] = (diag(sc) W)[:, col0:col0+ncol]^T
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < ncol; ++c) T[(size_t)c * rows + r] = (double)W[(size_t)r * cols + col0 + c] * (sc ? (*sc)[r] : 1.0);
    return T;
  };
  {  // rgb head^T: [H][3], k-steps (d_r | d_g), (d_b | 0)
    auto T = transposed(d->rgb_w, 3, H, 0, H);
    std::vector<KStep> ks = {{0, 1}, {2, -1}};
    for (int nb = 0; nb < sh.NB; ++nb) emit_body(blob, T.data(), H, 3, nb * 32, ks, 1);
  }
  for (int l = d->n_color - 1; l >= 1; --l) {
    const auto sc = film_row_scale(d->color_w[l], H, H);
    auto T = transposed(d->color_w[l], H, H, 0, H, &sc);
    auto ks = x_ksteps(0);
    for (int nb = 0; nb < sh.NB; ++nb) emit_body(blob, T.data(), H, H, nb * 32, ks, sh.KGXP);
  }
  {  // colour layer 0 (reference columns [dir(3) | grid(32) | x(H)], siren.py:1522) + the heads that read the trunk
    const int cin = 3 + d->grid_ch + H;
    std::vector<double> Wh, hb;
    fold_head(d, Wh, hb);
    const auto sc0 = film_row_scale(d->color_w[0], H, cin);
    std::vector<double> M((size_t)H * (H + 32), 0.0);   // row j: [W_c0[:, x_j] (H) | head[:, j] (32)]
    for (int j = 0; j < H; ++j) {
      for (int i = 0; i < H; ++i) M[(size_t)j * (H + 32) + i] = (double)d->color_w[0][(size_t)i * cin + 3 + d->grid_ch + j] * sc0[i];
      for (int r = 0; r < 32; ++r) M[(size_t)j * (H + 32) + H + r] = Wh[(size_t)r * H + j];
    }
    auto ks = x_ksteps(0);
    for (int s = 0; s < FENERF_HEAD_KSTEPS; ++s) ks.push_back({H + s, H + 16 + s});
    for (int nb = 0; nb < sh.NB; ++nb) emit_body(blob, M.data(), H, H + 32, nb * 32, ks, sh.c0_kgp);
    if (grid) {
      auto T = transposed(d->color_w[0], H, cin, 3, 32, &sc0);   // [32][H]
      emit_body(blob, T.data(), 32, H, 0, x_ksteps(0), sh.KGXP);
    }
  }
  for (int l = d->n_geo - 1; l >= 1; --l) {
    const auto sc = film_row_scale(d->geo_w[l], H, H);
    auto T = transposed(d->geo_w[l], H, H, 0, H, &sc);
    auto ks = x_ksteps(0);
    for (int nb = 0; nb < sh.NB; ++nb) emit_body(blob, T.data(), H, H, nb * 32, ks, sh.KGXP);
  }
  blob.insert(blob.end(), (size_t)FENERF_PF * 256, 0.f);
  if (blob.size() != (size_t)(sh.ht_entries + sh.ring_entries) * 256) { err = "internal: backward stream size mismatch"; return FENERF_E_INVALID; }
  return FENERF_OK;
}

// ------------------------------------------------------------------------------------------------
// bf16x3 backward-chain stream (fenerf_layout.h "bf16x3 backward chain")
// ------------------------------------------------------------------------------------------------
static inline uint16_t bf16_rne(float v) {
  uint32_t b; memcpy(&b, &v, 4);
  return (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
}
static inline float bf16_f32(uint16_t h) { uint32_t b = (uint32_t)h << 16; float v; memcpy(&v, &b, 4); return v; }

// index != nullptr: record the source VALUE of every half as an integer (index-map mode, weights hold 1 + flat index)
static void emit_body_bf16(std::vector<uint16_t>& out, std::vector<int32_t>* index, const double* W, int nrows, int ncols, int r0,
                           const std::vector<KStep16>& ks, int padded_entries) {
  const int real = 2 * (int)ks.size();
  for (int e = 0; e < padded_entries; ++e) {
    const int s = e >> 1, lo = e & 1;
    for (int lane = 0; lane < 64; ++lane) {
      const int row = r0 + (lane & 31), h = lane >> 5;
      for (int t = 0; t < 8; ++t) {
        uint16_t bits = 0;
        int32_t code = 0;
        if (e < real && row < nrows) {
          const int col = ks[s].col[h][t];
          if (col >= 0) {
            const float w = (float)W[(size_t)row * ncols + col];
            const uint16_t hi = bf16_rne(w);
            bits = lo ? bf16_rne(w - bf16_f32(hi)) : hi;
            code = (int32_t)w;
          }
        }
        out.push_back(bits);
        if (index) index->push_back(code);
      }
    }
  }
}

int pack_weights_bwd16(const FenerfModelDesc* d, std::vector<float>& blob, std::string& err, std::vector<int32_t>* index) {
  int rc = validate_desc(d, err);
  if (rc) return rc;
  const int H = d->hidden_dim;
  const bool grid = d->grid_ch != 0;
  const BwdShape16 sh = bwd_stream_shape16(H, d->n_geo, d->n_color, grid);
  blob.clear();
  std::vector<uint16_t> ring;
  ring.reserve((size_t)sh.ring_entries * 512);
  auto x_ksteps = [&]() {
    std::vector<KStep16> ks(sh.KS16);
    for (int s = 0; s < sh.KS16; ++s)
      for (int h = 0; h < 2; ++h)
        for (int t = 0; t < 8; ++t) ks[s].col[h][t] = feat16_of(s, h, t);
    return ks;
  };
  // rows scaled like the forward's f16x3 layers (see pack_weights_bwd); unscaled in index-map mode
  auto film_row_scale = [&](const float* W, int rows, int cols) {
    std::vector<double> sc(rows, 1.0);
    if (!index) {
      auto W64 = to_f64(W, (size_t)rows * cols);
      sc = row_scales(W64.data(), rows, cols);
      for (auto& v : sc) v *= (double)F16_ACT_SCALE;
    }
    return sc;
  };
  auto transposed = [&](const float* W, int rows, int cols, int col0, int ncol, const std::vector<double>* sc = nullptr) {
    std::vector<double> T((size_t)ncol * rows);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < ncol; ++c) T[(size_t)c * rows + r] = (double)W[(size_t)r * cols + col0 + c] * (sc ? (*sc)[r] : 1.0);
    return T;
  };
  {  // rgb head^T stays on the fp32 MFMA (two k-steps)
    auto T = transposed(d->rgb_w, 3, H, 0, H);
    std::vector<KStep> ks = {{0, 1}, {2, -1}};
    for (int nb = 0; nb < sh.NB; ++nb) emit_body(blob, T.data(), H, 3, nb * 32, ks, 1);
  }
  for (int l = d->n_color - 1; l >= 1; --l) {
    const auto sc = film_row_scale(d->color_w[l], H, H);
    auto T = transposed(d->color_w[l], H, H, 0, H, &sc);
    for (int nb = 0; nb < sh.NB; ++nb) emit_body_bf16(ring, index, T.data(), H, H, nb * 32, x_ksteps(), sh.body_ep);
  }
  {
    const int cin = 3 + d->grid_ch + H;
    std::vector<double> Wh, hb;
    fold_head(d, Wh, hb);
    const auto sc0 = film_row_scale(d->color_w[0], H, cin);
    std::vector<double> M((size_t)H * (H + 32), 0.0);   // row j: [W_c0[:, x_j] (H) | head[:, j] (32)]
    for (int j = 0; j < H; ++j) {
      for (int i = 0; i < H; ++i) M[(size_t)j * (H + 32) + i] = (double)d->color_w[0][(size_t)i * cin + 3 + d->grid_ch + j] * sc0[i];
      for (int r = 0; r < 32; ++r) M[(size_t)j * (H + 32) + H + r] = Wh[(size_t)r * H + j];
    }
    auto ks = x_ksteps();
    for (int s = 0; s < 2; ++s) {
      KStep16 k;
      for (int h = 0; h < 2; ++h)
        for (int t = 0; t < 8; ++t) k.col[h][t] = H + 16 * s + 8 * h + t;
      ks.push_back(k);
    }
    for (int nb = 0; nb < sh.NB; ++nb) emit_body_bf16(ring, index, M.data(), H, H + 32, nb * 32, ks, sh.c0_ep);
    if (grid) {
      auto T = transposed(d->color_w[0], H, cin, 3, 32, &sc0);   // [32][H]
      emit_body_bf16(ring, index, T.data(), 32, H, 0, x_ksteps(), sh.body_ep);
    }
  }
  for (int l = d->n_geo - 1; l >= 1; --l) {
    const auto sc = film_row_scale(d->geo_w[l], H, H);
    auto T = transposed(d->geo_w[l], H, H, 0, H, &sc);
    for (int nb = 0; nb < sh.NB; ++nb) emit_body_bf16(ring, index, T.data(), H, H, nb * 32, x_ksteps(), sh.body_ep);
  }
  ring.insert(ring.end(), (size_t)FENERF_PF16 * 512, 0);
  if (index) index->insert(index->end(), (size_t)FENERF_PF16 * 512, 0);
  if (ring.size() != (size_t)sh.ring_entries * 512) { err = "internal: bf16 backward stream size mismatch"; return FENERF_E_INVALID; }
  const size_t nf = blob.size();
  blob.resize(nf + ring.size() / 2);
  memcpy(blob.data() + nf, ring.data(), ring.size() * 2);
  return FENERF_OK;
}

}  // namespace fenerf

extern "C" int fenerf_pack_weights_host(const FenerfModelDesc* desc, float** blob, size_t* n_floats, float** consts,
                                        size_t* n_consts) {
  std::vector<float> b, c;
  std::string err;
  int rc = (desc && desc->precision == FENERF_PREC_F16X3) ? fenerf::pack_weights_f16(desc, b, c, err)
                                                          : fenerf::pack_weights(desc, b, c, err);
  if (rc) { fenerf::set_error(err); return rc; }
  if (!blob || !n_floats || !consts || !n_consts) { fenerf::set_error("NULL output pointer"); return FENERF_E_INVALID; }
  *blob = (float*)malloc(b.size() * sizeof(float));
  *consts = (float*)malloc(c.size() * sizeof(float));
  if (!*blob || !*consts) {
    free(*blob); free(*consts);
    *blob = *consts = nullptr;
    fenerf::set_error("malloc failed");
    return FENERF_E_NOMEM;
  }
  memcpy(*blob, b.data(), b.size() * sizeof(float));
  memcpy(*consts, c.data(), c.size() * sizeof(float));
  *n_floats = b.size();
  *n_consts = c.size();
  return FENERF_OK;
}

extern "C" int fenerf_pack_local_host(const FenerfModelDesc* desc, const FenerfLocalMapDesc* map, float** blob, size_t* n_floats,
                                      float** consts, size_t* n_consts) {
  std::vector<float> b, c;
  std::string err;
  int rc = fenerf::pack_local_weights(desc, map, b, c, err);
  if (rc) { fenerf::set_error(err); return rc; }
  if (!blob || !n_floats || !consts || !n_consts) { fenerf::set_error("NULL output pointer"); return FENERF_E_INVALID; }
  *blob = (float*)malloc(b.size() * sizeof(float));
  *consts = (float*)malloc(c.size() * sizeof(float));
  if (!*blob || !*consts) {
    free(*blob); free(*consts);
    *blob = *consts = nullptr;
    fenerf::set_error("malloc failed");
    return FENERF_E_NOMEM;
  }
  memcpy(*blob, b.data(), b.size() * sizeof(float));
  memcpy(*consts, c.data(), c.size() * sizeof(float));
  *n_floats = b.size();
  *n_consts = c.size();
  return FENERF_OK;
}

extern "C" int fenerf_pack_index_map_f16(const FenerfModelDesc* desc, int32_t** map, size_t* n) {
  if (!map || !n) { fenerf::set_error("NULL output pointer"); return FENERF_E_INVALID; }
  std::lock_guard<std::mutex> lock(fenerf::g_tags_mutex);
  std::vector<int32_t> tags;
  std::vector<float> b, c;
  std::string err;
  fenerf::g_tags = &tags;
  int rc = fenerf::pack_weights_f16(desc, b, c, err);
  fenerf::g_tags = nullptr;
  if (rc) { fenerf::set_error(err); return rc; }
  *map = (int32_t*)malloc(tags.size() * sizeof(int32_t));
  if (!*map) { fenerf::set_error("malloc failed"); return FENERF_E_NOMEM; }
  memcpy(*map, tags.data(), tags.size() * sizeof(int32_t));
  *n = tags.size();
  return FENERF_OK;
}

extern "C" int fenerf_pack_backward_host(const FenerfModelDesc* desc, float** blob, size_t* n_floats) {
  std::vector<float> b;
  std::string err;
  int rc = (desc && desc->precision == FENERF_PREC_F16X3) ? fenerf::pack_weights_bwd16(desc, b, err, nullptr)
                                                          : fenerf::pack_weights_bwd(desc, b, err);
  if (rc) { fenerf::set_error(err); return rc; }
  if (!blob || !n_floats) { fenerf::set_error("NULL output pointer"); return FENERF_E_INVALID; }
  *blob = (float*)malloc(b.size() * sizeof(float));
  if (!*blob) { fenerf::set_error("malloc failed"); return FENERF_E_NOMEM; }
  memcpy(*blob, b.data(), b.size() * sizeof(float));
  *n_floats = b.size();
  return FENERF_OK;
}

extern "C" int fenerf_pack_backward_index_map_bf16(const FenerfModelDesc* desc, int32_t** map, size_t* n) {
  if (!map || !n) { fenerf::set_error("NULL output pointer"); return FENERF_E_INVALID; }
  std::vector<int32_t> index;
  std::vector<float> b;
  std::string err;
  int rc = fenerf::pack_weights_bwd16(desc, b, err, &index);
  if (rc) { fenerf::set_error(err); return rc; }
  *map = (int32_t*)malloc(index.size() * sizeof(int32_t));
  if (!*map) { fenerf::set_error("malloc failed"); return FENERF_E_NOMEM; }
  memcpy(*map, index.data(), index.size() * sizeof(int32_t));
  *n = index.size();
  return FENERF_OK;
}

extern "C" void fenerf_free_host(void* p) { free(p); }
