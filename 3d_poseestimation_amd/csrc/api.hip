// C ABI of libposelift.so: arena layout, workspace plan and the forward / backward
// launch sequences of the lifter (include/poselift.h).  Host code only; every kernel
// lives in gemm_f32.hip / elementwise.hip.
//
// Layer numbering used everywhere: hidden layer 0 is LinearModel.w1/batch_norm1
// (baselineModel.py:67-68,90-94); residual block s owns hidden layers 1+2s (its w1) and
// 2+2s (its w2) (baselineModel.py:23-27,33-45); "final" is LinearModel.w2 (:77,100).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "adamw.h"
#include "pl_internal.h"

namespace pl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

int check_desc(const PLDesc* d, bool need_arenas) {
  if (!d) PL_FAIL(PL_EINVAL, "descriptor is NULL");
  if (d->in_dim <= 0 || d->out_dim <= 0 || d->hidden <= 0 || d->num_stage < 0)
    PL_FAIL(PL_ESHAPE, "bad dims in=%d hidden=%d out=%d stages=%d", d->in_dim, d->hidden, d->out_dim, d->num_stage);
  if (d->hidden % 4 != 0) PL_FAIL(PL_ESHAPE, "hidden=%d must be a multiple of 4", d->hidden);
  if (d->dtype != PL_F32 && d->dtype != PL_BF16 && d->dtype != PL_BF16X6 && d->dtype != PL_F16X3)
    PL_FAIL(PL_EDTYPE, "dtype %d is not a PLDtype", d->dtype);
  if (!(d->p_dropout >= 0.f && d->p_dropout <= 1.f)) PL_FAIL(PL_EINVAL, "p_dropout=%f outside [0,1]", d->p_dropout);
  if (need_arenas) {
    if (!d->params) PL_FAIL(PL_EINVAL, "params arena is NULL");
    if (reinterpret_cast<uintptr_t>(d->params) & 15) PL_FAIL(PL_EINVAL, "params arena not 16-byte aligned");
    if (d->bn && !d->bn_running) PL_FAIL(PL_EINVAL, "bn_running arena is NULL");
  }
  if (d->sync) {
    const PLSync* y = d->sync;
    if (y->world < 1 || y->rank < 0 || y->rank >= y->world)
      PL_FAIL(PL_EINVAL, "PLSync: rank %d of world %d", y->rank, y->world);
    if (y->world > 1 && !y->gather) PL_FAIL(PL_EINVAL, "PLSync: gather callback is NULL");
  }
  return PL_OK;
}

// cross-rank BatchNorm statistics (PLSync): world 1 = local statistics
inline int sync_world(const PLDesc* d) { return (d->sync && d->bn) ? d->sync->world : 1; }
inline int sync_rank(const PLDesc* d) { return (d->sync && d->bn) ? d->sync->rank : 0; }

// Partial-statistics buffer: [world][2][P][H] floats; rank r's slab holds its P partial rows of the
// first quantity then P rows of the second.  The finalize kernels walk all world*P partials.
int sync_gather(const PLDesc* d, float* base, int64_t floats_per_rank, hipStream_t s) {
  if (sync_world(d) <= 1) return PL_OK;
  const int rc = d->sync->gather(d->sync->user, base, floats_per_rank, (void*)s);
  if (rc != 0) PL_FAIL(PL_ESYNC, "PLSync gather callback failed (%d)", rc);
  return PL_OK;
}

// PL_F16X3: the 1024-wide GEMMs run on fp16 operand planes written by the kernels that produce the tensors
// (gemm_planes.hip).  That path wants whole 128x128 tiles and BatchNorm (its backward pass 1 supplies the range bound
// of dz; under cross-rank statistics every rank's maxima travel in the gathered slab, round 3); anything else runs the
// same fp32-grade arithmetic class on the round-1 kernels (PL_BF16X6: fp32 operands split inside the GEMM) -- never a
// lower precision, never the CPU.
// PL_BF16 takes the same path with ONE bf16 plane per tensor (kind 1): bf16 STORAGE of the GEMM operands
// (activations, dz, weight shadow), no scales -- bf16 has fp32's exponent range.
int tn_splits(int M, int N, int K) {
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int s = 256 / tiles;
  const int kmax = (K + 127) / 128;
  if (s > kmax) s = kmax;
  return s < 1 ? 1 : s;
}

inline int planes_kind(const PLDesc* d, int64_t B) {       // PlaneOut::kind of the operand planes, 0 = not on that path
  const bool ok = d->bn && d->hidden % 128 == 0 && B % 128 == 0 && d->num_stage >= 1 &&
                  B * (int64_t)d->hidden * 4 < (1ll << 30);
  if (!ok) return 0;
  // the weight-gradient GEMM splits K = B into tn_splits slices of whole 32-k tiles (H = 512 with B = 128 * 17 would
  // not): such a batch runs on the round-1 kernels like every other shape off the tile grid
  if (B % (32 * (int64_t)tn_splits(d->hidden, d->hidden, (int)B)) != 0) return 0;
  return d->dtype == PL_F16X3 ? 2 : (d->dtype == PL_BF16 ? 1 : 0);
}
// BatchNorm-backward pass 1 of hidden layer l is folded into the epilogue of the dX GEMM of layer l+1 (which produces
// its incoming gradient) whenever that GEMM is a planes GEMM: every hidden layer but the top one.  A pure function of
// (descriptor, layer), so the ranges of a cut backward agree on it.  POSELIFT_BNR_UNFUSED=1: the separate pass (A/B).
// The top layer's incoming gradient comes from the 51-wide output layer (g = dy W5, skinny.hip): that kernel carries the
// same epilogue (POSELIFT_BNR_TOP_UNFUSED=1: the stand-alone pass for the top layer only).
inline bool fused_reduce(const PLDesc* d, bool planes, int l, int L, bool eval_bn) {
  static const bool off = [] { const char* e = getenv("POSELIFT_BNR_UNFUSED"); return e && e[0] == '1'; }();
  static const bool top_off = [] { const char* e = getenv("POSELIFT_BNR_TOP_UNFUSED"); return e && e[0] == '1'; }();
  if (!planes || !d->bn || eval_bn || off) return false;
  return l < L - 1 || (!top_off && skinny_supported(d->out_dim, d->hidden));
}
// hidden layers of a small local batch off the planes path: one fused launch per BatchNorm direction (elementwise.hip)
inline bool bn_small_ok(const PLDesc* d, bool planes, int64_t B) {
  static const bool off = [] { const char* e = getenv("POSELIFT_BN_SMALL"); return e && e[0] == '0'; }();   // =0: same-box A/B
  return !off && d->bn && !planes && sync_world(d) == 1 && B >= 2 && B <= kBnSmallRows;
}
// ... and behind the first layer the Linear in front of it / the dX GEMM behind it ride in the same launch (small_layer.hip).
// A pure function of (descriptor, batch): forward and backward agree on which bitmap format a layer has.
inline bool small_layer_on(const PLDesc* d, bool planes, int64_t B) {
  return bn_small_ok(d, planes, B) && small_layer_ok((int)B, d->hidden, d->hidden);
}
// ... the first layer's forward likewise (tile-format bitmap for layer 0, too), and the top of the backward pass in one launch
inline bool small_first_on(const PLDesc* d, bool planes, int64_t B) { return small_layer_on(d, planes, B) && small_first_ok(d->in_dim); }
inline bool small_top_on(const PLDesc* d, bool planes, int64_t B) { return small_layer_on(d, planes, B) && small_top_ok(d->out_dim); }
// the fused train step at small batch: no launch for the output Linear (the last hidden layer's launch leaves its slabs)
inline bool small_head_on(const PLDesc* d, bool planes, int64_t B) {
  return small_top_on(d, planes, B) && (d->num_stage > 0 || small_first_on(d, planes, B));
}
inline bool adam_ride_on() {        // POSELIFT_SMALL_ADAM=0: the AdamW step of pl_lifter_train_step as one launch of its own (A/B)
  static const bool off = [] { const char* e = getenv("POSELIFT_SMALL_ADAM"); return e && e[0] == '0'; }();
  return !off;
}
inline int arith_of(const PLDesc* d) { return d->dtype == PL_F16X3 ? (int)PL_BF16X6 : d->dtype; }

struct ParamLayout {
  int L;                       // hidden layers
  std::vector<int64_t> off, numel;
  int64_t total;
};

ParamLayout param_layout(const PLDesc* d) {
  ParamLayout p;
  p.L = 1 + 2 * d->num_stage;
  int64_t o = 0;
  auto add = [&](int64_t n) {
    p.off.push_back(o);
    p.numel.push_back(n);
    o = align_up(o + n, 64);
  };
  for (int l = 0; l < p.L; ++l) {
    const int64_t fan_in = l == 0 ? d->in_dim : d->hidden;
    add((int64_t)d->hidden * fan_in);
    add(d->hidden);
    add(d->hidden);
    add(d->hidden);
  }
  add((int64_t)d->out_dim * d->hidden);
  add(d->out_dim);
  p.total = o;
  return p;
}

// Output layer y = h W^T + b with N = out_dim (51): only ceil(M/128) tiles, so the K = hidden
// contraction is split over workgroups (slabs) until the chip is full, then reduced with the bias.
int out_splits(int M, int N, int K) {
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int s = 256 / tiles;
  const int kmax = K / 128;
  if (s > kmax) s = kmax;
  return s < 1 ? 1 : s;
}

// Workspace plan; every region starts on a 256-byte boundary.
struct Ws {
  int L, G, RC;
  std::vector<size_t> z, act, bits, mean, rstd, dbpart;
  size_t stat, scale, shift, coef, ga, gb, dz, slabs, outpart, dyout, mse, total;
  size_t act_bytes, bits_bytes;
  size_t slab_floats;                 // size of the `slabs` region (also the thin GEMMs' split-K scratch)
  // PL_F16X3 planes path
  bool planes;
  int pkind;                          // PlaneOut::kind: 2 fp16 pair (PL_F16X3), 1 bf16 (PL_BF16)
  std::vector<size_t> actp, wp;       // activation planes of layers 0..L-2, weight planes of layers 1..L-1
  std::vector<size_t> sactp;          // small / ragged batches off the planes path, PL_F16X3: activation planes of layers 0..L-2
                                      // written by the layer kernels' tails (small_layer.hip: the fp16-planes contraction)
  std::vector<char> act_f32;          // is the fp32 activation of layer l materialised?
  size_t dzp, amax, dzscale;
  // partial sums whose combine is deferred to the ONE reduce launch at the end of a backward range
  size_t skp_out, skp_in;             // skinny weight-gradient partials of the output / input layer
  std::vector<size_t> wslab;          // split-K slabs of the 1024-wide weight gradients, one set per layer (planes path)
};

Ws plan(const PLDesc* d, int64_t B) {
  Ws w;
  w.L = 1 + 2 * d->num_stage;
  const int H = d->hidden;
  w.G = gemm_stat_groups((int)B);
  w.RC = bwd_row_chunks((int)B, H);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o = (size_t)align_up((int64_t)(o + bytes), 256);
    return at;
  };
  w.act_bytes = (size_t)B * H * sizeof(float);
  // (at least H words: the tile-format bitmap of the small-batch layer kernels, small_layer.hip)
  w.bits_bytes = std::max((size_t)B * bitmap_words_per_row(H), (size_t)H) * sizeof(uint64_t);
  w.pkind = planes_kind(d, B);
  w.planes = w.pkind != 0;
  for (int l = 0; l < w.L; ++l) {
    // planes path: a layer's output is kept in fp32 only where something reads it as fp32 -- the skip connection
    // (even layers) and the output Linear (last layer); the odd layers feed GEMMs only and exist as planes
    const bool f32 = !w.planes || (l % 2 == 0) || l == w.L - 1;
    w.act_f32.push_back(f32 ? 1 : 0);
    w.z.push_back(take(w.act_bytes));
    w.act.push_back(f32 ? take(w.act_bytes) : 0);
    w.bits.push_back(take(w.bits_bytes));
    w.mean.push_back(take((size_t)H * 4));
    w.rstd.push_back(take((size_t)H * 4));
    w.dbpart.push_back(take((size_t)w.RC * H * 4));
  }
  const int Pmax = std::max(std::max(w.G, skinny_stat_groups((int)B)), w.RC);
  // per rank: two sets of at most Pmax partial rows, and (planes path under SyncBN) the {max|dy|, max|zhat|} pairs of
  // BatchNorm-backward pass 1 behind them
  const size_t amax_pairs = std::max((size_t)((H + 255) / 256) * w.RC, (size_t)(B / 64 + 1) * (H / 32 + 1));
  w.stat = take((size_t)sync_world(d) * ((size_t)2 * Pmax * H + 2 * amax_pairs) * 4);
  w.scale = take((size_t)w.L * H * 4);
  w.shift = take((size_t)w.L * H * 4);
  w.coef = take((size_t)3 * H * 4);
  w.ga = take(w.act_bytes);
  w.gb = take(w.act_bytes);
  w.dz = take(w.act_bytes);
  size_t slab = 0;
  auto need = [&](int M, int N) {
    const int s = tn_splits(M, N, (int)B);
    if (s > 1) slab = std::max(slab, (size_t)s * M * N * 4);
  };
  need(H, d->in_dim);
  need(H, H);
  need(d->out_dim, H);
  {
    const int so = out_splits((int)B, d->out_dim, H);
    if (so > 1) slab = std::max(slab, (size_t)so * B * d->out_dim * 4);
    // partials of the skinny-layer kernels (skinny.hip)
    slab = std::max(slab, (size_t)skinny_chunks((int)B) * std::max(d->in_dim, d->out_dim) * H * 4);
    slab = std::max(slab, skinny_narrow_out_part_floats((int)B, H) * 4);
    if (B <= thin_gemm_max_m() && B % 128) slab = std::max(slab, thin_gemm_scratch_floats((int)B, H, H) * 4);
    if (B <= thin_gemm_max_m()) slab = std::max(slab, (size_t)(H / 16 + 1) * B * 64 * 4);    // output-layer slabs of small_layer.hip
  }
  w.slab_floats = slab / 4;
  w.slabs = take(slab);
  w.outpart = take((size_t)std::max(colsum_chunks((int)B), skinny_in_chunks((int)B)) * d->out_dim * 4);
  w.dyout = take((size_t)B * d->out_dim * 4);                 // d loss / d y of the fused train step
  w.mse = take(std::max(pl_mse_scratch_bytes(B * d->out_dim), (size_t)256));        // (64 partials of launch_small_mse)
  w.dzp = w.amax = w.dzscale = 0;
  w.skp_out = take((size_t)skinny_in_chunks((int)B) * d->out_dim * H * 4);
  w.skp_in = take((size_t)skinny_in_chunks((int)B) * d->in_dim * H * 4);
  if (w.planes) {
    w.wslab.push_back(0);
    for (int l = 1; l < w.L; ++l) w.wslab.push_back(take((size_t)tn_splits(H, H, (int)B) * H * H * 4));
    for (int l = 0; l + 1 < w.L; ++l) w.actp.push_back(take(w.act_bytes));           // two fp16 planes = 4 B per element
    w.wp.push_back(0);
    for (int l = 1; l < w.L; ++l) w.wp.push_back(take((size_t)H * H * 4));
    w.dzp = take(w.act_bytes);
    w.amax = take(std::max((size_t)((H + 255) / 256) * w.RC, (size_t)(B / 64) * (H / 32)) * 2 * 4);
    w.dzscale = take((size_t)w.L * 2 * 4);
  }
  if (d->dtype == PL_F16X3 && B <= thin_gemm_max_m())
    for (int l = 0; l + 1 < w.L; ++l) w.sactp.push_back(take(w.act_bytes));           // two fp16 planes = 4 B per element
  w.total = o;
  return w;
}

// Where BatchNorm-backward pass 1 of one layer puts its output: this rank's slab of the gather buffer -- rc partial rows of
// sum dy, rc of sum dy zhat and, behind them, n_amax {max|dy|, max|zhat|} pairs (PL_F16X3: the range bound of dz).  One
// rank: the sums at the head of the buffer and the maxima in the workspace's own amax region, as before round 3.
struct BnrSlab {
  float *mine, *amax_mine, *amax0;     // this rank's partial sums / maxima; rank 0's maxima (what the finalize kernel walks)
  int64_t floats_per_rank;
  int world;
};
BnrSlab bnr_slab(const PLDesc* d, const Ws& w, void* ws, int rc, int n_amax, bool eval_bn) {
  BnrSlab b;
  float* stat = reinterpret_cast<float*>(static_cast<char*>(ws) + w.stat);
  b.world = eval_bn ? 1 : sync_world(d);
  const int64_t sums = (int64_t)2 * rc * d->hidden;
  if (b.world == 1) {
    b.mine = stat;
    b.amax_mine = b.amax0 = w.amax ? reinterpret_cast<float*>(static_cast<char*>(ws) + w.amax) : nullptr;
    b.floats_per_rank = sums;
    return b;
  }
  b.floats_per_rank = sums + 2 * (int64_t)n_amax;
  b.mine = stat + (size_t)sync_rank(d) * b.floats_per_rank;
  b.amax_mine = b.mine + sums;
  b.amax0 = stat + sums;
  return b;
}

inline bool bn_small(const PLDesc* d, const Ws& w, int64_t B) { return bn_small_ok(d, w.planes, B); }

// Training forward of 128 ... 512 rows on the operand-planes path (fp16 pairs, local statistics): the 1024-wide Linears on the
// layer kernels' contraction (launch_small_linear_stats).  POSELIFT_MID_LINEAR=0: the tile GEMM (same-box A/B).
inline bool mid_linear_on(const PLDesc* d, const Ws& w, int64_t B) {
  static const bool off = [] { const char* e = getenv("POSELIFT_MID_LINEAR"); return e && e[0] == '0'; }();
  // (under SyncBN only where the concatenated batch would come here too: "the shards compute what one process computes on the
  //  concatenated batch, bit for bit" holds because both sides run the same kernel)
  return !off && w.planes && w.pkind == 2 && B * sync_world(d) <= 512 && small_layer_ok(2, d->hidden, d->hidden);
}

// The BatchNorm statistics finalize inside the apply launch (bn_apply_kernel, BnFin): local statistics, <= 4 groups (256
// rows).  Measured same-box, step in ms with / without (POSELIFT_BN_FIN_FUSED=0): B = 96 0.312 / 0.319, 128 0.295 / 0.300,
// 256 0.298 / 0.304 -- and, when tried up to 16 groups, 512 0.353 / 0.351, 1,024 0.419 / 0.391: the dependent prologue in
// every workgroup costs what the 4.9 us launch did as soon as there are more than a few groups (round 2 saw the same at 64).
inline bool fin_in_apply(const PLDesc* d, int groups) {
  static const bool off = [] { const char* e = getenv("POSELIFT_BN_FIN_FUSED"); return e && e[0] == '0'; }();
  return !off && sync_world(d) == 1 && groups >= 1 && groups <= 4 && d->hidden % 4 == 0;
}
inline bool mid_linear_f32_on(const PLDesc* d, int64_t B) {
  static const bool off = [] { const char* e = getenv("POSELIFT_MID_LINEAR"); return e && e[0] == '0'; }();
  return !off && d->bn && d->dtype != PL_BF16 && B > kBnSmallRows && B * sync_world(d) <= 512 &&
         small_layer_ok(2, d->hidden, d->hidden);
}

// PL_F16X3 descriptors: the small-batch layer kernels contract on fp16 planes (three MFMAs per product) instead of exact fp32
// MFMAs -- forward and evaluation; the first layer's launch (which must then be one of them) writes the first planes.
// POSELIFT_SMALL_F16=0: exact fp32 there, as for every other dtype (same-box A/B).
inline bool small_f16_on(const PLDesc* d, const Ws& w) {
  static const bool off = [] { const char* e = getenv("POSELIFT_SMALL_F16"); return e && e[0] == '0'; }();
  return !off && d->dtype == PL_F16X3 && !w.sactp.empty() && small_first_ok(d->in_dim);
}

struct Layer {
  const float *W, *b, *gamma, *beta;
  float *gW, *gb, *ggamma, *gbeta;
  float *rm, *rv;
  int64_t* nbt;
  int K;
};

Layer layer_of(const PLDesc* d, const ParamLayout& pl_, float* grads, int l) {
  Layer y;
  const int H = d->hidden;
  y.K = l == 0 ? d->in_dim : H;
  y.W = d->params + pl_.off[4 * l];
  y.b = d->params + pl_.off[4 * l + 1];
  y.gamma = d->params + pl_.off[4 * l + 2];
  y.beta = d->params + pl_.off[4 * l + 3];
  y.gW = grads ? grads + pl_.off[4 * l] : nullptr;
  y.gb = grads ? grads + pl_.off[4 * l + 1] : nullptr;
  y.ggamma = grads ? grads + pl_.off[4 * l + 2] : nullptr;
  y.gbeta = grads ? grads + pl_.off[4 * l + 3] : nullptr;
  y.rm = d->bn_running ? d->bn_running + (size_t)l * 2 * H : nullptr;
  y.rv = d->bn_running ? d->bn_running + (size_t)l * 2 * H + H : nullptr;
  y.nbt = d->bn_batches ? d->bn_batches + l : nullptr;
  return y;
}

inline float* f32(void* ws, size_t off) { return reinterpret_cast<float*>(static_cast<char*>(ws) + off); }
inline uint64_t* u64(void* ws, size_t off) { return reinterpret_cast<uint64_t*>(static_cast<char*>(ws) + off); }

int check_ws(const Ws& w, void* ws, size_t bytes) {
  if (!ws) PL_FAIL(PL_EWORKSPACE, "workspace is NULL");
  if (reinterpret_cast<uintptr_t>(ws) & 255) PL_FAIL(PL_EWORKSPACE, "workspace not 256-byte aligned");
  if (bytes < w.total) PL_FAIL(PL_EWORKSPACE, "workspace too small: %zu < %zu bytes", bytes, w.total);
  return PL_OK;
}

inline unsigned short* u16(void* ws, size_t off) { return reinterpret_cast<unsigned short*>(static_cast<char*>(ws) + off); }

// operand planes of hidden layer l's weight: in the caller's persistent buffer (PLDesc.wplanes) or in the workspace
inline unsigned short* wplane(const PLDesc* d, const Ws& w, void* ws, int l) {
  if (d->wplanes) {
    const size_t per = (size_t)d->hidden * d->hidden * 2 * (w.pkind == 2 ? 2 : 1);
    return reinterpret_cast<unsigned short*>(static_cast<char*>(d->wplanes) + (size_t)(l - 1) * per);
  }
  return u16(ws, w.wp[l]);
}

// operand planes of every 1024-wide weight matrix (layers 1..L-1), from the fp32 parameter arena -- unless the
// caller keeps them current across calls (PLDesc.wplanes + wplanes_valid: pl_adamw_flat_planes refreshed them)
int split_weight_planes(const PLDesc* d, const ParamLayout& P, const Ws& w, void* ws, hipStream_t s) {
  if (d->wplanes && d->wplanes_valid) return PL_OK;
  if (d->wplanes && (reinterpret_cast<uintptr_t>(d->wplanes) & 15)) PL_FAIL(PL_EINVAL, "wplanes not 16-byte aligned");
  const int64_t n = (int64_t)d->hidden * d->hidden;
  for (int l = 1; l < w.L; ++l) {
    unsigned short* q = wplane(d, w, ws, l);
    PlaneOut po = {q, q + n, kWeightPlaneScale, nullptr, w.pkind};
    PL_TRY(launch_split_planes(d->params + P.off[4 * l], n, po, s));
  }
  return PL_OK;
}

PlanesGemmArgs planes_args(int pkind, const unsigned short* A, int64_t a_plane, int lda, const unsigned short* Bm,
                           int64_t b_plane, int ldb, float* C, int M, int N, int K, float out_scale,
                           const float* dyn_inv) {
  PlanesGemmArgs g = {};
  g.A = A; g.B = Bm; g.a_plane = a_plane; g.b_plane = b_plane; g.lda = lda; g.ldb = ldb;
  g.mode = pkind == 2 ? 2 : 0;                    // plp::kF16x3 / plp::kBf16
  g.out_scale = pkind == 2 ? out_scale : 1.0f;
  g.dyn_inv = pkind == 2 ? dyn_inv : nullptr;
  g.e.C = C; g.e.M = M; g.e.N = N; g.e.K = K; g.e.ldc = N; g.e.split_k = 1;
  return g;
}

int gemm_out_layer(const float* h, const float* W, const float* bias, float* y, int M, int N, int K,
                   float* slabs, hipStream_t s) {
  if (skinny_narrow_out_supported(K, N)) return launch_skinny_narrow_out(h, W, bias, y, M, K, N, slabs, s);
  GemmArgs g = {};
  g.A = h; g.B = W; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
  const int splits = out_splits(M, N, K);
  if (splits > 1) {
    g.C = slabs; g.split_k = splits;
    PL_TRY(launch_gemm_f32(kNT, g, s));
    return launch_reduce_slabs_bias(slabs, splits, M, N, bias, y, s);
  }
  g.C = y; g.bias = bias; g.split_k = 1;
  return launch_gemm_f32(kNT, g, s);
}

// C[M][N] (+slab reduce) = A^T B with A [K][M], B [K][N]
int gemm_tn_reduced(const float* A, int lda, const float* Bm, int ldb, float* C, int M, int N, int K,
                    float* slabs, hipStream_t s) {
  GemmArgs g = {};
  g.A = A; g.B = Bm; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = N;
  const int splits = tn_splits(M, N, K);
  if (splits > 1) {
    g.C = slabs; g.split_k = splits;
    PL_TRY(launch_gemm_f32(kTN, g, s));
    return launch_reduce_slabs(slabs, splits, (int64_t)M * N, C, s);
  }
  g.C = C; g.split_k = 1;
  return launch_gemm_f32(kTN, g, s);
}

}  // namespace
}  // namespace pl

using namespace pl;

extern "C" int pl_version(void) { return PL_VERSION; }
extern "C" const char* pl_last_error(void) { return g_err; }

extern "C" int64_t pl_num_hidden(const PLDesc* d) { return d ? 1 + 2 * (int64_t)d->num_stage : PL_EINVAL; }
extern "C" int64_t pl_param_tensors(const PLDesc* d) { return d ? 4 * (1 + 2 * (int64_t)d->num_stage) + 2 : PL_EINVAL; }
extern "C" int64_t pl_param_offset(const PLDesc* d, int64_t i) {
  if (check_desc(d, false) != PL_OK) return PL_EINVAL;
  const ParamLayout p = param_layout(d);
  if (i < 0 || i >= (int64_t)p.off.size()) { set_error("tensor index %lld out of range", (long long)i); return PL_EINVAL; }
  return p.off[i];
}
extern "C" int64_t pl_param_numel(const PLDesc* d, int64_t i) {
  if (check_desc(d, false) != PL_OK) return PL_EINVAL;
  const ParamLayout p = param_layout(d);
  if (i < 0 || i >= (int64_t)p.numel.size()) { set_error("tensor index %lld out of range", (long long)i); return PL_EINVAL; }
  return p.numel[i];
}
extern "C" int64_t pl_param_arena_floats(const PLDesc* d) {
  if (check_desc(d, false) != PL_OK) return PL_EINVAL;
  return param_layout(d).total;
}

extern "C" size_t pl_wplanes_layer_bytes(const PLDesc* d) {
  if (check_desc(d, false) != PL_OK) return 0;
  if (!(d->bn && d->hidden % 128 == 0 && d->num_stage >= 1) || (d->dtype != PL_F16X3 && d->dtype != PL_BF16)) return 0;
  return (size_t)d->hidden * d->hidden * 2 * (d->dtype == PL_F16X3 ? 2 : 1);
}
extern "C" size_t pl_wplanes_bytes(const PLDesc* d) { return pl_wplanes_layer_bytes(d) * 2 * (size_t)(d ? d->num_stage : 0); }
extern "C" float pl_weight_plane_scale(void) { return kWeightPlaneScale; }

// refresh PLDesc.wplanes from the current parameters (what a forward call does first when wplanes_valid == 0)
extern "C" int pl_wplanes_refresh(const PLDesc* d, void* stream) {
  PL_TRY(check_desc(d, true));
  if (!d->wplanes || pl_wplanes_bytes(d) == 0) PL_FAIL(PL_EINVAL, "pl_wplanes_refresh: this descriptor has no weight planes");
  PLDesc t = *d;
  t.wplanes_valid = 0;
  const Ws w = plan(&t, 128);            // any batch on the planes path: only the layer count and plane kind are used
  if (!w.planes) PL_FAIL(PL_EINVAL, "pl_wplanes_refresh: this descriptor has no planes path");
  return split_weight_planes(&t, param_layout(&t), w, nullptr, (hipStream_t)stream);
}

extern "C" size_t pl_workspace_bytes(const PLDesc* d, int64_t B) {
  if (check_desc(d, false) != PL_OK || B <= 0) return 0;
  return plan(d, B).total;
}

extern "C" int pl_workspace_view(const PLDesc* d, int64_t B, int which, int64_t layer, size_t* off,
                                 size_t* size) {
  PL_TRY(check_desc(d, false));
  if (B <= 0 || !off || !size) PL_FAIL(PL_EINVAL, "pl_workspace_view: bad arguments");
  const Ws w = plan(d, B);
  if (layer < 0 || layer >= w.L) PL_FAIL(PL_EINVAL, "pl_workspace_view: layer %lld out of range", (long long)layer);
  const size_t hb = (size_t)d->hidden * 4;
  switch (which) {
    case 0: *off = w.z[layer]; *size = w.act_bytes; break;
    case 1:
      if (!w.act_f32[layer])
        PL_FAIL(PL_EINVAL, "pl_workspace_view: the planes path keeps the output of hidden layer %lld as 16-bit planes only", (long long)layer);
      *off = w.act[layer]; *size = w.act_bytes; break;
    case 2: *off = w.bits[layer]; *size = w.bits_bytes; break;
    case 3: *off = w.mean[layer]; *size = hb; break;
    case 4: *off = w.rstd[layer]; *size = hb; break;
    default: PL_FAIL(PL_EINVAL, "pl_workspace_view: which=%d", which);
  }
  return PL_OK;
}

extern "C" int pl_workspace_bitmap_format(const PLDesc* d, int64_t B, int64_t layer) {
  PL_TRY(check_desc(d, false));
  if (B <= 0 || layer < 0 || layer >= 1 + 2 * (int64_t)d->num_stage) PL_FAIL(PL_EINVAL, "pl_workspace_bitmap_format: bad arguments");
  const bool planes = planes_kind(d, B) != 0;
  return (layer > 0 ? small_layer_on(d, planes, B) : small_first_on(d, planes, B)) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------
// forward, eval mode
// ---------------------------------------------------------------------------------------
extern "C" int pl_lifter_fwd_eval(const PLDesc* d, const float* x, float* y, int64_t B, void* ws,
                                  size_t ws_bytes, void* stream) {
  PL_TRY(check_desc(d, true));
  if (!x || !y) PL_FAIL(PL_EINVAL, "pl_lifter_fwd_eval: null x/y");
  if (B <= 0) PL_FAIL(PL_ESHAPE, "pl_lifter_fwd_eval: B=%lld", (long long)B);
  const Ws w = plan(d, B);
  PL_TRY(check_ws(w, ws, ws_bytes));
  const ParamLayout P = param_layout(d);
  hipStream_t s = (hipStream_t)stream;
  const int H = d->hidden;
  // Small and ragged batches (everything the thin GEMMs took: M <= 512 rows off the tile grid): one launch per hidden layer
  // -- Linear, the BatchNorm fold on the running statistics, ReLU, residual -- on the layer kernels of small_layer.hip with
  // the grid also over 64-row blocks, the first layer on its vector-unit form, the output layer from the slabs the last
  // launch leaves: 6 launches instead of 16 at B = 64 (97 -> 47 us), every row the same bits whatever the batch.
  // POSELIFT_SMALL_EVAL=0: the thin-GEMM route (A/B).
  static const bool small_eval_off = [] { const char* e = getenv("POSELIFT_SMALL_EVAL"); return e && e[0] == '0'; }();
  // (whole-tile batches up to 512 rows, too: on the operand-planes path an evaluation of 128 ... 512 rows is ~20 launches of
  //  8 ... 32 tiles each -- 143 us at any of these sizes -- where the layer kernels take 50 ... 110 us)
  if (!small_eval_off && d->bn && d->bn_running && B <= thin_gemm_max_m() && small_layer_ok(2, H, H) &&
      small_first_ok(d->in_dim) && small_top_ok(d->out_dim)) {
    const float* a_in = x;
    for (int l = 0; l < w.L; ++l) {
      const Layer ly = layer_of(d, P, nullptr, l);
      const float* resid = (l >= 2 && (l % 2) == 0) ? f32(ws, w.act[l - 2]) : nullptr;
      const bool last = l == w.L - 1;
      const bool f16 = small_f16_on(d, w);
      // (a whole-tile batch's plan keeps no fp32 activation for the odd layers: their output goes through the z buffer)
      float* out = w.act_f32[l] ? f32(ws, w.act[l]) : f32(ws, w.z[l]);
      PL_TRY(launch_small_layer_eval(a_in, ly.W, ly.b, ly.gamma, ly.beta, d->bn_eps, ly.rm, ly.rv, resid, out,
                                     (int)B, H, ly.K, s, l == 0, last ? d->params + P.off[4 * w.L] : nullptr,
                                     last ? f32(ws, w.slabs) : nullptr, d->out_dim,
                                     (f16 && l > 0) ? u16(ws, w.sactp[l - 1]) : nullptr, (f16 && !last) ? u16(ws, w.sactp[l]) : nullptr));
      a_in = out;
    }
    return launch_small_out(f32(ws, w.slabs), H / 16, (int)B, d->out_dim, d->params + P.off[4 * w.L + 1], y, s);
  }
  for (int l = 0; l < w.L; ++l) {
    const Layer ly = layer_of(d, P, nullptr, l);
    PL_TRY(launch_bn_fold_eval(ly.b, ly.gamma, ly.beta, ly.rm, ly.rv, d->bn_eps, d->bn, H,
                               f32(ws, w.scale) + (size_t)l * H, f32(ws, w.shift) + (size_t)l * H, s));
  }
  const float* a_in = x;
  if (w.planes) PL_TRY(split_weight_planes(d, P, w, ws, s));
  const int64_t BH = B * H;
  for (int l = 0; l < w.L; ++l) {
    const Layer ly = layer_of(d, P, nullptr, l);
    // (planes path: the eval-mode output of an odd layer goes through its z buffer -- its fp32 activation is not kept)
    float* out = w.act_f32[l] ? f32(ws, w.act[l]) : f32(ws, w.z[l]);
    GemmArgs g = {};
    g.A = a_in; g.B = ly.W; g.C = out;
    g.M = (int)B; g.N = H; g.K = ly.K; g.lda = ly.K; g.ldb = ly.K; g.ldc = H; g.split_k = 1;
    g.arith = arith_of(d);
    g.col_scale = f32(ws, w.scale) + (size_t)l * H;
    g.col_shift = f32(ws, w.shift) + (size_t)l * H;
    g.relu = 1;
    if (l >= 2 && (l % 2) == 0) g.resid = f32(ws, w.act[l - 2]);
    g.thin_scratch = f32(ws, w.slabs); g.thin_scratch_floats = w.slab_floats;
    if (w.planes && l > 0) {
      PlanesGemmArgs pg = planes_args(w.pkind, u16(ws, w.actp[l - 1]), BH, H, wplane(d, w, ws, l), (int64_t)H * H, H, out, (int)B, H, H,
                                      1.0f / (kActPlaneScale * kWeightPlaneScale), nullptr);
      pg.e.col_scale = g.col_scale; pg.e.col_shift = g.col_shift; pg.e.relu = 1; pg.e.resid = g.resid;
      PL_TRY(launch_gemm_planes(kNT, pg, s));
    } else {
      PL_TRY(launch_gemm_f32(kNT, g, s));
    }
    if (w.planes && l + 1 < w.L) {
      PlaneOut po = {u16(ws, w.actp[l]), u16(ws, w.actp[l]) + BH, kActPlaneScale, nullptr, w.pkind};
      PL_TRY(launch_split_planes(out, BH, po, s));
    }
    a_in = out;
  }
  return gemm_out_layer(a_in, d->params + P.off[4 * w.L], d->params + P.off[4 * w.L + 1], y, (int)B,
                        d->out_dim, H, f32(ws, w.slabs), s);
}

// ---------------------------------------------------------------------------------------
// forward, training mode
// ---------------------------------------------------------------------------------------
// eval_bn: the forward of model.eval() computed by the TRAINING kernels, so that everything a backward pass needs is
// saved in the workspace (pre-activations, ReLU bitmaps): BatchNorm normalises with the running statistics (which are
// not touched), Dropout is the identity.  pl_lifter_fwd_eval is the fast, nothing-saved form of the same function.
static int fwd_saved_impl(const PLDesc* d, const float* x, float* y, int64_t B, void* ws, size_t ws_bytes,
                          uint64_t seed, uint64_t step, const uint64_t* inject_keep, void* stream, bool eval_bn,
                          bool defer_out_reduce = false);

extern "C" int pl_lifter_fwd_train(const PLDesc* d, const float* x, float* y, int64_t B, void* ws,
                                   size_t ws_bytes, uint64_t seed, uint64_t step,
                                   const uint64_t* inject_keep, void* stream) {
  return fwd_saved_impl(d, x, y, B, ws, ws_bytes, seed, step, inject_keep, stream, false);
}

extern "C" int pl_lifter_fwd_eval_saved(const PLDesc* d, const float* x, float* y, int64_t B, void* ws,
                                        size_t ws_bytes, void* stream) {
  return fwd_saved_impl(d, x, y, B, ws, ws_bytes, 0, 0, nullptr, stream, true);
}

static int fwd_saved_impl(const PLDesc* d, const float* x, float* y, int64_t B, void* ws, size_t ws_bytes,
                          uint64_t seed, uint64_t step, const uint64_t* inject_keep, void* stream, bool eval_bn,
                          bool defer_out_reduce) {
  PL_TRY(check_desc(d, true));
  if (!x || !y) PL_FAIL(PL_EINVAL, "pl_lifter_fwd_train: null x/y");
  if (B <= 0) PL_FAIL(PL_ESHAPE, "pl_lifter_fwd_train: B=%lld", (long long)B);
  if (d->bn && B < 2 && !eval_bn)
    PL_FAIL(PL_EBATCH, "Expected more than 1 value per channel when training (B=%lld)", (long long)B);
  const Ws w = plan(d, B);
  PL_TRY(check_ws(w, ws, ws_bytes));
  const ParamLayout P = param_layout(d);
  hipStream_t s = (hipStream_t)stream;
  const int H = d->hidden;
  const size_t inj_stride = (size_t)B * bitmap_words_per_row(H);
  const float* a_in = x;
  const int64_t BH = B * H;
  if (w.planes) PL_TRY(split_weight_planes(d, P, w, ws, s));
  // (small batches, fused train step: the last hidden layer's launch leaves the output Linear's slabs -- small_layer.hip)
  const bool head_slabs = defer_out_reduce && !eval_bn && bn_small(d, w, B) && small_head_on(d, w.planes, B);
  for (int l = 0; l < w.L; ++l) {
    const Layer ly = layer_of(d, P, nullptr, l);
    GemmArgs g = {};
    g.A = a_in; g.B = ly.W; g.C = f32(ws, w.z[l]); g.bias = ly.b;
    g.M = (int)B; g.N = H; g.K = ly.K; g.lda = ly.K; g.ldb = ly.K; g.ldc = H; g.split_k = 1;
    g.arith = arith_of(d);
    g.thin_scratch = f32(ws, w.slabs); g.thin_scratch_floats = w.slab_floats;
    const bool skinny = l == 0 && skinny_supported(ly.K, H);
    const int groups = skinny ? skinny_stat_groups((int)B) : w.G;
    float* stat = f32(ws, w.stat);
    // small batches: statistics, finalize and apply in ONE launch straight from z (bn_small_fwd_kernel) -- no partials
    const bool small = bn_small(d, w, B) && !eval_bn;
    if (small && (l > 0 ? small_layer_on(d, w.planes, B) : small_first_on(d, w.planes, B))) {
      // ... and the Linear rides in the same launch
      const float* resid = (l >= 2 && (l % 2) == 0) ? f32(ws, w.act[l - 2]) : nullptr;
      const bool slabs_here = head_slabs && l == w.L - 1;
      PL_TRY(launch_small_layer_fwd(a_in, ly.W, ly.b, ly.gamma, ly.beta, d->bn_eps, d->bn_momentum, ly.rm, ly.rv, ly.nbt,
                                    f32(ws, w.mean[l]), f32(ws, w.rstd[l]), resid, g.C, f32(ws, w.act[l]), u64(ws, w.bits[l]),
                                    (int)B, H, ly.K, d->p_dropout, seed, step, l,
                                    inject_keep ? inject_keep + (size_t)l * inj_stride : nullptr, s, d->step_dev, l == 0,
                                    slabs_here ? d->params + P.off[4 * w.L] : nullptr, slabs_here ? f32(ws, w.slabs) : nullptr,
                                    d->out_dim, (small_f16_on(d, w) && l > 0) ? u16(ws, w.sactp[l - 1]) : nullptr,
                                    (small_f16_on(d, w) && l + 1 < w.L) ? u16(ws, w.sactp[l]) : nullptr));
      a_in = f32(ws, w.act[l]);
      continue;
    }
    if (d->bn && !eval_bn && !small) {
      g.stat_sum = stat + (size_t)sync_rank(d) * 2 * groups * H;
      g.stat_m2 = g.stat_sum + (size_t)groups * H;
    }
    if (skinny) {
      PL_TRY(launch_skinny_wide_out(a_in, ly.W, ly.b, g.C, (int)B, ly.K, H, false, g.stat_sum, g.stat_m2, s));
    } else if (w.planes && l > 0 && mid_linear_on(d, w, B)) {
      // 128 ... 512 rows: the tile GEMM has 8 ... 32 tiles for 256 CUs (22 us whatever the size); one 64-row block x 16 columns
      // per workgroup on the layer kernels' contraction instead, same operand planes, same statistics partials
      PL_TRY(launch_small_linear_stats(nullptr, u16(ws, w.actp[l - 1]), ly.W, ly.b, g.C, (int)B, H, H, g.stat_sum, g.stat_m2, groups, s));
    } else if (w.planes && l > 0) {
      PlanesGemmArgs pg = planes_args(w.pkind, u16(ws, w.actp[l - 1]), BH, H, wplane(d, w, ws, l), (int64_t)H * H, H, g.C, (int)B, H, H,
                                      1.0f / (kActPlaneScale * kWeightPlaneScale), nullptr);
      pg.e.bias = ly.b; pg.e.stat_sum = g.stat_sum; pg.e.stat_m2 = g.stat_m2;
      PL_TRY(launch_gemm_planes(kNT, pg, s));
    } else if (l > 0 && !w.planes && mid_linear_f32_on(d, B)) {
      // off the planes path (ragged rows, exact-fp32 / bf16x6 descriptors), up to 512 rows: the same kernel on fp32 operands
      // (exact-fp32 MFMA) instead of the thin GEMM + its reduce or an 8 ... 32-tile GEMM
      PL_TRY(launch_small_linear_stats(a_in, nullptr, ly.W, ly.b, g.C, (int)B, H, H, g.stat_sum, g.stat_m2, groups, s));
    } else {
      PL_TRY(launch_gemm_f32(kNT, g, s));
    }
    const float *scale = nullptr, *shift = nullptr;
    BnFinalizeArgs fin = {};
    bool use_fin = false;
    if (d->bn && eval_bn) {
      float* sc = f32(ws, w.scale) + (size_t)l * H;
      float* sh = f32(ws, w.shift) + (size_t)l * H;
      PL_TRY(launch_bn_eval_stats(ly.gamma, ly.beta, ly.rm, ly.rv, d->bn_eps, H, f32(ws, w.mean[l]), f32(ws, w.rstd[l]),
                                  sc, sh, s));
      scale = sc; shift = sh;
    } else if (d->bn && !small && fin_in_apply(d, groups)) {
      // local statistics of at most 4 groups (<= 256 rows): finalized inside the apply launch below
      fin = BnFinalizeArgs{stat, groups, 64, ly.gamma, ly.beta, d->bn_eps, d->bn_momentum, ly.rm, ly.rv, ly.nbt,
                           f32(ws, w.mean[l]), f32(ws, w.rstd[l])};
      use_fin = true;
    } else if (d->bn && !small) {
      float* sc = f32(ws, w.scale) + (size_t)l * H;
      float* sh = f32(ws, w.shift) + (size_t)l * H;
      PL_TRY(sync_gather(d, stat, (int64_t)2 * groups * H, s));
      PL_TRY(launch_bn_finalize(stat, groups, sync_world(d), (int)B, H, ly.gamma, ly.beta, d->bn_eps,
                                d->bn_momentum, ly.rm, ly.rv, ly.nbt, f32(ws, w.mean[l]),
                                f32(ws, w.rstd[l]), sc, sh, s));
      scale = sc; shift = sh;
    }
    const float* resid = (l >= 2 && (l % 2) == 0) ? f32(ws, w.act[l - 2]) : nullptr;
    if (small) {
      PL_TRY(launch_bn_small_fwd(g.C, ly.gamma, ly.beta, d->bn_eps, d->bn_momentum, ly.rm, ly.rv, ly.nbt, f32(ws, w.mean[l]),
                                 f32(ws, w.rstd[l]), resid, f32(ws, w.act[l]), u64(ws, w.bits[l]), (int)B, H, d->p_dropout, seed,
                                 step, l, inject_keep ? inject_keep + (size_t)l * inj_stride : nullptr, s, d->step_dev));
      a_in = f32(ws, w.act[l]);
      continue;
    }
    PlaneOut po = {nullptr, nullptr, kActPlaneScale, nullptr, 0};
    if (w.planes && l + 1 < w.L) { po.h = u16(ws, w.actp[l]); po.l = po.h + BH; po.kind = w.pkind; }
    float* act = w.act_f32[l] ? f32(ws, w.act[l]) : nullptr;
    PL_TRY(launch_bn_apply(g.C, scale, shift, resid, act, u64(ws, w.bits[l]), (int)B, H,
                           eval_bn ? 0.f : d->p_dropout, seed, step, l,
                           inject_keep ? inject_keep + (size_t)l * inj_stride : nullptr, s, &po,
                           eval_bn ? nullptr : d->step_dev, use_fin ? &fin : nullptr));
    a_in = act;
  }
  if (head_slabs) return PL_OK;
  if (defer_out_reduce)   // (the fused train step: y = bias + slabs is formed by the MSE pass, mse_partial_from_slabs)
    return launch_skinny_narrow_out(a_in, d->params + P.off[4 * w.L], d->params + P.off[4 * w.L + 1], y, (int)B, H, d->out_dim,
                                    f32(ws, w.slabs), s, false);
  return gemm_out_layer(a_in, d->params + P.off[4 * w.L], d->params + P.off[4 * w.L + 1], y, (int)B,
                        d->out_dim, H, f32(ws, w.slabs), s);
}

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------
// Backward over the output layer (if do_output) and hidden layers l_hi .. l_lo (descending).
// The gradient flowing between two calls lives in the workspace (GA/GB), so the pass can be cut
// at any layer boundary: the data-parallel driver all-reduces the first half's gradients while
// the second half is still computing.
// loss_out != NULL (the fused train step): the MSE loss is finalised by this range's one reduce launch (a kind-2 job:
// mse_final_kernel's sum) and the device step counter ticks there, instead of in a launch of their own after the forward.
static int bwd_impl(const PLDesc* d, const float* x, const float* dy, int64_t B, void* ws, size_t ws_bytes,
                    float* dx, float* grads, void* stream, bool do_output, int l_hi, int l_lo, bool eval_bn = false,
                    float* loss_out = nullptr, int loss_partials = 0, const PLAdamWStep* adam = nullptr) {
  PL_TRY(check_desc(d, true));
  if (!x || !dy || !grads) PL_FAIL(PL_EINVAL, "pl_lifter_bwd: null x/dy/flat_grads");
  if (B <= 0) PL_FAIL(PL_ESHAPE, "pl_lifter_bwd: B=%lld", (long long)B);
  const Ws w = plan(d, B);
  PL_TRY(check_ws(w, ws, ws_bytes));
  const ParamLayout P = param_layout(d);
  hipStream_t s = (hipStream_t)stream;
  const int H = d->hidden, O = d->out_dim, Bi = (int)B;
  const float kscale = (!eval_bn && d->p_dropout > 0.f && d->p_dropout < 1.f) ? 1.0f / (1.0f - d->p_dropout) : 1.0f;
  float* slabs = f32(ws, w.slabs);
  float* GA = f32(ws, w.ga);
  float* GB = f32(ws, w.gb);
  float* DZ = f32(ws, w.dz);
  const int64_t BH = B * H;
  const int n_amax = ((H + 255) / 256) * w.RC;
  // bias gradients = column sums of partials; all of them are reduced by ONE launch at the end
  std::vector<const float*> jpart; std::vector<float*> jout; std::vector<int> jR, jH, jkind, jtrans;
  auto job = [&](const float* part, float* out, int R, int Hj, int kind, int transK) {
    jpart.push_back(part); jout.push_back(out); jR.push_back(R); jH.push_back(Hj); jkind.push_back(kind); jtrans.push_back(transK);
  };

  // small batches (small_layer.hip): which layers' forward left a tile-format bitmap, and what describes a layer's BatchNorm
  const bool sl_all = bn_small(d, w, B) && !eval_bn && small_layer_on(d, w.planes, B);
  const bool sl_first = sl_all && small_first_on(d, w.planes, B);
  auto bn_layer = [&](int l) {
    const Layer y = layer_of(d, P, grads, l);
    SmallBnLayer b = {f32(ws, w.z[l]), f32(ws, w.mean[l]), f32(ws, w.rstd[l]), y.gamma, u64(ws, w.bits[l]),
                      l == 0 && !sl_first, y.ggamma, y.gbeta, y.gb};
    return b;
  };
  // the output layer and the BatchNorm backward of the top hidden layer in one launch (dz of that layer: DZ)
  const int n_loss_part = loss_partials > 0 ? loss_partials : mse_partials(B * O);   // partial sums of the loss in w.mse
  const bool top_fused = do_output && sl_all && small_top_on(d, w.planes, B) && l_hi == w.L - 1 && l_hi >= l_lo;
  // the AdamW step carried by the backward launches (pl_lifter_train_step): which slice of the arena rides with layer l
  static const bool dw_off = [] { const char* e = getenv("POSELIFT_SMALL_DW"); return e && e[0] == '0'; }();
  const bool adam_rides = adam && top_fused && l_lo == 0 && !dw_off && H % 128 == 0 && adam_ride_on();
  auto ride = [&](int64_t lo, int64_t hi) {
    AdamWRide r = {};
    r.p = const_cast<float*>(d->params) + lo; r.g = grads + lo; r.m = adam->m + lo; r.v = adam->v + lo; r.n = hi - lo;
    r.lr = adam->lr; r.beta1 = adam->beta1; r.beta2 = adam->beta2; r.eps = adam->eps; r.wd = adam->weight_decay; r.gscale = 1.0f;
    r.t = adam->t; r.lr_dev = adam->lr_dev; r.t_dev = adam->t_dev;
    return r;
  };
  if (top_fused) {
    PL_TRY(launch_small_top_bwd(dy, d->params + P.off[4 * w.L], f32(ws, w.act[w.L - 1]), Bi, H, O, GA, grads + P.off[4 * w.L],
                                grads + P.off[4 * w.L + 1], bn_layer(w.L - 1), kscale, DZ, s, f32(ws, w.mse),
                                n_loss_part, 1.0f / (float)(B * O), loss_out, const_cast<uint64_t*>(d->step_dev)));
  } else if (do_output) {
  // final Linear (LinearModel.w2): dW = dy^T h, db = sum dy, g = dy W
  const float* W5 = d->params + P.off[4 * w.L];
  const float* h_last = f32(ws, w.act[w.L - 1]);
  if (skinny_supported(O, H)) {
    // dW5 partials, and -- the kernel holds every row of dy in its A fragments -- the bias gradient's partial column sums
    // (Round 3 tried this launch on a side stream -- nothing reads its output before the closing reduce, and it and the head of
    //  the chain below are both latency-bound -- forked and joined with events: 0.627 -> 0.660 ms per step eager, 0.644 -> 0.663
    //  replayed from a graph, same box: the two cross-stream dependencies cost more than the 13 us launch they hide.)
    PL_TRY(launch_skinny_wide_in(dy, h_last, grads + P.off[4 * w.L], Bi, O, H, false, f32(ws, w.skp_out), s, false,
                                 f32(ws, w.outpart)));
    job(f32(ws, w.skp_out), grads + P.off[4 * w.L], skinny_in_chunks(Bi), O * H, 0, 0);
    job(f32(ws, w.outpart), grads + P.off[4 * w.L + 1], skinny_in_chunks(Bi), O, 0, 0);
  } else {
    PL_TRY(gemm_tn_reduced(dy, O, h_last, H, grads + P.off[4 * w.L], O, H, Bi, slabs, s));
    PL_TRY(launch_colsum_partial(dy, Bi, O, f32(ws, w.outpart), s));
    job(f32(ws, w.outpart), grads + P.off[4 * w.L + 1], colsum_chunks(Bi), O, 0, 0);
  }
  {
    GemmArgs g = {};
    g.A = dy; g.B = W5; g.C = GA; g.M = Bi; g.N = H; g.K = O; g.lda = O; g.ldb = H; g.ldc = H; g.split_k = 1;
    if (skinny_supported(O, H)) {
      GemmArgs be = {};
      const int lt = w.L - 1;
      if (fused_reduce(d, w.planes, lt, w.L, eval_bn)) {      // pass 1 of the top hidden layer, on the block just produced
        const BnrSlab sl = bnr_slab(d, w, ws, Bi / 64, (w.pkind == 2 && lt > 0) ? (Bi / 64) * (H / 32) : 0, eval_bn);
        be.bnr_z = f32(ws, w.z[lt]); be.bnr_bits = u64(ws, w.bits[lt]);
        be.bnr_mean = f32(ws, w.mean[lt]); be.bnr_rstd = f32(ws, w.rstd[lt]); be.bnr_kscale = kscale;
        be.bnr_part_dy = sl.mine; be.bnr_part_dyz = sl.mine + (size_t)(Bi / 64) * H;
        be.bnr_amax = (w.pkind == 2 && lt > 0) ? sl.amax_mine : nullptr;
      }
      PL_TRY(launch_skinny_wide_out(dy, W5, nullptr, GA, Bi, O, H, true, nullptr, nullptr, s, &be));
    } else {
      PL_TRY(launch_gemm_f32(kNN, g, s));
    }
  }
  }  // do_output

  bool first_wgrad_done = false;
  for (int l = l_hi; l >= l_lo; --l) {
    const Layer ly = layer_of(d, P, grads, l);
    // gradient w.r.t. this layer's activation: GA for layer 0 and even layers, GB for odd ones
    const float* gin = (l % 2 == 1) ? GB : GA;
    const uint64_t* bits = u64(ws, w.bits[l]);
    const float* z = f32(ws, w.z[l]);
    const bool pl_layer = w.planes && l > 0;       // this layer's dz feeds the planes GEMM pair
    float* dzs = (pl_layer && w.pkind == 2) ? f32(ws, w.dzscale) + 2 * l : nullptr;   // fp16 planes of dz are range-scaled
    PlaneOut dzo = {nullptr, nullptr, 1.0f, dzs, 0};
    if (pl_layer) { dzo.h = u16(ws, w.dzp); dzo.l = dzo.h + BH; dzo.kind = w.pkind; }
    const bool small = bn_small(d, w, B) && !eval_bn;
    // small batches, layer kernels (small_layer.hip): the dX launch of layer l + 1 already ran this layer's BatchNorm backward
    // (dz_l sits in dzbuf(l)) unless this layer heads the range
    const bool sl = small && small_layer_on(d, w.planes, B);
    auto dzbuf = [&](int layer) { return (sl && (layer & 1)) ? GB : DZ; };   // (alternating: a launch reads dz_l and writes dz_{l-1})
    float* DZl = dzbuf(l);
    if (small && sl && (l < l_hi || top_fused)) {
      // (nothing: done by launch_small_layer_bwd of layer l + 1 / by launch_small_top_bwd)
    } else if (small) {
      // pass 1, the coefficients, dz, the bias gradient and dgamma / dbeta of this layer in one launch (small batches)
      PL_TRY(launch_bn_small_bwd(gin, bits, z, f32(ws, w.mean[l]), f32(ws, w.rstd[l]), ly.gamma, kscale, Bi, H, DZl, ly.ggamma,
                                 ly.gbeta, ly.gb, s, sl && (l > 0 || sl_first)));
    } else if (d->bn) {
      // pass 1 (column sums of dy and dy*zhat): a streaming kernel of its own, or -- round 2 -- already done by the
      // LDS-staged epilogue of the planes GEMM that produced `gin` (round 1 tried it in the dword-per-lane epilogue of
      // the fp32-operand GEMM: +17 us per GEMM for the 7.5 us kernel it removed)
      const bool fr = fused_reduce(d, w.planes, l, w.L, eval_bn);
      const int rc_l = fr ? Bi / 64 : w.RC;
      const int n_amax_l = fr ? (Bi / 64) * (l == w.L - 1 ? H / 32 : H / 64) : n_amax;   // (top layer: skinny epilogue, 32-column strips)
      const BnrSlab sl = bnr_slab(d, w, ws, rc_l, dzs ? n_amax_l : 0, eval_bn);
      float* stat = f32(ws, w.stat);
      if (!fr)
        PL_TRY(launch_bn_bwd_reduce(gin, bits, z, f32(ws, w.mean[l]), f32(ws, w.rstd[l]), kscale, Bi, H,
                                    sl.mine, sl.mine + (size_t)rc_l * H, s, 0, dzs ? sl.amax_mine : nullptr, rc_l));
      // (fused: the partials were written by the GEMM / skinny epilogue that produced `gin`, into this rank's slab)
      if (!eval_bn) PL_TRY(sync_gather(d, stat, sl.floats_per_rank, s));
      PL_TRY(launch_bn_bwd_finalize(stat, rc_l, eval_bn ? 1 : sync_world(d), eval_bn ? 0 : sync_rank(d), Bi, H, ly.gamma,
                                    f32(ws, w.rstd[l]), f32(ws, w.coef), ly.ggamma, ly.gbeta, s,
                                    dzs ? sl.amax0 : nullptr, n_amax_l, dzs, eval_bn ? 1 : 0, sl.floats_per_rank, sl.world));
    } else {
      PL_TRY(launch_fill(ly.ggamma, H, 0.f, s));
      PL_TRY(launch_fill(ly.gbeta, H, 0.f, s));
    }
    if (!small) {
      PL_TRY(launch_bn_bwd_dz(gin, bits, z, f32(ws, w.mean[l]), f32(ws, w.rstd[l]), f32(ws, w.coef), kscale,
                              d->bn, Bi, H, pl_layer ? nullptr : DZ, f32(ws, w.dbpart[l]), s, 0, &dzo, w.RC));
      job(f32(ws, w.dbpart[l]), ly.gb, w.RC, H, 0, 0);
    }
    const float* a_in = l == 0 ? x : (w.planes ? nullptr : f32(ws, w.act[l - 1]));
    if (pl_layer) {
      // dX = dz W (NN) and dW = dz^T a (TN, split-K slabs) on the planes: one launch
      const int splits = tn_splits(H, H, Bi);
      PlanesGemmArgs nn = planes_args(w.pkind, u16(ws, w.dzp), BH, H, wplane(d, w, ws, l), (int64_t)H * H, H,
                                      (l % 2 == 1) ? GA : GB, Bi, H, H, 1.0f / kWeightPlaneScale, dzs ? dzs + 1 : nullptr);
      if (l % 2 == 1) nn.e.addend = GA;
      if (fused_reduce(d, w.planes, l - 1, w.L, eval_bn)) {      // pass 1 of the layer below, on the block just produced
        const bool lower_scaled = w.pkind == 2 && l - 1 > 0;       // its dz planes (fp16) want the range maxima too
        const BnrSlab sl = bnr_slab(d, w, ws, Bi / 64, lower_scaled ? (Bi / 64) * (H / 64) : 0, eval_bn);
        nn.e.bnr_z = f32(ws, w.z[l - 1]);
        nn.e.bnr_bits = u64(ws, w.bits[l - 1]);
        nn.e.bnr_mean = f32(ws, w.mean[l - 1]);
        nn.e.bnr_rstd = f32(ws, w.rstd[l - 1]);
        nn.e.bnr_kscale = kscale;
        nn.e.bnr_part_dy = sl.mine;
        nn.e.bnr_part_dyz = sl.mine + (size_t)(Bi / 64) * H;
        nn.e.bnr_amax = lower_scaled ? sl.amax_mine : nullptr;
      }
      float* wsl = f32(ws, w.wslab[l]);                // this layer's own slabs: combined by the range's one reduce launch
      PlanesGemmArgs tn = planes_args(w.pkind, u16(ws, w.dzp), BH, H, u16(ws, w.actp[l - 1]), BH, H,
                                      splits > 1 ? wsl : ly.gW, H, H, Bi, 1.0f / kActPlaneScale, dzs ? dzs + 1 : nullptr);
      tn.e.split_k = splits;
      PL_TRY(launch_gemm_planes_pair(nn, tn, s));
      if (splits > 1) job(wsl, ly.gW, splits, H * H, 1, 0);
    } else if (l > 0 && sl) {
      // small batches: dX = dz W (+ the skip gradient) and the BatchNorm backward of the layer below in one launch when that
      // layer belongs to this range; the weight gradient is one whole-K launch (K = B <= 64)
      GemmArgs t = {};
      t.arith = arith_of(d);
      t.A = DZl; t.B = a_in; t.M = H; t.N = H; t.K = Bi; t.lda = H; t.ldb = H; t.ldc = H; t.split_k = 1; t.C = ly.gW;
      // (the weight gradient as extra workgroups of the same launch: POSELIFT_SMALL_DW=0 keeps it a launch of its own, A/B)
      const bool dw_rides = !dw_off && H % 128 == 0;
      if (l - 1 >= l_lo) {
        // (l == 1: the first layer's weight gradient dW1 = dz_0^T x follows its BatchNorm backward in the same workgroups)
        first_wgrad_done = l == 1 && small_first_ok(d->in_dim);
        // AdamW on this launch's spare workgroups: this layer's bias and BatchNorm parameters (their gradients came with the
        // launch before) and everything above them up to where the launch before started -- the weight matrix of layer
        // l + 1 (its gradient, too), or the output layer behind the top hidden layer
        AdamWRide ar = {};
        if (adam_rides) ar = ride(P.off[4 * l + 1], l == w.L - 1 ? P.total : P.off[4 * (l + 1) + 1]);
        PL_TRY(launch_small_layer_bwd(DZl, ly.W, (l % 2 == 1) ? GA : nullptr, (l % 2 == 1) ? GA : nullptr, Bi, H, H,
                                      bn_layer(l - 1), kscale, dzbuf(l - 1), s, dw_rides ? a_in : nullptr,
                                      dw_rides ? ly.gW : nullptr, first_wgrad_done ? x : nullptr,
                                      first_wgrad_done ? grads + P.off[0] : nullptr, d->in_dim, adam_rides ? &ar : nullptr));
        if (dw_rides) continue;
      } else {
        GemmArgs g = {};
        g.A = DZl; g.B = ly.W; g.M = Bi; g.N = H; g.K = H; g.lda = H; g.ldb = H; g.ldc = H; g.split_k = 1;
        if (l % 2 == 1) { g.C = GA; g.addend = GA; } else { g.C = GB; }
        g.arith = arith_of(d);
        g.thin_scratch = slabs; g.thin_scratch_floats = w.slab_floats;
        PL_TRY(launch_gemm_f32(kNN, g, s));
      }
      PL_TRY(launch_gemm_f32(kTN, t, s));
    } else if (l > 0) {
      // da_in = dz W and dW = dz^T a_in share dz and are independent: ONE launch.  A residual
      // block's first Linear also receives the skip gradient (in GA, added in the epilogue).
      // (Tried: dW on a side stream so that the next layer's BatchNorm-backward kernels overlap it --
      //  -2 % per step only: two single-GEMM workgroups do not fit one CU together, so dX and dW
      //  time-slice the CUs and the dual launch's co-residency is lost.  Same-box A/B, tools/ab_env.py.)
      GemmArgs g = {};
      g.A = DZ; g.B = ly.W; g.M = Bi; g.N = H; g.K = H; g.lda = H; g.ldb = H; g.ldc = H; g.split_k = 1;
      if (l % 2 == 1) { g.C = GA; g.addend = GA; } else { g.C = GB; }
      g.arith = arith_of(d);
      g.thin_scratch = slabs; g.thin_scratch_floats = w.slab_floats;   // (the pair runs as two launches off the tile grid)
      GemmArgs t = {};
      t.arith = g.arith;
      t.A = DZ; t.B = a_in; t.M = H; t.N = H; t.K = Bi; t.lda = H; t.ldb = H; t.ldc = H;
      const int splits = tn_splits(H, H, Bi);
      t.split_k = splits; t.C = splits > 1 ? slabs : ly.gW;
      PL_TRY(launch_gemm_f32_pair(g, t, s));
      if (splits > 1) PL_TRY(launch_reduce_slabs(slabs, splits, (int64_t)H * H, ly.gW, s));
    } else if (first_wgrad_done) {
      // (layer 0's weight gradient came with its BatchNorm backward, small_layer.hip)
    } else if (skinny_supported(ly.K, H)) {
      PL_TRY(launch_skinny_wide_in(a_in, DZ, ly.gW, Bi, ly.K, H, true, f32(ws, w.skp_in), s, false));
      job(f32(ws, w.skp_in), ly.gW, skinny_in_chunks(Bi), ly.K * H, 0, ly.K);
    } else {
      PL_TRY(gemm_tn_reduced(DZ, H, a_in, ly.K, ly.gW, H, ly.K, Bi, slabs, s));
    }
    if (l == 0 && dx) {
      GemmArgs g = {};
      g.A = DZ; g.B = ly.W; g.C = dx; g.M = Bi; g.N = d->in_dim; g.K = H; g.lda = H; g.ldb = d->in_dim;
      g.ldc = d->in_dim; g.split_k = 1;
      PL_TRY(launch_gemm_f32(kNN, g, s));
    }
  }
  float inv_n = 0.f;
  uint64_t* tick = nullptr;
  if (loss_out && !top_fused) {
    const int64_t n = B * O;
    job(f32(ws, w.mse), loss_out, n_loss_part, 1, 2, 0);
    inv_n = 1.0f / (float)n;
    tick = const_cast<uint64_t*>(d->step_dev);
  }
  if (!jpart.empty())
    PL_TRY(launch_reduce_rows_multi(jpart.data(), jR.data(), jH.data(), jout.data(), (int)jpart.size(), s, jkind.data(),
                                    jtrans.data(), inv_n, tick));
  if (adam) {
    // what no backward launch carried: the bottom of the arena (first layer and the first residual Linear) -- or all of it
    const int64_t n = adam_rides ? (w.L > 1 ? P.off[4 * 1 + 1] : P.total) : P.total;
    float* p = const_cast<float*>(d->params);
    if (adam->lr_dev)
      PL_TRY(pl_adamw_flat_dev(p, grads, adam->m, adam->v, n, adam->lr_dev, adam->beta1, adam->beta2, adam->eps,
                               adam->weight_decay, adam->t, adam->t_dev, 1.0f, stream));
    else
      PL_TRY(pl_adamw_flat(p, grads, adam->m, adam->v, n, adam->lr, adam->beta1, adam->beta2, adam->eps, adam->weight_decay,
                           adam->t, 1.0f, stream));
  }
  return PL_OK;
}

extern "C" int pl_lifter_bwd(const PLDesc* d, const float* x, const float* dy, int64_t B, void* ws,
                             size_t ws_bytes, float* dx, float* grads, void* stream) {
  if (!d) PL_FAIL(PL_EINVAL, "descriptor is NULL");
  return bwd_impl(d, x, dy, B, ws, ws_bytes, dx, grads, stream, true, 2 * d->num_stage, 0);
}

extern "C" int pl_lifter_bwd_eval(const PLDesc* d, const float* x, const float* dy, int64_t B, void* ws,
                                  size_t ws_bytes, float* dx, float* grads, void* stream) {
  if (!d) PL_FAIL(PL_EINVAL, "descriptor is NULL");
  return bwd_impl(d, x, dy, B, ws, ws_bytes, dx, grads, stream, true, 2 * d->num_stage, 0, true);
}

static int check_range(const PLDesc* d, int hi, int lo, const char* who) {
  const int L = 1 + 2 * d->num_stage;
  if (lo < 0 || hi < lo || hi > L) PL_FAIL(PL_EINVAL, "%s: layer range hi=%d lo=%d outside 0..%d", who, hi, lo, L);
  return PL_OK;
}

extern "C" int pl_lifter_bwd_layers(const PLDesc* d, const float* x, const float* dy, int64_t B, void* ws,
                                    size_t ws_bytes, float* dx, float* grads, int hi, int lo, void* stream) {
  PL_TRY(check_desc(d, true));
  PL_TRY(check_range(d, hi, lo, "pl_lifter_bwd_layers"));
  const int L = 1 + 2 * d->num_stage;
  return bwd_impl(d, x, dy, B, ws, ws_bytes, lo == 0 ? dx : nullptr, grads, stream, hi == L, hi == L ? L - 1 : hi, lo);
}

// ---------------------------------------------------------------------------------------
// GEMM building block (tests)
// ---------------------------------------------------------------------------------------
extern "C" int pl_gemm_f32(int layout, const float* A, const float* Bm, float* C, int64_t M, int64_t N,
                           int64_t K, const float* bias, int split_k, float* slabs, void* stream) {
  return pl_gemm_arith(layout, PL_F32, A, Bm, C, M, N, K, bias, split_k, slabs, stream);
}

extern "C" int pl_gemm_arith(int layout, int arith, const float* A, const float* Bm, float* C, int64_t M,
                             int64_t N, int64_t K, const float* bias, int split_k, float* slabs, void* stream) {
  if (layout < 0 || layout > 2) PL_FAIL(PL_EINVAL, "pl_gemm_f32: layout %d", layout);
#ifdef PL_ABLATE
  if (arith < 0 || arith > 4) PL_FAIL(PL_EDTYPE, "pl_gemm_arith: arith %d", arith);
#else
  // 5 / 6: test hooks forcing the PL_BF16X6 planes / fragment-split main loop (2 = the library's choice)
  if ((arith < 0 || arith > 2) && arith != 5 && arith != 6) PL_FAIL(PL_EDTYPE, "pl_gemm_arith: arith %d", arith);
#endif
  if (M <= 0 || N <= 0 || K <= 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX)
    PL_FAIL(PL_ESHAPE, "pl_gemm_f32: bad shape");
  GemmArgs g = {};
  g.A = A; g.B = Bm; g.C = C; g.M = (int)M; g.N = (int)N; g.K = (int)K; g.bias = bias;
  g.lda = layout == kTN ? (int)M : (int)K;
  g.ldb = layout == kNT ? (int)K : (int)N;
  g.ldc = (int)N;
  g.split_k = 1;
  g.arith = arith;
  hipStream_t s = (hipStream_t)stream;
  if (split_k > 1) {
    if (layout != kTN || !slabs || bias) PL_FAIL(PL_EINVAL, "pl_gemm_f32: split_k needs layout 2, slabs and no bias");
    g.C = slabs; g.split_k = split_k;
    PL_TRY(launch_gemm_f32(kTN, g, s));
    return launch_reduce_slabs(slabs, split_k, M * N, C, s);
  }
  return launch_gemm_f32((GemmLayout)layout, g, s);
}

// C = op(A) op(B) on 16-bit operand planes: the fp32 operands are split into planes in `scratch` first (the lifter's
// own producers write planes directly), then the planes GEMM of gemm_planes.hip runs.  mode: PL_F16X3 or PL_BF16.
extern "C" size_t pl_gemm_planes_scratch_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (size_t)(align_up(M * K * 4, 256) + align_up(N * K * 4, 256));
}

extern "C" int pl_gemm_planes(int layout, int mode, const float* A, const float* Bm, float* C, int64_t M, int64_t N,
                              int64_t K, const float* bias, float scale_a, float scale_b, void* scratch, void* stream) {
  if (layout < 0 || layout > 2) PL_FAIL(PL_EINVAL, "pl_gemm_planes: layout %d", layout);
  if (mode != PL_F16X3 && mode != PL_BF16) PL_FAIL(PL_EDTYPE, "pl_gemm_planes: mode %d", mode);
  if (!A || !Bm || !C || !scratch) PL_FAIL(PL_EINVAL, "pl_gemm_planes: null pointer");
  if (M <= 0 || N <= 0 || K <= 0 || M * K >= (1ll << 29) || N * K >= (1ll << 29)) PL_FAIL(PL_ESHAPE, "pl_gemm_planes: bad shape");
  if (!(scale_a > 0.f) || !(scale_b > 0.f)) PL_FAIL(PL_EINVAL, "pl_gemm_planes: scales must be positive powers of two");
  hipStream_t s = (hipStream_t)stream;
  unsigned short* pa = static_cast<unsigned short*>(scratch);
  unsigned short* pb = reinterpret_cast<unsigned short*>(static_cast<char*>(scratch) + align_up(M * K * 4, 256));
  const int kind = mode == PL_F16X3 ? 2 : 1;
  PlaneOut oa = {pa, pa + M * K, scale_a, nullptr, kind}, ob = {pb, pb + N * K, scale_b, nullptr, kind};
  PL_TRY(launch_split_planes(A, M * K, oa, s));
  PL_TRY(launch_split_planes(Bm, N * K, ob, s));
  PlanesGemmArgs g = {};
  g.A = pa; g.B = pb; g.a_plane = M * K; g.b_plane = N * K;
  g.lda = layout == kTN ? (int)M : (int)K;
  g.ldb = layout == kNT ? (int)K : (int)N;
  g.mode = mode == PL_F16X3 ? 2 : 0;
  g.out_scale = mode == PL_F16X3 ? 1.0f / (scale_a * scale_b) : 1.0f;
  g.e.C = C; g.e.M = (int)M; g.e.N = (int)N; g.e.K = (int)K; g.e.ldc = (int)N; g.e.split_k = 1; g.e.bias = bias;
  return launch_gemm_planes((GemmLayout)layout, g, s);
}

// K slices of a planes GEMM with few output tiles (the conv weight gradients: K = pixels): enough workgroups for every
// CU twice, slices of whole 32-k tiles, at most 64 slabs -- 128 where the output is small enough for the slab reduce not to
// matter (the stem's weight gradient: 64 x 224 outputs, 2 tiles, 4.2 M pixels: 128 workgroups took 1.49 ms, 256 take half)
extern "C" int pl_gemm_planes_splits(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
  const int cap = M * N <= 32768 ? 128 : 64;
  int s = 1;
  while (s < cap && tiles * s < 512 && K % (32 * 2 * s) == 0 && K / (2 * s) >= 256) s *= 2;
  return s;
}

extern "C" int pl_gemm_planes_raw(int layout, int mode, const void* A, int64_t a_plane, int64_t lda, const void* Bm,
                                  int64_t b_plane, int64_t ldb, float* C, int64_t M, int64_t N, int64_t K, const float* bias,
                                  float out_scale, const float* dyn_inv, float* slabs, float* stat, void* stream) {
  if (layout < 0 || layout > 2) PL_FAIL(PL_EINVAL, "pl_gemm_planes_raw: layout %d", layout);
  if (mode != PL_F16X3 && mode != PL_BF16) PL_FAIL(PL_EDTYPE, "pl_gemm_planes_raw: mode %d", mode);
  if (!A || !Bm || !C) PL_FAIL(PL_EINVAL, "pl_gemm_planes_raw: null pointer");
  if (M <= 0 || N <= 0 || K <= 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX || lda > INT32_MAX || ldb > INT32_MAX)
    PL_FAIL(PL_ESHAPE, "pl_gemm_planes_raw: bad shape");
  hipStream_t s = (hipStream_t)stream;
  PlanesGemmArgs g = {};
  g.A = static_cast<const unsigned short*>(A); g.B = static_cast<const unsigned short*>(Bm);
  g.a_plane = a_plane; g.b_plane = b_plane; g.lda = (int)lda; g.ldb = (int)ldb;
  g.mode = mode == PL_F16X3 ? 2 : 0;
  g.out_scale = mode == PL_F16X3 ? out_scale : 1.0f;
  g.dyn_inv = mode == PL_F16X3 ? dyn_inv : nullptr;
  g.e.M = (int)M; g.e.N = (int)N; g.e.K = (int)K; g.e.ldc = (int)N;
  const int splits = pl_gemm_planes_splits(M, N, K);
  if (splits > 1) {
    if (!slabs) PL_FAIL(PL_EWORKSPACE, "pl_gemm_planes_raw: %d K slices need slabs", splits);
    if (bias || stat) PL_FAIL(PL_EINVAL, "pl_gemm_planes_raw: no bias / statistics on a split-K problem");
    g.e.C = slabs; g.e.split_k = splits;
    PL_TRY(launch_gemm_planes((GemmLayout)layout, g, s));
    return launch_reduce_slabs(slabs, splits, M * N, C, s);
  }
  g.e.C = C; g.e.split_k = 1; g.e.bias = bias;
  if (stat) { g.e.stat_sum = stat; g.e.stat_m2 = stat + (size_t)gemm_stat_groups((int)M) * N; }
  return launch_gemm_planes((GemmLayout)layout, g, s);
}

extern "C" int pl_gemm_stat_groups(int64_t M) { return M > 0 && M <= INT32_MAX ? gemm_stat_groups((int)M) : 0; }

// KxK convolutions on the planes GEMM with the input gathered by the loader waves (implicit GEMM, gemm_planes16.h)
static int conv_planes_geom(GemmArgs& e, int64_t B, int64_t H, int64_t W, int64_t Cin, int KH, int KW, int stride, int pad,
                            int64_t* Ho, int64_t* Wo, const char* who, int stride_w = 0, int pad_w = -1, int pad_w_right = -1) {
  if (stride_w <= 0) stride_w = stride;
  if (pad_w < 0) pad_w = pad;
  if (pad_w_right < 0) pad_w_right = pad_w;          // (the gather pads by its bounds test: only Wo knows the right padding)
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
    PL_FAIL(PL_ESHAPE, "%s: bad geometry", who);
  *Ho = (H + 2 * pad - KH) / stride + 1;
  *Wo = (W + pad_w + pad_w_right - KW) / stride_w + 1;
  if (*Ho <= 0 || *Wo <= 0 || B * *Ho * *Wo > INT32_MAX) PL_FAIL(PL_ESHAPE, "%s: bad geometry", who);
  e.conv_cin = (int)Cin; e.conv_h = (int)H; e.conv_w = (int)W; e.conv_ho = (int)*Ho; e.conv_wo = (int)*Wo;
  e.conv_kw = KW; e.conv_stride = stride; e.conv_stride_w = stride_w; e.conv_pad_h = pad; e.conv_pad_w = pad_w;
  return PL_OK;
}

extern "C" int pl_conv2d_planes_fwd(int mode, const void* x_planes, int64_t x_plane, int64_t B, int64_t H, int64_t W, int64_t Cin,
                                    const void* w_planes, int64_t w_plane, int64_t Cout, int KH, int KW, int stride, int pad,
                                    float* y, float out_scale, const float* dyn_inv, float* stat, void* stream) {
  return pl_conv2d_planes_fwd_hw(mode, x_planes, x_plane, B, H, W, Cin, w_planes, w_plane, Cout, KH, KW, stride, stride, pad, pad,
                                 pad, y, out_scale, dyn_inv, stat, stream);
}

extern "C" int pl_conv2d_planes_fwd_hw(int mode, const void* x_planes, int64_t x_plane, int64_t B, int64_t H, int64_t W,
                                       int64_t Cin, const void* w_planes, int64_t w_plane, int64_t Cout, int KH, int KW,
                                       int stride_h, int stride_w, int pad_h, int pad_w, int pad_w_right, float* y,
                                       float out_scale, const float* dyn_inv, float* stat, void* stream) {
  if (mode != PL_F16X3 && mode != PL_BF16) PL_FAIL(PL_EDTYPE, "pl_conv2d_planes_fwd: mode %d", mode);
  if (!x_planes || !w_planes || !y || Cout <= 0) PL_FAIL(PL_EINVAL, "pl_conv2d_planes_fwd: null pointer");
  PlanesGemmArgs g = {};
  int64_t Ho, Wo;
  PL_TRY(conv_planes_geom(g.e, B, H, W, Cin, KH, KW, stride_h, pad_h, &Ho, &Wo, "pl_conv2d_planes_fwd", stride_w, pad_w, pad_w_right));
  const int64_t K = (int64_t)KH * KW * Cin;
  g.A = static_cast<const unsigned short*>(x_planes); g.B = static_cast<const unsigned short*>(w_planes);
  g.a_plane = x_plane; g.b_plane = w_plane; g.lda = 0; g.ldb = (int)K;
  g.mode = mode == PL_F16X3 ? 2 : 0;
  g.out_scale = mode == PL_F16X3 ? out_scale : 1.0f;
  g.dyn_inv = mode == PL_F16X3 ? dyn_inv : nullptr;
  g.e.C = y; g.e.M = (int)(B * Ho * Wo); g.e.N = (int)Cout; g.e.K = (int)K; g.e.ldc = (int)Cout; g.e.split_k = 1;
  if (stat) { g.e.stat_sum = stat; g.e.stat_m2 = stat + (size_t)gemm_stat_groups(g.e.M) * Cout; }
  return launch_gemm_planes(kNT, g, (hipStream_t)stream);
}

// the folded eval-mode epilogue and the planes output of a planes convolution (PLPlanesEpilogue)
static int apply_planes_epilogue(GemmArgs& e, int mode, const PLPlanesEpilogue* ep, int64_t n_out, const char* who) {
  if (!ep) return PL_OK;
  if ((ep->scale != nullptr) != (ep->shift != nullptr)) PL_FAIL(PL_EINVAL, "%s: scale without shift", who);
  if (ep->relu < 0 || ep->relu > 2) PL_FAIL(PL_EINVAL, "%s: relu %d", who, ep->relu);
  e.bias = ep->bias; e.col_scale = ep->scale; e.col_shift = ep->shift; e.resid = ep->resid; e.relu = ep->relu;
  if (ep->y_planes) {
    if ((reinterpret_cast<uintptr_t>(ep->y_planes) & 15) || (n_out & 7)) PL_FAIL(PL_EINVAL, "%s: y_planes misaligned", who);
    e.cpl_h = static_cast<unsigned short*>(ep->y_planes);
    e.cpl_l = e.cpl_h + n_out;
    e.cpl_scale = kConvActPlaneScale;
    e.cpl_kind = mode == PL_F16X3 ? 2 : 1;
  }
  return PL_OK;
}

extern "C" int pl_conv2d_planes_fwd_ep(int mode, const void* x_planes, int64_t x_plane, int64_t B, int64_t H, int64_t W,
                                       int64_t Cin, const void* w_planes, int64_t w_plane, int64_t Cout, int KH, int KW,
                                       int stride, int pad, float* y, float out_scale, const PLPlanesEpilogue* ep, void* stream) {
  return pl_conv2d_planes_fwd_ep_hw(mode, x_planes, x_plane, B, H, W, Cin, w_planes, w_plane, Cout, KH, KW, stride, stride, pad,
                                    pad, pad, y, out_scale, ep, stream);
}

extern "C" int pl_conv2d_planes_fwd_ep_hw(int mode, const void* x_planes, int64_t x_plane, int64_t B, int64_t H, int64_t W,
                                          int64_t Cin, const void* w_planes, int64_t w_plane, int64_t Cout, int KH, int KW,
                                          int stride_h, int stride_w, int pad_h, int pad_w, int pad_w_right, float* y,
                                          float out_scale, const PLPlanesEpilogue* ep, void* stream) {
  if (mode != PL_F16X3 && mode != PL_BF16) PL_FAIL(PL_EDTYPE, "pl_conv2d_planes_fwd_ep: mode %d", mode);
  if (!x_planes || !w_planes || (!y && !(ep && ep->y_planes)) || Cout <= 0) PL_FAIL(PL_EINVAL, "pl_conv2d_planes_fwd_ep: null pointer");
  PlanesGemmArgs g = {};
  int64_t Ho, Wo;
  PL_TRY(conv_planes_geom(g.e, B, H, W, Cin, KH, KW, stride_h, pad_h, &Ho, &Wo, "pl_conv2d_planes_fwd_ep", stride_w, pad_w, pad_w_right));
  const int64_t K = (int64_t)KH * KW * Cin;
  g.A = static_cast<const unsigned short*>(x_planes); g.B = static_cast<const unsigned short*>(w_planes);
  g.a_plane = x_plane; g.b_plane = w_plane; g.lda = 0; g.ldb = (int)K;
  g.mode = mode == PL_F16X3 ? 2 : 0;
  g.out_scale = mode == PL_F16X3 ? out_scale : 1.0f;
  g.e.C = y; g.e.M = (int)(B * Ho * Wo); g.e.N = (int)Cout; g.e.K = (int)K; g.e.ldc = (int)Cout; g.e.split_k = 1;
  PL_TRY(apply_planes_epilogue(g.e, mode, ep, B * Ho * Wo * Cout, "pl_conv2d_planes_fwd_ep"));
  return launch_gemm_planes(kNT, g, (hipStream_t)stream);
}

extern "C" int pl_deconv4x4s2_planes_fwd_ep(int mode, const void* x_planes, int64_t x_plane, int64_t B, int64_t H, int64_t W,
                                            int64_t Cin, const void* wsub_planes, int64_t wsub_plane, int64_t Cout, float* y,
                                            float out_scale, const PLPlanesEpilogue* ep, void* stream) {
  if (mode != PL_F16X3 && mode != PL_BF16) PL_FAIL(PL_EDTYPE, "pl_deconv4x4s2_planes_fwd_ep: mode %d", mode);
  if (!x_planes || !wsub_planes || (!y && !(ep && ep->y_planes)) || Cout <= 0) PL_FAIL(PL_EINVAL, "pl_deconv4x4s2_planes_fwd_ep: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || B * H * W > INT32_MAX) PL_FAIL(PL_ESHAPE, "pl_deconv4x4s2_planes_fwd_ep: bad geometry");
  if (ep && ep->resid) PL_FAIL(PL_EINVAL, "pl_deconv4x4s2_planes_fwd_ep: no residual on a transposed convolution");
  const int64_t K = 4 * Cin, nsub = Cout * K;
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      PlanesGemmArgs g = {};
      g.e.conv_cin = (int)Cin; g.e.conv_h = (int)H; g.e.conv_w = (int)W; g.e.conv_ho = (int)H; g.e.conv_wo = (int)W;
      g.e.conv_kw = 2; g.e.conv_stride = 1;
      g.e.conv_pad_h = ph ? 0 : 1; g.e.conv_pad_w = pw ? 0 : 1;
      g.e.scat_on = 1; g.e.scat_ph = ph; g.e.scat_pw = pw;
      g.A = static_cast<const unsigned short*>(x_planes);
      g.B = static_cast<const unsigned short*>(wsub_planes) + (size_t)(ph * 2 + pw) * nsub;
      g.a_plane = x_plane; g.b_plane = wsub_plane; g.lda = 0; g.ldb = (int)K;
      g.mode = mode == PL_F16X3 ? 2 : 0;
      g.out_scale = mode == PL_F16X3 ? out_scale : 1.0f;
      g.e.C = y; g.e.M = (int)(B * H * W); g.e.N = (int)Cout; g.e.K = (int)K; g.e.ldc = (int)Cout; g.e.split_k = 1;
      PL_TRY(apply_planes_epilogue(g.e, mode, ep, B * 4 * H * W * Cout, "pl_deconv4x4s2_planes_fwd_ep"));
      PL_TRY(launch_gemm_planes(kNT, g, (hipStream_t)stream));
    }
  return PL_OK;
}

// nn.ConvTranspose2d(4, 2, 1, bias=False) forward on the planes GEMM: four 2x2-tap convolutions at the INPUT resolution,
// one per output parity (no zero insertion), each storing straight into its pixels of y [B][2H][2W][Cout].
// wsub_planes: planes of conv.deconv_subkernels(weight) = [4 parities][Cout][2][2][Cin].
extern "C" int pl_deconv4x4s2_planes_fwd(int mode, const void* x_planes, int64_t x_plane, int64_t B, int64_t H, int64_t W,
                                         int64_t Cin, const void* wsub_planes, int64_t wsub_plane, int64_t Cout, float* y,
                                         float out_scale, const float* dyn_inv, void* stream) {
  if (mode != PL_F16X3 && mode != PL_BF16) PL_FAIL(PL_EDTYPE, "pl_deconv4x4s2_planes_fwd: mode %d", mode);
  if (!x_planes || !wsub_planes || !y || Cout <= 0) PL_FAIL(PL_EINVAL, "pl_deconv4x4s2_planes_fwd: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || B * H * W > INT32_MAX) PL_FAIL(PL_ESHAPE, "pl_deconv4x4s2_planes_fwd: bad geometry");
  const int64_t K = 4 * Cin, nsub = Cout * K;
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      PlanesGemmArgs g = {};
      g.e.conv_cin = (int)Cin; g.e.conv_h = (int)H; g.e.conv_w = (int)W; g.e.conv_ho = (int)H; g.e.conv_wo = (int)W;
      g.e.conv_kw = 2; g.e.conv_stride = 1;
      g.e.conv_pad_h = ph ? 0 : 1; g.e.conv_pad_w = pw ? 0 : 1;      // even outputs: taps from rows a-1, a; odd: a, a+1
      g.e.scat_on = 1; g.e.scat_ph = ph; g.e.scat_pw = pw;
      g.A = static_cast<const unsigned short*>(x_planes);
      g.B = static_cast<const unsigned short*>(wsub_planes) + (size_t)(ph * 2 + pw) * nsub;
      g.a_plane = x_plane; g.b_plane = wsub_plane; g.lda = 0; g.ldb = (int)K;
      g.mode = mode == PL_F16X3 ? 2 : 0;
      g.out_scale = mode == PL_F16X3 ? out_scale : 1.0f;
      g.dyn_inv = mode == PL_F16X3 ? dyn_inv : nullptr;
      g.e.C = y; g.e.M = (int)(B * H * W); g.e.N = (int)Cout; g.e.K = (int)K; g.e.ldc = (int)Cout; g.e.split_k = 1;
      PL_TRY(launch_gemm_planes(kNT, g, (hipStream_t)stream));
    }
  return PL_OK;
}

extern "C" int pl_conv2d_planes_wgrad(int mode, const void* dz_planes, int64_t dz_plane, const void* x_planes, int64_t x_plane,
                                      int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int KH, int KW, int stride,
                                      int pad, float* dw, float out_scale, const float* dyn_inv, float* slabs, void* stream) {
  return pl_conv2d_planes_wgrad_hw(mode, dz_planes, dz_plane, x_planes, x_plane, B, H, W, Cin, Cout, KH, KW, stride, stride, pad,
                                   pad, pad, dw, out_scale, dyn_inv, slabs, stream);
}

extern "C" int pl_conv2d_planes_wgrad_hw(int mode, const void* dz_planes, int64_t dz_plane, const void* x_planes,
                                         int64_t x_plane, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int KH,
                                         int KW, int stride_h, int stride_w, int pad_h, int pad_w, int pad_w_right, float* dw,
                                         float out_scale, const float* dyn_inv, float* slabs, void* stream) {
  if (mode != PL_F16X3 && mode != PL_BF16) PL_FAIL(PL_EDTYPE, "pl_conv2d_planes_wgrad: mode %d", mode);
  if (!dz_planes || !x_planes || !dw || Cout <= 0) PL_FAIL(PL_EINVAL, "pl_conv2d_planes_wgrad: null pointer");
  PlanesGemmArgs g = {};
  int64_t Ho, Wo;
  PL_TRY(conv_planes_geom(g.e, B, H, W, Cin, KH, KW, stride_h, pad_h, &Ho, &Wo, "pl_conv2d_planes_wgrad", stride_w, pad_w, pad_w_right));
  const int64_t N = (int64_t)KH * KW * Cin, K = B * Ho * Wo;
  g.A = static_cast<const unsigned short*>(dz_planes); g.B = static_cast<const unsigned short*>(x_planes);
  g.a_plane = dz_plane; g.b_plane = x_plane; g.lda = (int)Cout; g.ldb = 0;
  g.mode = mode == PL_F16X3 ? 2 : 0;
  g.out_scale = mode == PL_F16X3 ? out_scale : 1.0f;
  g.dyn_inv = mode == PL_F16X3 ? dyn_inv : nullptr;
  g.e.M = (int)Cout; g.e.N = (int)N; g.e.K = (int)K; g.e.ldc = (int)N;
  const int splits = pl_gemm_planes_splits(Cout, N, K);
  hipStream_t s = (hipStream_t)stream;
  if (splits > 1) {
    if (!slabs) PL_FAIL(PL_EWORKSPACE, "pl_conv2d_planes_wgrad: %d K slices need slabs", splits);
    g.e.C = slabs; g.e.split_k = splits;
    PL_TRY(launch_gemm_planes(kTN, g, s));
    return launch_reduce_slabs(slabs, splits, Cout * N, dw, s);
  }
  g.e.C = dw; g.e.split_k = 1;
  return launch_gemm_planes(kTN, g, s);
}

// ---------------------------------------------------------------------------------------
// measurement hook (bench.py): per-launch GEMM durations from HIP events on the launch stream
// ---------------------------------------------------------------------------------------
extern "C" int pl_prof_enable(int on) { return prof_enable(on); }
extern "C" int pl_prof_read(double min_flops, double max_flops, double* ms_total, int64_t* launches,
                            double* flops_total) {
  if (!ms_total || !launches || !flops_total) PL_FAIL(PL_EINVAL, "pl_prof_read: null pointer");
  return prof_read(min_flops, max_flops, ms_total, launches, flops_total);
}

// ---------------------------------------------------------------------------------------
// fused train step: forward (training mode) + MSE(mean) + backward in one call
// ---------------------------------------------------------------------------------------
static int train_fwd_bwd_impl(const PLDesc* d, const float* x, const float* target, int64_t B, void* ws, size_t ws_bytes,
                              uint64_t seed, uint64_t step, float* y, float* loss, float* grads, int hi, int lo, void* stream,
                              const PLAdamWStep* adam);

extern "C" int pl_lifter_train_fwd_bwd(const PLDesc* d, const float* x, const float* target, int64_t B, void* ws,
                                       size_t ws_bytes, uint64_t seed, uint64_t step, float* y, float* loss,
                                       float* grads, int hi, int lo, void* stream) {
  return train_fwd_bwd_impl(d, x, target, B, ws, ws_bytes, seed, step, y, loss, grads, hi, lo, stream, nullptr);
}

extern "C" int pl_lifter_step_carries_adamw(const PLDesc* d, int64_t B) {
  PL_TRY(check_desc(d, false));
  if (B <= 0) return 0;
  const bool planes = planes_kind(d, B) != 0;
  static const bool dw_off = [] { const char* e = getenv("POSELIFT_SMALL_DW"); return e && e[0] == '0'; }();
  return (bn_small_ok(d, planes, B) && small_head_on(d, planes, B) && !dw_off && d->hidden % 128 == 0 && adam_ride_on()) ? 1 : 0;
}

extern "C" int pl_lifter_train_step(const PLDesc* d, const float* x, const float* target, int64_t B, void* ws, size_t ws_bytes,
                                    uint64_t seed, uint64_t step, float* y, float* loss, float* grads, const PLAdamWStep* opt,
                                    void* stream) {
  if (!opt || !opt->m || !opt->v || (opt->lr_dev != nullptr) != (opt->t_dev != nullptr) || opt->t < (opt->t_dev ? 0 : 1))
    PL_FAIL(PL_EINVAL, "pl_lifter_train_step: bad optimizer description");
  if (!d) PL_FAIL(PL_EINVAL, "descriptor is NULL");
  return train_fwd_bwd_impl(d, x, target, B, ws, ws_bytes, seed, step, y, loss, grads, 1 + 2 * d->num_stage, 0, stream, opt);
}

static int train_fwd_bwd_impl(const PLDesc* d, const float* x, const float* target, int64_t B, void* ws, size_t ws_bytes,
                              uint64_t seed, uint64_t step, float* y, float* loss, float* grads, int hi, int lo, void* stream,
                              const PLAdamWStep* adam) {
  PL_TRY(check_desc(d, true));
  if (!x || !target || !y || !loss || !grads) PL_FAIL(PL_EINVAL, "pl_lifter_train_fwd_bwd: null pointer");
  if (B <= 0) PL_FAIL(PL_ESHAPE, "pl_lifter_train_fwd_bwd: B=%lld", (long long)B);
  PL_TRY(check_range(d, hi, lo, "pl_lifter_train_fwd_bwd"));
  const Ws w = plan(d, B);
  PL_TRY(check_ws(w, ws, ws_bytes));
  float* dy = f32(ws, w.dyout);
  const int L = 1 + 2 * d->num_stage;
  bool small_head = false;      // the loss partials in w.mse are launch_small_mse's
  if (hi == L) {
    const int H = d->hidden, O = d->out_dim;
    static const bool head_off = [] { const char* e = getenv("POSELIFT_HEAD_UNFUSED"); return e && e[0] == '1'; }();
    const int so = skinny_narrow_out_supported(H, O) ? skinny_narrow_out_splits((int)B, H) : 0;
    if (!head_off && bn_small(d, w, B) && small_head_on(d, w.planes, B)) {
      // small batches: the output Linear's slabs come from the last hidden layer's launch (small_layer.hip)
      PL_TRY(fwd_saved_impl(d, x, y, B, ws, ws_bytes, seed, step, nullptr, stream, false, true));
      const ParamLayout P = param_layout(d);
      PL_TRY(launch_small_mse(f32(ws, w.slabs), H / 16, (int)B, O, d->params + P.off[4 * L + 1], target, 1.0f, y, dy,
                              f32(ws, w.mse), (hipStream_t)stream));
      small_head = true;
    } else if (!head_off && so && mse_from_slabs_supported(so, O)) {
      // the output Linear's slab reduce folded into the MSE pass: one launch less, the same bits
      PL_TRY(fwd_saved_impl(d, x, y, B, ws, ws_bytes, seed, step, nullptr, stream, false, true));
      const ParamLayout P = param_layout(d);
      PL_TRY(mse_partial_from_slabs(f32(ws, w.slabs), so, (int)B, O, d->params + P.off[4 * L + 1], target, 1.0f, y, dy,
                                    f32(ws, w.mse), stream));
    } else {
      PL_TRY(pl_lifter_fwd_train(d, x, y, B, ws, ws_bytes, seed, step, nullptr, stream));
      PL_TRY(mse_partial_only(y, target, B * d->out_dim, 1.0f, dy, f32(ws, w.mse), stream));
    }
  }
  return bwd_impl(d, x, dy, B, ws, ws_bytes, nullptr, grads, stream, hi == L, hi == L ? L - 1 : hi, lo, false,
                  hi == L ? loss : nullptr, small_head ? small_mse_partials((int)B, d->out_dim) : 0, adam);
}
