// HBM-streaming kernels of the lifter path: BatchNorm statistics/apply/backward, ReLU,
// dropout, residual add, slab reductions, MSE, MPJPE and the flat AdamW.
//
// They replace, on the reference path, native_batch_norm(+backward), relu_, dropout, add
// (phase1_lifting/baselineModel.py:35-37,41-45,92-94), mse_loss (train_1.py:37,94),
// loss_MPJPE (train_1.py:19-23) and AdamW.step (train_1.py:39,96).
//
// Common shape: a (B x H) fp32 activation is walked with one float4 per lane, a
// wavefront covering 256 consecutive columns of one row (1 KiB per wave-instruction,
// fully coalesced).  Every column reduction is a fixed-order two-stage sum (per-block
// partials, then a small finalize kernel): results are bitwise reproducible, no float
// atomics anywhere.
#include "adamw.h"
#include "bn_pieces.h"
#include "philox.h"
#include <stdlib.h>

#include "pl_internal.h"
#include "plane_store.h"

namespace pl {
namespace {

constexpr int NTHR = 256;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---- GEMM operand planes written by the kernel that produces the tensor (pl_internal.h PlaneOut) ----------
// Four consecutive elements per lane: one 8-byte store per plane (512 B per wave instruction).
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int64_t n4, PlaneOut o) {
  const PlaneDst d = plane_dst(o);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    store_planes4(d, (size_t)i * 4, ld4(x + 4 * i));
}

// The same from a STRIDED 4-D view of the source (signed element strides; output contiguous [d0][d1][d2][d3], d3 % 4 == 0): a
// convolution weight in the layout a GEMM wants -- OIHW -> OHWI, its transpose, its spatial flip -- goes from the parameter
// to planes in ONE launch instead of a layout copy (+ a flip) + a split.  Weights are small: the gathers do not matter.
struct Strided4 { int d1, d2, d3; int64_t s0, s1, s2, s3; };
__global__ __launch_bounds__(256) void split_planes_strided_kernel(const float* __restrict__ x, Strided4 v, int64_t n4, PlaneOut o) {
  const PlaneDst d = plane_dst(o);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int q3 = v.d3 >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int i3 = (int)(i % q3) * 4;
    int64_t r = i / q3;
    const int i2 = (int)(r % v.d2); r /= v.d2;
    const int i1 = (int)(r % v.d1);
    const int64_t i0 = r / v.d1;
    const float* p = x + i0 * v.s0 + i1 * v.s1 + i2 * v.s2 + (int64_t)i3 * v.s3;
    store_planes4(d, (size_t)i * 4, make_float4(p[0], p[v.s3], p[2 * v.s3], p[3 * v.s3]));
  }
}

// -------------------------------------------------------------------------------------
// Small fixed-order column reductions.  Shape of all three: block = 256 threads = 16 columns
// x 16 row-parts, grid = ceil(H/16): every thread sums a strided 1/16 of the partial rows
// (loads coalesce to 64 B per part), then the 16 parts are combined through LDS in a fixed
// order.  (A first version used 64 columns x 4 parts on 16 workgroups: 128 dependent-free
// loads per thread on 16 CUs took 13-31 us per call, 10 % of a step.)
// -------------------------------------------------------------------------------------
constexpr int RCOLS = 16, RPARTS = 16;

__device__ __forceinline__ float parts_sum(float v, float (*red)[RCOLS], int cl, int part) {
  red[part][cl] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int q = 0; q < RPARTS; ++q) t += red[q][cl];
  __syncthreads();
  return t;
}

// Pieces of the statistics finalize.  Floating-point contraction is OFF inside them: "SyncBN == one process on the
// concatenated batch, bit for bit" is a tested property, and it must not depend on what the compiler contracts.
__device__ __forceinline__ float bn_m2_term(float gsum, float gm2, float fn, float mean) {
#pragma clang fp contract(off)
  const float d = gsum / fn - mean;
  return gm2 + fn * d * d;
}
// (bn_shift_of, bn_running_update: bn_pieces.h)

// BatchNorm statistics finalize: Chan merge of (sum, M2) over 64-row groups.
// stat = [world][2][G][H] (rank-major: G rows of sums, then G rows of M2); Br rows per rank.  Walking
// the world*G partials in rank-major order is walking the concatenated batch's groups in row order,
// so the result is the one a single process gets on the concatenated batch (when Br % 128 == 0 even
// bit for bit: same partials, same order).
__global__ __launch_bounds__(NTHR) void bn_finalize_kernel(
    const float* __restrict__ stat, int G, int world, int Br, int H, int gs,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* running_mean, float* running_var, int64_t* batches, float* mean_out, float* rstd_out,
    float* scale_out, float* shift_out) {
  __shared__ float red[RPARTS][RCOLS];
  const int cl = threadIdx.x & (RCOLS - 1), part = threadIdx.x / RCOLS;
  const int c = blockIdx.x * RCOLS + cl;
  const bool ok = c < H;
  const int GT = G * world;
  const float Bt = (float)Br * (float)world;
  const size_t GH = (size_t)G * H;
  // This kernel is pure latency (16,384 threads): everything it will need is requested up front -- the partials of
  // both passes (the lifter has 4 per thread: held in registers) and the per-column parameters -- so that it pays
  // ONE memory round trip instead of four dependent ones (6.6 -> ~3.5 us per launch, five launches per step).
  constexpr int KEEP = 4;
  float ks[KEEP], km[KEEP];
  const bool keep = GT <= KEEP * RPARTS;
#pragma unroll
  for (int q = 0; q < KEEP; ++q) {
    const int g = part + q * RPARTS;
    ks[q] = km[q] = 0.f;
    if (ok && keep && g < GT) {
      const int r = g / G, gl = g - r * G;
      const size_t at = (size_t)r * 2 * GH + (size_t)gl * H + c;
      ks[q] = stat[at]; km[q] = stat[at + GH];
    }
  }
  float ga = 0.f, be = 0.f, rm0 = 0.f, rv0 = 0.f;
  if (part == 0 && ok) {
    ga = gamma[c]; be = beta[c];
    if (running_mean) { rm0 = running_mean[c]; rv0 = running_var[c]; }
  }
  float s = 0.f;
  if (ok) {
    if (keep) {
#pragma unroll
      for (int q = 0; q < KEEP; ++q) s += ks[q];         // (absent groups hold 0)
    } else {
      // (many groups -- the conv path's maps: up to 2048): eight loads in flight per trip, at clamped addresses so that no
      // branch sits between them; added in group order.  One load per trip made this kernel 29 us on those maps.
      for (int g0 = part; g0 < GT; g0 += 8 * RPARTS) {
        float sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int g = g0 + u * RPARTS, gg = min(g, GT - 1);
          const int r = gg / G, gl = gg - r * G;
          const float v = stat[(size_t)r * 2 * GH + (size_t)gl * H + c];
          sv[u] = g < GT ? v : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);     // (all eight requested before the first is waited for)
#pragma unroll
        for (int u = 0; u < 8; ++u) s += sv[u];
      }
    }
  }
  const float mean = parts_sum(s, red, cl, part) / Bt;
  float m2 = 0.f;
  if (ok) {
    if (keep) {
#pragma unroll
      for (int q = 0; q < KEEP; ++q) {
        const int g = part + q * RPARTS;
        if (g < GT) {
          const int gl = g % G;
          const int n = max(0, min(gs, Br - gl * gs));
          if (n > 0) m2 += bn_m2_term(ks[q], km[q], (float)n, mean);
        }
      }
    } else {
      for (int g0 = part; g0 < GT; g0 += 8 * RPARTS) {
        float sv[8], qv[8];
        int nv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int g = g0 + u * RPARTS, gg = min(g, GT - 1);
          const int r = gg / G, gl = gg - r * G;
          const size_t at = (size_t)r * 2 * GH + (size_t)gl * H + c;
          sv[u] = stat[at]; qv[u] = stat[at + GH];
          nv[u] = g < GT ? max(0, min(gs, Br - gl * gs)) : 0;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (nv[u] > 0) m2 += bn_m2_term(sv[u], qv[u], (float)nv[u], mean);
      }
    }
  }
  const float m2t = parts_sum(m2, red, cl, part);
  if (part == 0 && ok) {
    const float var = m2t / Bt;  // biased
    const float rstd = 1.0f / sqrtf(var + eps);
    const float sc = ga * rstd;
    mean_out[c] = mean;
    rstd_out[c] = rstd;
    scale_out[c] = sc;
    shift_out[c] = bn_shift_of(be, mean, sc);
    if (running_mean) {
      bn_running_update(rm0, rv0, mean, var, Bt, momentum);
      running_mean[c] = rm0; running_var[c] = rv0;
    }
  }
  if (batches && blockIdx.x == 0 && threadIdx.x == 0) batches[0] += 1;
}

// block-level combine of per-wave float4 column partials; wave 0 returns the total
__device__ __forceinline__ float4 combine4(float4 v, float4 (*sm)[64], int wave, int lane) {
  sm[wave][lane] = v;
  __syncthreads();
  float4 t = sm[0][lane];
  if (wave == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 u = sm[w][lane];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
  }
  __syncthreads();
  return t;
}

// Column statistics of a [B][H] matrix in the layout bn_finalize merges: per 64-row group the column sums
// and the sums of squares about the group mean (what the GEMM epilogue emits for the lifter; the conv path's
// BatchNorm2d over [B*H*W][C] computes them here).  grid = (ceil(H/256), ceil(B/64)).
// Narrow maps (C = 64, 128): the caller views [rows][C] as [rows/R][R*C] so that every lane of a 256-column strip
// works; virtual column vc is replica vc / Hc of real column vc % Hc, and the partials are written replica-major
// ([R][2][G][Hc]) -- exactly the [world][2][G][H] layout bn_finalize merges, with the replicas in the role of ranks.
// One pass over z: a workgroup's gs rows are taken 64 at a time, each wave holding its 16 rows of the sub-group in
// registers (16 loads in flight per lane) for the sum AND the squares about the sub-group mean; the sub-groups are
// merged pairwise in order (Chan et al.) into the one partial per gs rows.  (The first version re-read the group
// from memory for the second moment: 2x the traffic on maps far larger than the L2.)
__global__ __launch_bounds__(NTHR) void bn_colstats_kernel(const float* __restrict__ z, int B, int H, int gs, int Hc,
                                                           float* __restrict__ stat) {
  __shared__ float4 sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const bool active = c < H;
  const int r0 = blockIdx.y * gs, r1 = min(B, r0 + gs);
  float4 tsum = make_float4(0, 0, 0, 0), tm2 = tsum;       // merged so far (wave 0)
  float ndone = 0.f;
  for (int q0 = r0; q0 < r1; q0 += 64) {
    const int q1 = min(r1, q0 + 64);
    float4 v[16];
    float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = q0 + wave + 4 * i;
      v[i] = (active && r < q1) ? ld4(z + (size_t)r * H + c) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
    float4 t = combine4(s, sm, wave, lane);
    if (wave == 0) sm[0][lane] = t;
    __syncthreads();
    t = sm[0][lane];
    __syncthreads();
    const float nb = (float)(q1 - q0), inv = 1.0f / nb;
    const float4 mu = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    float4 m2 = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (q0 + wave + 4 * i < q1) {
        const float dx = v[i].x - mu.x, dy = v[i].y - mu.y, dz = v[i].z - mu.z, dw = v[i].w - mu.w;
        m2.x = fmaf(dx, dx, m2.x); m2.y = fmaf(dy, dy, m2.y); m2.z = fmaf(dz, dz, m2.z); m2.w = fmaf(dw, dw, m2.w);
      }
    const float4 q = combine4(m2, sm, wave, lane);
    if (ndone == 0.f) { tsum = t; tm2 = q; }
    else {
      const float w = ndone * nb / (ndone + nb), ia = 1.0f / ndone;
      const float dx = mu.x - tsum.x * ia, dy = mu.y - tsum.y * ia, dz = mu.z - tsum.z * ia, dw = mu.w - tsum.w * ia;
      tm2.x += q.x + dx * dx * w; tm2.y += q.y + dy * dy * w; tm2.z += q.z + dz * dz * w; tm2.w += q.w + dw * dw * w;
      tsum.x += t.x; tsum.y += t.y; tsum.z += t.z; tsum.w += t.w;
    }
    ndone += nb;
  }
  if (wave == 0 && active) {
    const int rep = c / Hc, cc = c - rep * Hc;
    const size_t GH = (size_t)gridDim.y * Hc;
    float* base = stat + (size_t)rep * 2 * GH + (size_t)blockIdx.y * Hc + cc;
    st4(base, tsum);
    st4(base + GH, tm2);
  }
}

// out = relu(a + b), bitmap of (out > 0): the residual join of a Bottleneck (Resnet.py:90-91) in training mode
__global__ __launch_bounds__(NTHR) void add_relu_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ out, uint64_t* __restrict__ bits, int B,
                                                        int H, PlaneOut po) {
  const PlaneDst pd = plane_dst(po);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x;
  const int c = strip * 256 + lane * 4;
  const bool active = c < H;
  const int wpr = ((H + 255) >> 8) * 4;
  for (int r = blockIdx.y * 4 + wave; r < B; r += gridDim.y * 4) {
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
      const float4 u = ld4(a + (size_t)r * H + c), v = ld4(b + (size_t)r * H + c);
      o[0] = u.x + v.x; o[1] = u.y + v.y; o[2] = u.z + v.z; o[3] = u.w + v.w;
    }
    uint64_t word = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool on = active && o[j] > 0.f;
      o[j] = on ? o[j] : 0.f;
      const uint64_t m = __ballot(on);
      if (lane == j) word = m;
    }
    if (lane < 4) bits[(size_t)r * wpr + strip * 4 + lane] = word;
    if (active) {
      const float4 v = make_float4(o[0], o[1], o[2], o[3]);
      st4(out + (size_t)r * H + c, v);
      if (pd.kind) store_planes4(pd, (size_t)r * H + c, v);
    }
  }
}

// dx = g where the bitmap says the forward output was positive, else 0 (backward of add_relu: both inputs get it)
__global__ __launch_bounds__(NTHR) void mask_by_bits_kernel(const float* __restrict__ g, const uint64_t* __restrict__ bits,
                                                            float* __restrict__ dx, int B, int H, const float* __restrict__ g2) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x;
  const int c = strip * 256 + lane * 4;
  if (c >= H) return;
  const int wpr = ((H + 255) >> 8) * 4;
  for (int r = blockIdx.y * 4 + wave; r < B; r += gridDim.y * 4) {
    float4 v = ld4(g + (size_t)r * H + c);
    if (g2) { const float4 u = ld4(g2 + (size_t)r * H + c); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    const uint64_t* bw = bits + (size_t)r * wpr + strip * 4;
    float4 d;
    d.x = ((bw[0] >> lane) & 1ull) ? v.x : 0.f;
    d.y = ((bw[1] >> lane) & 1ull) ? v.y : 0.f;
    d.z = ((bw[2] >> lane) & 1ull) ? v.z : 0.f;
    d.w = ((bw[3] >> lane) & 1ull) ? v.w : 0.f;
    st4(dx + (size_t)r * H + c, d);
  }
}

// -------------------------------------------------------------------------------------
// act = [resid +] dropout(relu(z*scale + shift)); bitmap of (positive & kept).
// grid = (ceil(H/256), gy), block = 256: wave w walks rows blockIdx.y*4+w, +4*gy, ...
// Bitmap layout: row r, 256-column strip q, word j (0..3): bit l <-> column 256q + 4l + j.
// -------------------------------------------------------------------------------------
// the row loop of bn_apply_kernel (sc, sh: the lane's four columns).
// (Round 2 tried folding the statistics finalize into this kernel -- every workgroup re-deriving mean / rstd of its
//  256-column strip from the 64 partials -- to drop the 6.6 us finalize launch per layer; likewise for the backward
//  pass.  Same-box A/B, B = 4096: 0.7095 ms per step unfused against 0.80 (64 rows per workgroup), 0.80 (32), 0.92
//  (128), 0.90 (16): the 128 KB prologue per workgroup costs more than the launch it removes.  Not kept.
//  Also tried: two rows per trip with both rows' loads issued first (here and in bn_bwd_dz_rows): 0.6543 / 0.6524 ms
//  against 0.6446 / 0.6446 for one row per trip, same box -- eight short waves per CU already overlap their round trips.
//  Tried again on the conv path's maps (up to 1 GB, far beyond the caches; bn_apply, bn_bwd_dz and bn_bwd_reduce with two rows'
//  operands AND bitmap words requested up front): Model_3D step at B = 256 92.9 / 92.8 ms against 91.9 / 91.6 ms for one row per
//  trip, interleaved on one box -- these passes are not short of requests in flight.)
// RESID (is there a residual / join operand) is a template parameter for the same reason as bn_bwd_dz_rows' BN: its load leaves
// with z's instead of after the wait for it.
// one row of bn_apply: the lane's four columns c .. c+3 of row r (v = z, rv = the residual / join operand)
template <bool RESID>
__device__ __forceinline__ void bn_apply_row(
    const int r, const int c, const bool active, const float4 v, const float4 rv, const float4 sc, const float4 sh, float* act,
    uint64_t* __restrict__ bits, int H, int mode, bool norelu, uint32_t thr, float kscale, uint32_t k0,
    uint32_t k1, uint32_t c3, uint32_t layer, const uint64_t* __restrict__ inject, const PlaneDst& pd, bool resid_first) {
  const int lane = threadIdx.x & 63;
  const int strip = c >> 8;
  const int wpr = ((H + 255) >> 8) * 4;
  const size_t off = (size_t)r * H + c;
  float y[4] = {0.f, 0.f, 0.f, 0.f};
  bool keep[4] = {true, true, true, true};
  if (active) {
    y[0] = fmaf(v.x, sc.x, sh.x); y[1] = fmaf(v.y, sc.y, sh.y);
    y[2] = fmaf(v.z, sc.z, sh.z); y[3] = fmaf(v.w, sc.w, sh.w);
    if (RESID && resid_first) { y[0] += rv.x; y[1] += rv.y; y[2] += rv.z; y[3] += rv.w; }
    if (mode == 1) {
      const uint64_t g = ((uint64_t)r * (uint64_t)H + (uint64_t)c) >> 2;
      const Philox4 u = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), layer, c3, k0, k1);
#pragma unroll
      for (int j = 0; j < 4; ++j) keep[j] = u.v[j] >= thr;
    } else if (mode == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) keep[j] = (inject[(size_t)r * wpr + strip * 4 + j] >> lane) & 1ull;
    } else if (mode == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) keep[j] = false;
    }
  }
  bool on[4];
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    on[j] = active && keep[j] && (norelu || y[j] > 0.f);
    o[j] = on[j] ? y[j] * kscale : 0.f;
  }
  uint64_t word = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t b = __ballot(on[j]);
    if (lane == j) word = b;
  }
  if (lane < 4) bits[(size_t)r * wpr + strip * 4 + lane] = word;
  if (active) {
    float4 out = make_float4(o[0], o[1], o[2], o[3]);
    if (RESID && !resid_first) { out.x += rv.x; out.y += rv.y; out.z += rv.z; out.w += rv.w; }
    if (act) st4_nt(act + off, out, pd.nt);
    if (pd.kind) store_planes4(pd, off, out);
  }
}

template <bool RESID>
__device__ __forceinline__ void bn_apply_rows(
    const float* __restrict__ z, const float4 sc, const float4 sh, const float* resid, float* act,
    uint64_t* __restrict__ bits, int B, int H, int mode, bool norelu, uint32_t thr, float kscale, uint32_t k0,
    uint32_t k1, uint32_t c3, uint32_t layer, const uint64_t* __restrict__ inject, const PlaneDst& pd,
    bool resid_first = false) {
  // resid_first: resid joins BEFORE the ReLU -- relu(bn(z) + resid), the Bottleneck's join (Resnet.py:90-91) -- instead of
  // after it (the lifter's block: x + relu(bn(z)))
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const bool active = c < H;
  for (int r = blockIdx.y * 4 + wave; r < B; r += gridDim.y * 4) {
    const size_t off = (size_t)r * H + c;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f), rv = v;
    if (active) {
      v = ld4(z + off);
      if constexpr (RESID) rv = ld4(resid + off);
    }
    bn_apply_row<RESID>(r, c, active, v, rv, sc, sh, act, bits, H, mode, norelu, thr, kscale, k0, k1, c3, layer, inject, pd,
                        resid_first);
  }
}

// The statistics finalize inside the apply launch (BnFin::stat != NULL; local statistics, at most 16 groups = 1,024 rows):
// every workgroup re-derives mean / rstd / scale / shift of its four columns per lane from the partials -- with <= 16 groups
// bn_finalize_kernel's part p holds group p alone, so its result is the sum over the groups in order, which is what is done
// here, bit for bit -- and the workgroups with blockIdx.y == 0 write mean, rstd and the running statistics.  At B = 4096 (64
// groups, 128 KB of partials per workgroup) this cost more than the launch it removes (round 2); at 128 ... 512 rows the
// prologue is 2 ... 8 float4 pairs per lane and the 4.9 us finalize launch is a twentieth of the step.
struct BnFin {
  const float* stat;            // [2][G][H]
  const float *gamma, *beta;
  float *running_mean, *running_var, *mean_out, *rstd_out;
  int64_t* batches;
  float eps, momentum;
  int G, Br, gs;
};

__global__ __launch_bounds__(NTHR) void bn_apply_kernel(
    const float* __restrict__ z, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* resid, float* act, uint64_t* __restrict__ bits, int B, int H, int mode,
    uint32_t thr, float kscale, uint32_t k0, uint32_t k1, uint32_t c3, uint32_t layer,
    const uint64_t* __restrict__ inject, int Hc, PlaneOut po, const uint64_t* __restrict__ step_dev,
    uint32_t seed_hi, BnFin fin) {
  const PlaneDst pd = plane_dst(po);
  if (step_dev) {
    // graph replay: the step number is base (baked into c3 / k1's slot as the low / high word) + the device counter
    const uint64_t step = (((uint64_t)k1 << 32) | c3) + step_dev[0];
    c3 = (uint32_t)step;
    k1 = seed_hi ^ (uint32_t)(step >> 32);
  }
  // Hc: real columns behind the H virtual ones (bn_colstats_kernel); Hc == H for the lifter
  // mode: 0 keep all, 1 philox, 2 injected bitmap, 3 drop all; + 8: no ReLU (BatchNorm alone; bitmap all ones)
  const bool norelu = (mode & 8) != 0, resid_first = (mode & 32) != 0;      // + 32: the residual joins before the ReLU
  mode &= 7;
  const int c = blockIdx.x * 256 + (threadIdx.x & 63) * 4;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (fin.stat) {                                           // (kernel-uniform; Hc == H on this route)
    if (c < H) {
      const size_t GH = (size_t)fin.G * H;
      float4 ks[16], km[16];
#pragma unroll
      for (int g = 0; g < 16; ++g)
        if (g < fin.G) { ks[g] = ld4(fin.stat + (size_t)g * H + c); km[g] = ld4(fin.stat + GH + (size_t)g * H + c); }
      const float4 ga = ld4(fin.gamma + c), be = ld4(fin.beta + c);
      const float Bt = (float)fin.Br;
      float s4[4] = {0.f, 0.f, 0.f, 0.f}, m4[4] = {0.f, 0.f, 0.f, 0.f}, mean4[4];
#pragma unroll
      for (int g = 0; g < 16; ++g)
        if (g < fin.G) { s4[0] += ks[g].x; s4[1] += ks[g].y; s4[2] += ks[g].z; s4[3] += ks[g].w; }
#pragma unroll
      for (int j = 0; j < 4; ++j) mean4[j] = s4[j] / Bt;
#pragma unroll
      for (int g = 0; g < 16; ++g)
        if (g < fin.G) {
          const int n = max(0, min(fin.gs, fin.Br - g * fin.gs));
          if (n > 0) {
            m4[0] += bn_m2_term(ks[g].x, km[g].x, (float)n, mean4[0]); m4[1] += bn_m2_term(ks[g].y, km[g].y, (float)n, mean4[1]);
            m4[2] += bn_m2_term(ks[g].z, km[g].z, (float)n, mean4[2]); m4[3] += bn_m2_term(ks[g].w, km[g].w, (float)n, mean4[3]);
          }
        }
      const float g4[4] = {ga.x, ga.y, ga.z, ga.w}, b4[4] = {be.x, be.y, be.z, be.w};
      float var4[4], rs4[4], sc4[4], sh4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        var4[j] = m4[j] / Bt;
        rs4[j] = 1.0f / sqrtf(var4[j] + fin.eps);
        sc4[j] = g4[j] * rs4[j];
        sh4[j] = bn_shift_of(b4[j], mean4[j], sc4[j]);
      }
      sc = make_float4(sc4[0], sc4[1], sc4[2], sc4[3]);
      sh = make_float4(sh4[0], sh4[1], sh4[2], sh4[3]);
      if (blockIdx.y == 0 && threadIdx.x < 64) {
        st4(fin.mean_out + c, make_float4(mean4[0], mean4[1], mean4[2], mean4[3]));
        st4(fin.rstd_out + c, make_float4(rs4[0], rs4[1], rs4[2], rs4[3]));
        if (fin.running_mean) {
          float4 rm = ld4(fin.running_mean + c), rv = ld4(fin.running_var + c);
          bn_running_update(rm.x, rv.x, mean4[0], var4[0], Bt, fin.momentum);
          bn_running_update(rm.y, rv.y, mean4[1], var4[1], Bt, fin.momentum);
          bn_running_update(rm.z, rv.z, mean4[2], var4[2], Bt, fin.momentum);
          bn_running_update(rm.w, rv.w, mean4[3], var4[3], Bt, fin.momentum);
          st4(fin.running_mean + c, rm); st4(fin.running_var + c, rv);
        }
      }
    }
    if (fin.batches && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) fin.batches[0] += 1;
  } else if (c < H && scale) { sc = ld4(scale + c % Hc); sh = ld4(shift + c % Hc); }
  if (resid) bn_apply_rows<true>(z, sc, sh, resid, act, bits, B, H, mode, norelu, thr, kscale, k0, k1, c3, layer, inject, pd, resid_first);
  else bn_apply_rows<false>(z, sc, sh, resid, act, bits, B, H, mode, norelu, thr, kscale, k0, k1, c3, layer, inject, pd, resid_first);
}

// -------------------------------------------------------------------------------------
// Small batches (B <= kBnSmallRows: the reference's own batch of 64, train_1.py:194): the three launches of a hidden layer's
// forward tail -- statistics finalize, apply -- and of its backward head -- pass 1, finalize, dz -- are launch latency and
// nothing else at this size (5 us each for 256 KB of data).  Here ONE workgroup owns a 256-column strip for ALL rows:
// statistics straight from z (no partials, no Chan merge: two passes over rows that sit in L2), then the apply loop itself.
// -------------------------------------------------------------------------------------
// every wave gets the strip total of a per-wave float4 partial (fixed order); NW waves
template <int NW>
__device__ __forceinline__ float4 allwaves4(float4 v, float4 (*sm)[64], int wave, int lane) {
  sm[wave][lane] = v;
  __syncthreads();
  float4 t = sm[0][lane];
#pragma unroll
  for (int w = 1; w < NW; ++w) {
    const float4 u = sm[w][lane];
    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
  }
  __syncthreads();
  return t;
}

// 512 threads: wave w holds rows w, w + 8, ... (RU of them: 8 for B <= 64, 16 for B <= 128) of the strip IN REGISTERS -- one
// memory round trip for the whole kernel: statistics, the apply and the bitmap all work on the registers.
// (First version: 256 threads streaming the rows twice for the statistics and a third time for the apply, four rows in
//  flight: 0.313 ms per step at B = 64 against 0.304 for the three separate launches -- a dozen dependent round trips in
//  four workgroups cost more than two launch boundaries.)
constexpr int kSmallThreads = 512;
template <int RU, bool RESID>
__global__ __launch_bounds__(kSmallThreads) void bn_small_fwd_kernel(
    const float* __restrict__ z, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* running_mean, float* running_var, int64_t* batches, float* __restrict__ mean_out, float* __restrict__ rstd_out,
    const float* resid, float* act, uint64_t* __restrict__ bits, int B, int H, int mode, uint32_t thr, float kscale,
    uint32_t k0, uint32_t k1, uint32_t c3, uint32_t layer, const uint64_t* __restrict__ inject,
    const uint64_t* __restrict__ step_dev, uint32_t seed_hi) {
  __shared__ float4 sm[8][64];
  if (step_dev) {
    const uint64_t step = (((uint64_t)k1 << 32) | c3) + step_dev[0];
    c3 = (uint32_t)step;
    k1 = seed_hi ^ (uint32_t)(step >> 32);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const bool active = c < H;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 zv[RU], rv[RESID ? RU : 1];
  float4 ga = zero, be = zero, rm = zero, rvar = zero;
#pragma unroll
  for (int u = 0; u < RU; ++u) {
    const int r = wave + 8 * u;
    const bool ok = active && r < B;
    zv[u] = ok ? ld4(z + (size_t)r * H + c) : zero;
    if constexpr (RESID) rv[u] = ok ? ld4(resid + (size_t)r * H + c) : zero;
  }
  if (active) {
    ga = ld4(gamma + c); be = ld4(beta + c);
    if (running_mean && wave == 0) { rm = ld4(running_mean + c); rvar = ld4(running_var + c); }
  }
  float4 s = zero;
#pragma unroll
  for (int u = 0; u < RU; ++u) { s.x += zv[u].x; s.y += zv[u].y; s.z += zv[u].z; s.w += zv[u].w; }
  s = allwaves4<8>(s, sm, wave, lane);
  const float Bt = (float)B;
  const float4 mean = make_float4(s.x / Bt, s.y / Bt, s.z / Bt, s.w / Bt);
  float4 q = zero;
#pragma unroll
  for (int u = 0; u < RU; ++u) {
    if (wave + 8 * u >= B) continue;
    const float dx = zv[u].x - mean.x, dy = zv[u].y - mean.y, dz = zv[u].z - mean.z, dw = zv[u].w - mean.w;
    q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
  }
  q = allwaves4<8>(q, sm, wave, lane);
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero;
  if (active) {
    const float var[4] = {q.x / Bt, q.y / Bt, q.z / Bt, q.w / Bt};
    const float mu[4] = {mean.x, mean.y, mean.z, mean.w};
    const float g4[4] = {ga.x, ga.y, ga.z, ga.w}, b4[4] = {be.x, be.y, be.z, be.w};
    float rs[4], scv[4], shv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rs[j] = 1.0f / sqrtf(var[j] + eps);
      scv[j] = g4[j] * rs[j];
      shv[j] = bn_shift_of(b4[j], mu[j], scv[j]);
    }
    sc = make_float4(scv[0], scv[1], scv[2], scv[3]);
    sh = make_float4(shv[0], shv[1], shv[2], shv[3]);
    if (wave == 0) {
      st4(mean_out + c, mean);
      st4(rstd_out + c, make_float4(rs[0], rs[1], rs[2], rs[3]));
      if (running_mean) {
        bn_running_update(rm.x, rvar.x, mu[0], var[0], Bt, momentum);
        bn_running_update(rm.y, rvar.y, mu[1], var[1], Bt, momentum);
        bn_running_update(rm.z, rvar.z, mu[2], var[2], Bt, momentum);
        bn_running_update(rm.w, rvar.w, mu[3], var[3], Bt, momentum);
        st4(running_mean + c, rm); st4(running_var + c, rvar);
      }
    }
  }
  if (batches && blockIdx.x == 0 && threadIdx.x == 0) batches[0] += 1;
  const bool norelu = (mode & 8) != 0;
  const PlaneDst pd = {nullptr, nullptr, 1.f, 0, 0};
#pragma unroll
  for (int u = 0; u < RU; ++u) {
    const int r = wave + 8 * u;
    if (r >= B) break;                                 // (wave-uniform)
    bn_apply_row<RESID>(r, c, active, zv[u], RESID ? rv[u] : zero, sc, sh, act, bits, H, mode & 7, norelu, thr, kscale, k0, k1,
                        c3, layer, inject, pd, false);
  }
}

// backward head of one hidden layer for a small batch: pass 1 (sum dy, sum dy zhat), the coefficients, dz = c0 (dy - c1 -
// zhat c2), the bias gradient and dgamma / dbeta -- one workgroup per strip, the rows (masked dy and zhat) in registers
// TILE: the bitmap is in the tile format of small_layer.hip (the layer's forward ran there)
template <int RU, bool TILE>
__global__ __launch_bounds__(kSmallThreads) void bn_small_bwd_kernel(
    const float* __restrict__ g, const uint64_t* __restrict__ bits, const float* __restrict__ z,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma, float kscale, int B, int H,
    float* __restrict__ dz, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias) {
  __shared__ float4 sm[8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x;
  const int c = strip * 256 + lane * 4;
  const bool active = c < H;
  const int wpr = ((H + 255) >> 8) * 4;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 mu = zero, rs = zero, ga = zero;
  float4 dv[RU], zh[RU];                                // masked, scaled dy and zhat of the wave's rows
  {
    float4 gv[RU], zv[RU];
    ulonglong2 b01[RU], b23[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int r = min(wave + 8 * u, B - 1);
      gv[u] = active ? ld4(g + (size_t)r * H + c) : zero;
      zv[u] = active ? ld4(z + (size_t)r * H + c) : zero;
      const uint64_t* bw = TILE ? bits + (size_t)(c >> 4) * 16 + (r >> 4) * 4 : bits + (size_t)r * wpr + strip * 4;
      b01[u] = *reinterpret_cast<const ulonglong2*>(bw);
      b23[u] = *reinterpret_cast<const ulonglong2*>(bw + 2);
    }
    if (active) { mu = ld4(mean + c); rs = ld4(rstd + c); ga = ld4(gamma + c); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const bool ok = wave + 8 * u < B;
      const int bit = TILE ? ((min(wave + 8 * u, B - 1) & 15) * 4 + (lane & 3)) : lane;
      dv[u].x = (ok && ((b01[u].x >> bit) & 1ull)) ? gv[u].x * kscale : 0.f;
      dv[u].y = (ok && ((b01[u].y >> bit) & 1ull)) ? gv[u].y * kscale : 0.f;
      dv[u].z = (ok && ((b23[u].x >> bit) & 1ull)) ? gv[u].z * kscale : 0.f;
      dv[u].w = (ok && ((b23[u].y >> bit) & 1ull)) ? gv[u].w * kscale : 0.f;
      zh[u] = ok ? make_float4((zv[u].x - mu.x) * rs.x, (zv[u].y - mu.y) * rs.y, (zv[u].z - mu.z) * rs.z, (zv[u].w - mu.w) * rs.w) : zero;
    }
  }
  float4 s1 = zero, s2 = zero, sz = zero;
#pragma unroll
  for (int u = 0; u < RU; ++u) {
    s1.x += dv[u].x; s1.y += dv[u].y; s1.z += dv[u].z; s1.w += dv[u].w;
    s2.x = fmaf(dv[u].x, zh[u].x, s2.x); s2.y = fmaf(dv[u].y, zh[u].y, s2.y);
    s2.z = fmaf(dv[u].z, zh[u].z, s2.z); s2.w = fmaf(dv[u].w, zh[u].w, s2.w);
    sz.x += zh[u].x; sz.y += zh[u].y; sz.z += zh[u].z; sz.w += zh[u].w;
  }
  s1 = allwaves4<8>(s1, sm, wave, lane);
  s2 = allwaves4<8>(s2, sm, wave, lane);
  sz = allwaves4<8>(sz, sm, wave, lane);
  const float Bt = (float)B;
  const float4 c0 = make_float4(ga.x * rs.x, ga.y * rs.y, ga.z * rs.z, ga.w * rs.w);
  const float4 c1 = make_float4(s1.x / Bt, s1.y / Bt, s1.z / Bt, s1.w / Bt);
  const float4 c2 = make_float4(s2.x / Bt, s2.y / Bt, s2.z / Bt, s2.w / Bt);
  if (wave == 0 && active) {
    st4(dgamma + c, s2); st4(dbeta + c, s1);
    // bias gradient = sum_r dz_r = c0 (sum d - B c1 - c2 sum zhat): its true value is 0 (a bias in front of BatchNorm); formed
    // from the three sums it carries a few ulps of them, where the sum of B rounded dz values carries sqrt(B) ulps of |dz|
    float4 db;
    db.x = c0.x * ((s1.x - Bt * c1.x) - c2.x * sz.x); db.y = c0.y * ((s1.y - Bt * c1.y) - c2.y * sz.y);
    db.z = c0.z * ((s1.z - Bt * c1.z) - c2.z * sz.z); db.w = c0.w * ((s1.w - Bt * c1.w) - c2.w * sz.w);
    st4(dbias + c, db);
  }
  if (!active) return;
#pragma unroll
  for (int u = 0; u < RU; ++u) {
    const int r = wave + 8 * u;
    if (r >= B) break;
    float4 d;
    d.x = c0.x * (dv[u].x - c1.x - zh[u].x * c2.x); d.y = c0.y * (dv[u].y - c1.y - zh[u].y * c2.y);
    d.z = c0.z * (dv[u].z - c1.z - zh[u].z * c2.z); d.w = c0.w * (dv[u].w - c1.w - zh[u].w * c2.w);
    st4(dz + (size_t)r * H + c, d);
  }
}

// -------------------------------------------------------------------------------------
// BN backward pass 1: per-block partial column sums of dy and dy*zhat.
// -------------------------------------------------------------------------------------
// join_g2 / join_dx (the Bottleneck's bn3 + residual join, pl_bn_join_bwd): the incoming gradient is g + join_g2 (join_g2 may be
// NULL) and the masked value is also stored to join_dx -- the residual join's backward (mask_by_bits_kernel) folded into this pass.
__global__ __launch_bounds__(NTHR) void bn_bwd_reduce_kernel(
    const float* __restrict__ g, const uint64_t* __restrict__ bits, const float* __restrict__ z,
    const float* __restrict__ mean, const float* __restrict__ rstd, float kscale, int B, int H,
    float* __restrict__ part_dy, float* __restrict__ part_dyz, int Hc, float* __restrict__ part_amax,
    const float* __restrict__ join_g2, float* __restrict__ join_dx) {
  __shared__ float4 sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x;
  const int c = strip * 256 + lane * 4;
  const bool active = c < H;
  const int wpr = ((H + 255) >> 8) * 4;
  float4 mu = make_float4(0, 0, 0, 0), rs = mu, s1 = mu, s2 = mu;
  float mxd = 0.f, mxz = 0.f;                     // max |dy|, max |zhat| seen by this lane (part_amax)
  if (active) { mu = ld4(mean + c % Hc); rs = ld4(rstd + c % Hc); }
  for (int r = blockIdx.y * 4 + wave; r < B; r += gridDim.y * 4) {
    if (!active) continue;
    const size_t off = (size_t)r * H + c;
    float4 gv = ld4(g + off);
    const float4 zv = ld4(z + off);
    if (join_g2) { const float4 u = ld4(join_g2 + off); gv.x += u.x; gv.y += u.y; gv.z += u.z; gv.w += u.w; }
    const uint64_t* bw = bits + (size_t)r * wpr + strip * 4;
    const float d0 = ((bw[0] >> lane) & 1ull) ? gv.x * kscale : 0.f;
    const float d1 = ((bw[1] >> lane) & 1ull) ? gv.y * kscale : 0.f;
    const float d2 = ((bw[2] >> lane) & 1ull) ? gv.z * kscale : 0.f;
    const float d3 = ((bw[3] >> lane) & 1ull) ? gv.w * kscale : 0.f;
    if (join_dx) st4(join_dx + off, make_float4(d0, d1, d2, d3));
    s1.x += d0; s1.y += d1; s1.z += d2; s1.w += d3;
    const float z0 = (zv.x - mu.x) * rs.x, z1 = (zv.y - mu.y) * rs.y, z2 = (zv.z - mu.z) * rs.z, z3 = (zv.w - mu.w) * rs.w;
    s2.x = fmaf(d0, z0, s2.x);
    s2.y = fmaf(d1, z1, s2.y);
    s2.z = fmaf(d2, z2, s2.z);
    s2.w = fmaf(d3, z3, s2.w);
    if (part_amax) {
      mxd = fmaxf(fmaxf(mxd, fmaxf(fabsf(d0), fabsf(d1))), fmaxf(fabsf(d2), fabsf(d3)));
      mxz = fmaxf(fmaxf(mxz, fmaxf(fabsf(z0), fabsf(z1))), fmaxf(fabsf(z2), fabsf(z3)));
    }
  }
  const float4 t1 = combine4(s1, sm, wave, lane);
  const float4 t2 = combine4(s2, sm, wave, lane);
  if (part_amax) {                                 // wave-uniform; one pair per workgroup, no atomics
    sm[wave][lane] = make_float4(mxd, mxz, 0.f, 0.f);
    __syncthreads();
    if (wave == 0) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { a = fmaxf(a, sm[w][lane].x); b = fmaxf(b, sm[w][lane].y); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { a = fmaxf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); }
      if (lane == 0) {
        float* q = part_amax + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 2;
        q[0] = a; q[1] = b;
      }
    }
    __syncthreads();
  }
  if (wave == 0 && active) {
    // replica-major when Hc < H: [R][2][RC][Hc] (part_dyz = part_dy + RC*Hc); plain [2][RC][H] otherwise
    const int rep = c / Hc, cc = c - rep * Hc;
    const size_t off = (size_t)rep * 2 * gridDim.y * Hc + (size_t)blockIdx.y * Hc + cc;
    st4(part_dy + off, t1);
    st4(part_dyz + off, t2);
  }
}

// part = [world][2][RC][H] (rank-major: RC rows of sum dy, then RC rows of sum dy*zhat).  The dz
// coefficients use every rank's partials (the statistics were global, so is their gradient); dgamma
// and dbeta are this rank's own sums -- the gradient all-reduce adds the other ranks'.
__global__ __launch_bounds__(NTHR) void bn_bwd_finalize_kernel(
    const float* __restrict__ part_all, int RC, int world, int rank, int Br, int H,
    const float* __restrict__ gamma, const float* __restrict__ rstd, float* __restrict__ coef,
    float* __restrict__ dgamma, float* __restrict__ dbeta, const float* __restrict__ part_amax, int n_amax,
    float* __restrict__ dz_scale, int eval_mode, long long rstride, int amax_world) {
  // rstride: floats between two ranks' slabs (0: the dense 2 RC H); amax_world > 1: every rank's slab carries its n_amax
  // {max|dy|, max|zhat|} pairs behind its partial sums (part_amax points at rank 0's) -- under SyncBN the batch means in dz
  // are global, so the range bound of dz needs the GLOBAL maxima
  __shared__ float red[RPARTS][RCOLS];
  if (dz_scale && blockIdx.x == gridDim.x - 1) {
    // the extra workgroup: range bound of dz = c0 (dy - c1 - zhat c2), |c1| <= max|dy|, |c2| <= max|dy| (mean |zhat| <= 1):
    //   |dz| <= max|c0| max|dy| (2 + max|zhat|)  ->  S = the power of two that maps the bound into (2^13, 2^14]
    float a = 0.f, b = 0.f, c0 = 0.f;
    // (this one workgroup is the kernel's critical path: its loads go out eight at a time, at clamped indices -- a maximum does
    //  not care about repeats -- instead of one dependent round trip per trip)
    for (int r = 0; r < max(amax_world, 1); ++r) {
      const float2* pa = reinterpret_cast<const float2*>(part_amax + (size_t)r * rstride);
      for (int k0 = threadIdx.x; k0 < n_amax; k0 += 8 * NTHR) {
        float2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = pa[min(k0 + u * NTHR, n_amax - 1)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a = fmaxf(a, t[u].x); b = fmaxf(b, t[u].y); }
      }
    }
    for (int k0 = threadIdx.x; k0 < H; k0 += 4 * NTHR) {
      float gv[4], rv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int k = min(k0 + u * NTHR, H - 1); gv[u] = gamma[k]; rv[u] = rstd[k]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) c0 = fmaxf(c0, fabsf(gv[u] * rv[u]));
    }
    float* sm = &red[0][0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a = fmaxf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); c0 = fmaxf(c0, __shfl_xor(c0, o));
    }
    if (lane == 0) { sm[wave * 3] = a; sm[wave * 3 + 1] = b; sm[wave * 3 + 2] = c0; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w) { a = fmaxf(a, sm[w * 3]); b = fmaxf(b, sm[w * 3 + 1]); c0 = fmaxf(c0, sm[w * 3 + 2]); }
      const float bound = c0 * a * (2.0f + b);
      int e = 0;
      float S = 1.0f, Si = 1.0f;
      if (bound > 0.f && bound < 3.0e38f) {          // zero / inf / nan gradients: scale 1 (they stay what they are)
        (void)frexpf(bound, &e);                     // bound = m 2^e, 0.5 <= m < 1
        e = min(max(14 - e, -100), 100);
        S = ldexpf(1.0f, e); Si = ldexpf(1.0f, -e);
      }
      dz_scale[0] = S; dz_scale[1] = Si;
    }
    return;
  }
  const int cl = threadIdx.x & (RCOLS - 1), part = threadIdx.x / RCOLS;
  const int c = blockIdx.x * RCOLS + cl;
  const bool ok = c < H;
  const size_t RH = (size_t)RC * H;
  const size_t RS = rstride > 0 ? (size_t)rstride : 2 * RH;
  // rank < 0: the "ranks" are replicas of a narrow map (bn_colstats_kernel): every partial is this process's own
  const float* mine = part_all + (size_t)max(rank, 0) * RS;
  float gac = 0.f, rsc = 0.f;               // requested with the partials: one memory round trip, not two
  if (part == 0 && ok) { gac = gamma[c]; rsc = rstd[c]; }
  float a = 0.f, b = 0.f;
  if (ok && rank >= 0)                    // a replica view only needs the totals below
    for (int k0 = part; k0 < RC; k0 += 8 * RPARTS) {        // eight row-chunks' loads in flight (as bn_finalize_kernel)
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u * RPARTS, kk = min(k, RC - 1);
        const float x = mine[(size_t)kk * H + c], y = mine[RH + (size_t)kk * H + c];
        av[u] = k < RC ? x : 0.f; bv[u] = k < RC ? y : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += av[u]; b += bv[u]; }
    }
  const float sdy = parts_sum(a, red, cl, part);
  const float sdyz = parts_sum(b, red, cl, part);
  float tdy = sdy, tdyz = sdyz;
  if (world > 1) {                        // fixed rank-major order: every rank computes the same totals
    a = 0.f; b = 0.f;
    if (ok)
      for (int k0 = part; k0 < RC * world; k0 += 8 * RPARTS) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + u * RPARTS, kk = min(k, RC * world - 1);
          const int r = kk / RC, kl = kk - r * RC;
          const float x = part_all[(size_t)r * RS + (size_t)kl * H + c], y = part_all[(size_t)r * RS + RH + (size_t)kl * H + c];
          av[u] = k < RC * world ? x : 0.f; bv[u] = k < RC * world ? y : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a += av[u]; b += bv[u]; }
      }
    tdy = parts_sum(a, red, cl, part);
    tdyz = parts_sum(b, red, cl, part);
  }
  if (part == 0 && ok) {
    const float Bt = (float)Br * (float)world;
    coef[c] = gac * rsc;
    // eval mode (running statistics are constants of the graph): dz = gamma rstd dy, no batch-mean terms
    coef[H + c] = eval_mode ? 0.f : tdy / Bt;
    coef[2 * H + c] = eval_mode ? 0.f : tdyz / Bt;
    dgamma[c] = rank < 0 ? tdyz : sdyz;
    dbeta[c] = rank < 0 ? tdy : sdy;
  }
}

// out[c] = sum_r part[r][c]: the column-parallel form of reduce_slabs for many short slabs
__global__ __launch_bounds__(NTHR) void reduce_rows_kernel(const float* __restrict__ part, int R, int H,
                                                           float* __restrict__ out) {
  __shared__ float red[RPARTS][RCOLS];
  const int cl = threadIdx.x & (RCOLS - 1), p = threadIdx.x / RCOLS;
  const int c = blockIdx.x * RCOLS + cl;
  float a = 0.f;
  if (c < H)
#pragma unroll 8
    for (int k = p; k < R; k += RPARTS) a += part[(size_t)k * H + c];
  const float t = parts_sum(a, red, cl, p);
  if (p == 0 && c < H) out[c] = t;
}

// -------------------------------------------------------------------------------------
// BN backward pass 2: dz = c0*(dy - c1 - zhat*c2) (or dz = dy without BN) + db partials.
// -------------------------------------------------------------------------------------
// the row loop of bn_bwd_dz_kernel
// BN is a template parameter so that the z load sits in the same basic block as the g load: behind a run-time `if (bn)` it was
// issued only after the wait for g -- two dependent round trips per row.
template <bool BN>
__device__ __forceinline__ void bn_bwd_dz_rows(
    const float* __restrict__ g, const uint64_t* __restrict__ bits, const float* __restrict__ z, const float4 mu,
    const float4 rs, const float4 c0, const float4 c1, const float4 c2, float kscale, int B, int H,
    float* __restrict__ dz, float* __restrict__ part_db, const PlaneDst& pd, float4 (*sm)[64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x;
  const int c = strip * 256 + lane * 4;
  const bool active = c < H;
  const int wpr = ((H + 255) >> 8) * 4;
  float4 sdb = make_float4(0, 0, 0, 0);
  for (int r = blockIdx.y * 4 + wave; r < B; r += gridDim.y * 4) {
    if (!active) continue;
    const size_t off = (size_t)r * H + c;
    const float4 gv = ld4(g + off);
    float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (BN) zv = ld4(z + off);
    const uint64_t* bw = bits + (size_t)r * wpr + strip * 4;
    float4 d;
    d.x = ((bw[0] >> lane) & 1ull) ? gv.x * kscale : 0.f;
    d.y = ((bw[1] >> lane) & 1ull) ? gv.y * kscale : 0.f;
    d.z = ((bw[2] >> lane) & 1ull) ? gv.z * kscale : 0.f;
    d.w = ((bw[3] >> lane) & 1ull) ? gv.w * kscale : 0.f;
    if constexpr (BN) {
      d.x = c0.x * (d.x - c1.x - (zv.x - mu.x) * rs.x * c2.x);
      d.y = c0.y * (d.y - c1.y - (zv.y - mu.y) * rs.y * c2.y);
      d.z = c0.z * (d.z - c1.z - (zv.z - mu.z) * rs.z * c2.z);
      d.w = c0.w * (d.w - c1.w - (zv.w - mu.w) * rs.w * c2.w);
    }
    if (dz) st4_nt(dz + off, d, pd.nt);
    if (pd.kind) store_planes4(pd, off, d);
    sdb.x += d.x; sdb.y += d.y; sdb.z += d.z; sdb.w += d.w;
  }
  const float4 t = combine4(sdb, sm, wave, lane);
  if (wave == 0 && active) st4(part_db + (size_t)blockIdx.y * H + c, t);
}

__global__ __launch_bounds__(NTHR) void bn_bwd_dz_kernel(
    const float* __restrict__ g, const uint64_t* __restrict__ bits, const float* __restrict__ z,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ coef,
    float kscale, int bn, int B, int H, float* __restrict__ dz, float* __restrict__ part_db, int Hc, PlaneOut po) {
  __shared__ float4 sm[4][64];
  const PlaneDst pd = plane_dst(po);
  const int c = blockIdx.x * 256 + (threadIdx.x & 63) * 4;
  float4 zero = make_float4(0, 0, 0, 0);
  float4 mu = zero, rs = zero, c0 = zero, c1 = zero, c2 = zero;
  if (c < H && bn) {
    const int cc = c % Hc;
    mu = ld4(mean + cc); rs = ld4(rstd + cc);
    c0 = ld4(coef + cc); c1 = ld4(coef + Hc + cc); c2 = ld4(coef + 2 * Hc + cc);
  }
  if (bn) bn_bwd_dz_rows<true>(g, bits, z, mu, rs, c0, c1, c2, kscale, B, H, dz, part_db, pd, sm);
  else bn_bwd_dz_rows<false>(g, bits, z, mu, rs, c0, c1, c2, kscale, B, H, dz, part_db, pd, sm);
}

// -------------------------------------------------------------------------------------
// small helpers
// -------------------------------------------------------------------------------------
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int nslab, int64_t n,
                                    float* __restrict__ out, int vec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const int64_t n4 = n >> 2;
    for (; i < n4; i += stride) {
      float4 a = ld4(slabs + 4 * i);
      int s = 1;
      for (; s + 4 <= nslab; s += 4) {                   // four slabs' loads in flight, added in slab order
        float4 b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = ld4(slabs + (size_t)(s + u) * n + 4 * i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
      }
      for (; s < nslab; ++s) {
        const float4 b = ld4(slabs + (size_t)s * n + 4 * i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      st4(out + 4 * i, a);
    }
  } else {
    for (; i < n; i += stride) {
      float a = slabs[i];
      for (int s = 1; s < nslab; ++s) a += slabs[(size_t)s * n + i];
      out[i] = a;
    }
  }
}

// out[r][c] = bias[c] + sum_s slabs[s][r][c]   (split-K forward of the skinny output layer)
__global__ void reduce_slabs_bias_kernel(const float* __restrict__ slabs, int nslab, int64_t n, int cols,
                                         const float* __restrict__ bias, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float a = bias ? bias[i % cols] : 0.f;
    for (int s = 0; s < nslab; ++s) a += slabs[(size_t)s * n + i];
    out[i] = a;
  }
}

// block = 4 row-lanes x 64 column-lanes; fixed-order combine through LDS
__global__ __launch_bounds__(NTHR) void colsum_partial_kernel(const float* __restrict__ X, int rows,
                                                              int cols, int rpc, float* __restrict__ part) {
  __shared__ float sm[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rpc, r1 = min(rows, r0 + rpc);
  for (int c0 = 0; c0 < cols; c0 += 64) {
    const int c = c0 + cl;
    float s = 0.f;
    if (c < cols)
      for (int r = r0 + rl; r < r1; r += 4) s += X[(size_t)r * cols + c];
    sm[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < cols) part[(size_t)blockIdx.x * cols + c] = (sm[0][cl] + sm[1][cl]) + (sm[2][cl] + sm[3][cl]);
    __syncthreads();
  }
}

// several independent reduce_rows problems in one launch: blockIdx.y = job
// Every fixed-order partial-sum combine of one backward range in ONE launch (blockIdx.y = job): the bias gradients
// (many partial rows, few columns: 16 columns x 16 row-parts per workgroup, as reduce_rows_kernel), the skinny
// layers' weight-gradient partials (transK: the transposed store of dW0), and the split-K slabs of the 1024-wide weight
// gradients (few slabs, 1M outputs: streamed as float4, slab by slab in order -- kind 1).
__device__ __forceinline__ float block_sum(float v, float* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  float t = 0.f;
  const int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; ++w) t += sm[w];
  __syncthreads();
  return t;
}

constexpr int kMaxRowJobs = 16;
struct RowJobs { const float* part[kMaxRowJobs]; float* out[kMaxRowJobs]; int R[kMaxRowJobs]; int H[kMaxRowJobs];
                 int kind[kMaxRowJobs]; int transK[kMaxRowJobs]; float inv_n; uint64_t* tick; };
__global__ __launch_bounds__(NTHR) void reduce_rows_multi_kernel(RowJobs j) {
  __shared__ float red[RPARTS][RCOLS];
  const int job = blockIdx.y;
  const int R = j.R[job], H = j.H[job];
  const float* __restrict__ part = j.part[job];
  if (j.kind[job] == 2) {                    // the MSE loss of the fused train step: mse_final_kernel's sum, bit for bit
    if (blockIdx.x) return;
    float acc = 0.f;
    for (int i = threadIdx.x; i < R; i += blockDim.x) acc += part[i];
    const float t = block_sum(acc, &red[0][0]);
    if (threadIdx.x == 0) {
      j.out[job][0] = t * j.inv_n;
      if (j.tick) j.tick[0] += 1;
    }
    return;
  }
  if (j.kind[job] == 1) {                    // H % 4 == 0, 16-byte aligned (checked on the host)
    const int n4 = H >> 2;
    if (R == 4) {
      // the four split-K slabs of a 1024-wide weight gradient: all four loads in flight, added in slab order (with the
      // slab count a run-time loop bound every load waited for the one before it)
      for (int i = blockIdx.x * NTHR + threadIdx.x; i < n4; i += gridDim.x * NTHR) {
        const float* __restrict__ q = part + 4 * (size_t)i;
        float4 a = ld4(q);
        const float4 b1 = ld4(q + (size_t)H), b2 = ld4(q + 2 * (size_t)H), b3 = ld4(q + 3 * (size_t)H);
        a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
        a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
        a.x += b3.x; a.y += b3.y; a.z += b3.z; a.w += b3.w;
        st4(j.out[job] + 4 * (size_t)i, a);
      }
      return;
    }
    for (int i = blockIdx.x * NTHR + threadIdx.x; i < n4; i += gridDim.x * NTHR) {
      float4 a = ld4(part + 4 * (size_t)i);
      for (int sl = 1; sl < R; ++sl) {
        const float4 b = ld4(part + (size_t)sl * H + 4 * (size_t)i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      st4(j.out[job] + 4 * (size_t)i, a);
    }
    return;
  }
  if (blockIdx.x * RCOLS >= H) return;
  const int cl = threadIdx.x & (RCOLS - 1), p = threadIdx.x / RCOLS;
  const int c = blockIdx.x * RCOLS + cl;
  float a = 0.f;
  if (c < H)
#pragma unroll 8
    for (int k = p; k < R; k += RPARTS) a += part[(size_t)k * H + c];
  const float t = parts_sum(a, red, cl, p);
  if (p == 0 && c < H) {
    const int K = j.transK[job];
    if (K > 0) {                             // c = k * (H / K) + col  ->  out[col][k]
      const int Hc = H / K, k = c / Hc, col = c - k * Hc;
      j.out[job][(size_t)col * K + k] = t;
    } else {
      j.out[job][c] = t;
    }
  }
}

__global__ void bn_fold_eval_kernel(const float* __restrict__ bias, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ rm,
                                    const float* __restrict__ rv, float eps, int bn, int H,
                                    float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= H) return;
  if (bn) {
    const float s = gamma[c] * (1.0f / sqrtf(rv[c] + eps));
    scale[c] = s;
    shift[c] = fmaf(bias[c] - rm[c], s, beta[c]);
  } else {
    scale[c] = 1.f;
    shift[c] = bias[c];
  }
}

// eval-mode BatchNorm as the training kernels see it: "batch" statistics := the running ones
__global__ void bn_eval_stats_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ rm, const float* __restrict__ rv, float eps, int H,
                                     float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ scale,
                                     float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= H) return;
  const float r = 1.0f / sqrtf(rv[c] + eps), sc = gamma[c] * r;
  mean[c] = rm[c]; rstd[c] = r; scale[c] = sc; shift[c] = bn_shift_of(beta[c], rm[c], sc);
}

__global__ void fill_kernel(float* p, int64_t n, float v) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// ---- MSE(mean) forward + backward ---------------------------------------------------------
__global__ __launch_bounds__(NTHR) void mse_partial_kernel(const float* __restrict__ pred,
                                                           const float* __restrict__ tgt, int64_t n,
                                                           float coef, float* __restrict__ dpred,
                                                           float* __restrict__ part) {
  __shared__ float sm[4];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = pred[i] - tgt[i];
    acc = fmaf(d, d, acc);
    if (dpred) dpred[i] = d * coef;
  }
  const float t = block_sum(acc, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// mse_partial_kernel with pred[i] formed on the fly from the output Linear's split-K slabs (part[s][r][64 padded columns]) and
// stored: y[i] = bias[n] + slabs in order -- skinny_narrow_out_reduce_kernel's sum -- then the same d, fmaf and block_sum on the
// same thread -> element map: y, dpred and the partials equal the two-launch route's bit for bit.  NS slabs, N columns.
template <int NS, int N>
__global__ __launch_bounds__(NTHR) void mse_partial_from_slabs_kernel(const float* __restrict__ part, int B,
                                                                      const float* __restrict__ bias,
                                                                      const float* __restrict__ tgt, float coef,
                                                                      float* __restrict__ y, float* __restrict__ dpred,
                                                                      float* __restrict__ mpart) {
  __shared__ float sm[4];
  const int64_t n = (int64_t)B * N;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const size_t slab = (size_t)B * 64;
  float acc = 0.f;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
    float u[4][NS], t[4], bs[4];
    int64_t idx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                       // up to four elements of this thread: every load issued first
      idx[q] = i0 + q * stride;
      const bool ok = idx[q] < n;
      const int r = ok ? (int)(idx[q] / N) : 0, c = ok ? (int)(idx[q] - (int64_t)r * N) : 0;
      const float* p = part + (size_t)r * 64 + c;
#pragma unroll
      for (int s = 0; s < NS; ++s) u[q][s] = ok ? p[(size_t)s * slab] : 0.f;
      t[q] = ok ? tgt[idx[q]] : 0.f;
      bs[q] = (ok && bias) ? bias[c] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (idx[q] >= n) continue;
      float a = bs[q];
#pragma unroll
      for (int s = 0; s < NS; ++s) a += u[q][s];
      y[idx[q]] = a;
      const float d = a - t[q];
      acc = fmaf(d, d, acc);
      dpred[idx[q]] = d * coef;
    }
  }
  const float tsum = block_sum(acc, sm);
  if (threadIdx.x == 0) mpart[blockIdx.x] = tsum;
}

__global__ __launch_bounds__(NTHR) void mse_final_kernel(const float* __restrict__ part, int np,
                                                         float inv_n, float* __restrict__ loss, uint64_t* tick) {
  __shared__ float sm[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < np; i += blockDim.x) acc += part[i];
  const float t = block_sum(acc, sm);
  if (threadIdx.x == 0) {
    loss[0] = t * inv_n;
    // graph replay: one training step is complete as far as its readers of the counter go (every dropout kernel of the
    // forward ran before this kernel, AdamW runs after it)
    if (tick) tick[0] += 1;
  }
}

// ---- L1(mean) terms, forward + backward, several (a, b) pairs per launch -----------------------
constexpr int L1_BLOCKS = 64;
struct L1Terms { PLL1Term t[PL_L1_MAX_TERMS]; };

__global__ __launch_bounds__(NTHR) void l1_partial_kernel(L1Terms T, float grad_scale, float* __restrict__ part) {
  __shared__ float sm[4];
  const PLL1Term q = T.t[blockIdx.y];
  const float coef = grad_scale / (float)q.n;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * NTHR + threadIdx.x; i < q.n; i += (int64_t)L1_BLOCKS * NTHR) {
    const float d = q.a[i] - q.b[i];
    acc += fabsf(d);
    const float g = d > 0.f ? coef : (d < 0.f ? -coef : 0.f);
    if (q.da) q.da[i] = g;
    if (q.db) q.db[i] = -g;
  }
  const float t = block_sum(acc, sm);
  if (threadIdx.x == 0) part[blockIdx.y * L1_BLOCKS + blockIdx.x] = t;
}

__global__ __launch_bounds__(64) void l1_final_kernel(const float* __restrict__ part, L1Terms T,
                                                      float* __restrict__ losses) {
  float v = part[blockIdx.x * L1_BLOCKS + threadIdx.x];       // L1_BLOCKS == 64 == one wavefront
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (threadIdx.x == 0) losses[blockIdx.x] = v / (float)T.t[blockIdx.x].n;
}

// ---- loss_MPJPE --------------------------------------------------------------------------
__global__ void mpjpe_partial_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                     int B, int J, int rpc, float* __restrict__ part) {
  const int r0 = blockIdx.x * rpc, r1 = min(B, r0 + rpc);
  for (int j = threadIdx.x; j < J; j += blockDim.x) {
    float s = 0.f;
    for (int r = r0; r < r1; ++r) {
      const size_t o = ((size_t)r * J + j) * 3;
      const float dx = pred[o] - tgt[o], dy = pred[o + 1] - tgt[o + 1], dz = pred[o + 2] - tgt[o + 2];
      s += sqrtf(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
    }
    part[(size_t)blockIdx.x * J + j] = s;
  }
}

__global__ void mpjpe_final_kernel(const float* __restrict__ part, int np, int J,
                                   float* __restrict__ metric) {
  for (int j = threadIdx.x; j < J; j += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < np; ++k) s += part[(size_t)k * J + j];
    metric[j] += s;
  }
}

// ---- flip_pose (phase3_direct/my_HybrIK/utils.py:372-396) -----------------------------------
// horizontal flip of (B, 17, D) poses: x -> 1-x (D = 2, image-normalised) or -x (D = 3), then
// left joints [4,5,6,11,12,13] <-> right joints [1,2,3,14,15,16]
__global__ void flip_pose_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int d = (int)(i % D);
  const int64_t bj = i / D;
  const int j = (int)(bj % 17);
  const int64_t b = bj / 17;
  // source joint of destination joint j
  const int src = (j >= 1 && j <= 3) ? j + 3 : (j >= 4 && j <= 6) ? j - 3
                : (j >= 11 && j <= 13) ? j + 3 : (j >= 14 && j <= 16) ? j - 3 : j;
  float v = in[(b * 17 + src) * D + d];
  if (d == 0) v = (D == 2) ? 1.0f - v : -v;
  out[i] = v;
}

// Flip test-time augmentation around ONE forward of 2B rows (eval-mode BatchNorm is row-wise, so the
// two passes the reference makes, train_5 copy.py:160-171 / train_1.py:128-134, are one batch):
//   pack : xx[0:B) = x, xx[B:2B) = flip(x)
//   merge: y = (yy[0:B) + flip(yy[B:2B))) / 2
__device__ __forceinline__ int flip_src_joint(int j) {
  return (j >= 1 && j <= 3) ? j + 3 : (j >= 4 && j <= 6) ? j - 3
       : (j >= 11 && j <= 13) ? j + 3 : (j >= 14 && j <= 16) ? j - 3 : j;
}

__global__ void flip_tta_pack_kernel(const float* __restrict__ x, float* __restrict__ xx, int64_t n, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int d = (int)(i % D);
  const int64_t bj = i / D;
  const int j = (int)(bj % 17);
  const int64_t b = bj / 17;
  float v = x[(b * 17 + flip_src_joint(j)) * D + d];
  if (d == 0) v = (D == 2) ? 1.0f - v : -v;
  xx[i] = x[i];
  xx[n + i] = v;
}

__global__ void flip_tta_merge_kernel(const float* __restrict__ yy, float* __restrict__ y, int64_t n, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int d = (int)(i % D);
  const int64_t bj = i / D;
  const int j = (int)(bj % 17);
  const int64_t b = bj / 17;
  float v = yy[n + (b * 17 + flip_src_joint(j)) * D + d];
  if (d == 0) v = (D == 2) ? 1.0f - v : -v;
  y[i] = (v + yy[i]) / 2.0f;
}

// The training-mode Flip branch of the phase5 cycle step (train_5 copy.py:174-199) composes flip_pose with an average:
//   y = (flip_pose(a) + b) / 2, and its backward  da = flip_pose'(g) / 2  (x negated, joints swapped: no 1 - x offset).
// out = (x_offset_if_d0 - / + in[src joint]) [+ addend] ) * scale in ONE pass.
__global__ void flip_pose_ex_kernel(const float* __restrict__ in, const float* __restrict__ addend, float* __restrict__ out,
                                    int64_t n, int D, float x_offset, float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int d = (int)(i % D);
  const int64_t bj = i / D;
  const int j = (int)(bj % 17);
  const int64_t b = bj / 17;
  float v = in[(b * 17 + flip_src_joint(j)) * D + d];
  if (d == 0) v = x_offset - v;
  if (addend) v += addend[i];
  out[i] = v * scale;
}

// torch.flip(frame, (W,)) of NHWC frames (train_5 copy.py:176 flips the NCHW frame along dim 3 = width): one element per
// thread, writes coalesced, reads reversed in groups of C
__global__ void flip_w_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, int W, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const int64_t p = i / C;
  const int w = (int)(p % W);
  const int64_t row = p / W;
  out[i] = in[(row * W + (W - 1 - w)) * C + c];
}

// ---- batch gather (data.py PoseFeeder): rows idx[0..n) of two resident row-major tables ----------
// oa[i][:] = a[idx[i]][:] (wa floats), ob[i][:] = b[idx[i]][:] (wb floats); one thread per output float
__global__ void gather_rows2_kernel(const float* __restrict__ a, int wa, const float* __restrict__ b, int wb,
                                    const int64_t* __restrict__ idx, int64_t n, float* __restrict__ oa,
                                    float* __restrict__ ob) {
  const int w = wa + wb;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * w) return;
  const int64_t i = t / w;
  const int c = (int)(t - i * w);
  const int64_t r = idx[i];
  if (c < wa) oa[i * wa + c] = a[r * wa + c];
  else ob[i * wb + (c - wa)] = b[r * wb + (c - wa)];
}

// ---- flat AdamW (torch single-tensor update order): AdamWIn, adamw_consts, adamw_one in adamw.h ---------
__global__ __launch_bounds__(NTHR) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v,
                                                     int64_t n, AdamWIn in, int vec) {
  const AdamWK k = adamw_consts(in);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const int64_t n4 = n >> 2;
    for (; i < n4; i += stride) {
      float4 pv = ld4(p + 4 * i), mv = ld4(m + 4 * i), vv = ld4(v + 4 * i);
      const float4 gv = ld4(g + 4 * i);
      adamw_one(pv.x, gv.x, mv.x, vv.x, k);
      adamw_one(pv.y, gv.y, mv.y, vv.y, k);
      adamw_one(pv.z, gv.z, mv.z, vv.z, k);
      adamw_one(pv.w, gv.w, mv.w, vv.w, k);
      st4(p + 4 * i, pv); st4(m + 4 * i, mv); st4(v + 4 * i, vv);
      if (in.nseg) {                      // segment starts and lengths are multiples of 4: a float4 lies inside one
        const int64_t e = 4 * i;
#pragma unroll 1
        for (int q = 0; q < in.nseg; ++q) {
          const int64_t r = e - in.seg_off[q];
          if (r >= 0 && r < in.seg_n[q]) {
            const PlaneDst d = {in.seg_h[q], in.seg_l[q], in.pscale, in.kind};
            store_planes4(d, (size_t)r, pv);
            break;
          }
        }
      }
    }
    i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  }
  for (; i < n; i += stride) adamw_one(p[i], g[i], m[i], v[i], k);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int stream_rows_grid(int B, int strips) {
  // ~2048 workgroups in flight (8 per CU), 4 rows per workgroup pass
  int gy = (2048 + strips - 1) / strips;
  const int need = (B + 3) / 4;
  if (gy > need) gy = need;
  return gy < 1 ? 1 : gy;
}

}  // namespace

// =====================================================================================
// launchers
// =====================================================================================
// (sum, M2) partials over 64-row groups (what a GEMM epilogue emits) merged F at a time, in group order (Chan et al.), into
// the [2][Gp][H] layout bn_finalize walks with group_rows = 64 F: a 1M-pixel map has 16384 such partials per column.
__global__ __launch_bounds__(NTHR) void bn_merge_groups_kernel(const float* __restrict__ in, int G, int rows, int H, int F,
                                                               float* __restrict__ out, int Gp) {
  const int c = blockIdx.x * NTHR + threadIdx.x;
  if (c >= H) return;
  const int gp = blockIdx.y;
  const size_t GH = (size_t)G * H;
  float S = 0.f, Q = 0.f, na = 0.f;
  for (int f0 = 0; f0 < F; f0 += 8) {
    float s[8], q[8], n[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {                       // eight groups' loads in flight
      const int g = gp * F + f0 + u;
      n[u] = (f0 + u < F && g < G) ? (float)max(0, min(64, rows - 64 * g)) : 0.f;
      s[u] = q[u] = 0.f;
      if (n[u] > 0.f) { s[u] = in[(size_t)g * H + c]; q[u] = in[GH + (size_t)g * H + c]; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (n[u] == 0.f) continue;
      if (na == 0.f) { S = s[u]; Q = q[u]; na = n[u]; continue; }
      const float d = s[u] / n[u] - S / na;
      Q += q[u] + d * d * (na * n[u] / (na + n[u]));
      S += s[u];
      na += n[u];
    }
  }
  out[(size_t)gp * H + c] = S;
  out[(size_t)Gp * H + (size_t)gp * H + c] = Q;
}

int launch_bn_finalize(const float* stat, int G, int world, int B, int H,
                       const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, int64_t* batches, float* mean,
                       float* rstd, float* scale, float* shift, hipStream_t s, int group_rows) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((H + RCOLS - 1) / RCOLS), dim3(NTHR), 0, s, stat, G, world, B,
                     H, group_rows, gamma, beta, eps, momentum, running_mean, running_var, batches, mean, rstd,
                     scale, shift);
  PL_CHECK_LAUNCH("bn_finalize");
  return PL_OK;
}

int launch_bn_apply(const float* z, const float* scale, const float* shift, const float* resid,
                    float* act, uint64_t* bits, int B, int H, float p, uint64_t seed, uint64_t step,
                    int layer, const uint64_t* inject_keep, hipStream_t s, const PlaneOut* planes,
                    const uint64_t* step_dev, const BnFinalizeArgs* finalize) {
  int mode = 0;
  PlaneOut po = planes ? *planes : PlaneOut{nullptr, nullptr, 1.f, nullptr, 0};
  po.nt = (nontemporal_on() && (int64_t)B * H * 4 >= kNontemporalBytes) ? 1 : 0;
  if (!act && !po.kind) PL_FAIL(PL_EINVAL, "bn_apply: nothing to write");
  float kscale = 1.f;
  if (p >= 1.f) mode = 3;
  else if (p > 0.f) {
    mode = inject_keep ? 2 : 1;
    kscale = 1.0f / (1.0f - p);
  }
  const uint32_t thr = dropout_threshold(p);
  // step_dev: the kernel forms the key itself from (step + *step_dev); it receives the step's two words instead
  const uint32_t k0 = (uint32_t)seed, seed_hi = (uint32_t)(seed >> 32);
  const uint32_t k1 = step_dev ? (uint32_t)(step >> 32) : seed_hi ^ (uint32_t)(step >> 32);
  const int strips = (H + 255) / 256;
  dim3 grid(strips, stream_rows_grid(B, strips));
  BnFin fin = {};
  if (finalize) {
    const BnFinalizeArgs& f = *finalize;
    if (!f.stat || !f.gamma || !f.beta || !f.mean || !f.rstd || f.G < 1 || f.G > 16 || (H & 3) ||
        (f.running_mean != nullptr) != (f.running_var != nullptr))
      PL_FAIL(PL_EINVAL, "bn_apply: statistics finalize (G=%d)", f.G);
    fin.stat = f.stat; fin.gamma = f.gamma; fin.beta = f.beta; fin.running_mean = f.running_mean; fin.running_var = f.running_var;
    fin.mean_out = f.mean; fin.rstd_out = f.rstd; fin.batches = f.batches; fin.eps = f.eps; fin.momentum = f.momentum;
    fin.G = f.G; fin.Br = B; fin.gs = f.group_rows;
  }
  hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(NTHR), 0, s, z, scale, shift, resid, act, bits, B, H,
                     mode, thr, kscale, k0, k1, (uint32_t)step, (uint32_t)layer, inject_keep, H, po, step_dev, seed_hi, fin);
  PL_CHECK_LAUNCH("bn_apply");
  return PL_OK;
}

int launch_bn_small_fwd(const float* z, const float* gamma, const float* beta, float eps, float momentum, float* rm, float* rv,
                        int64_t* nbt, float* mean, float* rstd, const float* resid, float* act, uint64_t* bits, int B, int H,
                        float p, uint64_t seed, uint64_t step, int layer, const uint64_t* inject_keep, hipStream_t s,
                        const uint64_t* step_dev) {
  if (!z || !act || !bits || !mean || !rstd || B < 2 || (H & 3)) PL_FAIL(PL_EINVAL, "bn_small_fwd: bad arguments");
  int mode = 0;
  float kscale = 1.f;
  if (p >= 1.f) mode = 3;
  else if (p > 0.f) { mode = inject_keep ? 2 : 1; kscale = 1.0f / (1.0f - p); }
  const uint32_t thr = dropout_threshold(p);
  const uint32_t k0 = (uint32_t)seed, seed_hi = (uint32_t)(seed >> 32);
  const uint32_t k1 = step_dev ? (uint32_t)(step >> 32) : seed_hi ^ (uint32_t)(step >> 32);
  if (B > kBnSmallRows) PL_FAIL(PL_ESHAPE, "bn_small_fwd: B=%d", B);
  const dim3 grid((H + 255) / 256), block(kSmallThreads);
#define PL_SMALL_FWD(RU, RES)                                                                                               \
  hipLaunchKernelGGL((bn_small_fwd_kernel<RU, RES>), grid, block, 0, s, z, gamma, beta, eps, momentum, rm, rv, nbt, mean, rstd,   \
                     resid, act, bits, B, H, mode, thr, kscale, k0, k1, (uint32_t)step, (uint32_t)layer, inject_keep, step_dev, \
                     seed_hi)
  if (resid) PL_SMALL_FWD(8, true); else PL_SMALL_FWD(8, false);
#undef PL_SMALL_FWD
  PL_CHECK_LAUNCH("bn_small_fwd");
  return PL_OK;
}

int launch_bn_small_bwd(const float* g, const uint64_t* bits, const float* z, const float* mean, const float* rstd,
                        const float* gamma, float keep_scale, int B, int H, float* dz, float* dgamma, float* dbeta, float* dbias,
                        hipStream_t s, bool tile_bits) {
  if (!g || !bits || !z || !dz || !dgamma || !dbeta || !dbias || B < 1 || (H & 3)) PL_FAIL(PL_EINVAL, "bn_small_bwd: bad arguments");
  if (B > kBnSmallRows) PL_FAIL(PL_ESHAPE, "bn_small_bwd: B=%d", B);
  const dim3 grid((H + 255) / 256), block(kSmallThreads);
  if (tile_bits && (H & 255)) PL_FAIL(PL_ESHAPE, "bn_small_bwd: tile-format bitmap with H=%d", H);
  if (tile_bits)
    hipLaunchKernelGGL((bn_small_bwd_kernel<8, true>), grid, block, 0, s, g, bits, z, mean, rstd, gamma, keep_scale, B, H, dz,
                       dgamma, dbeta, dbias);
  else
    hipLaunchKernelGGL((bn_small_bwd_kernel<8, false>), grid, block, 0, s, g, bits, z, mean, rstd, gamma, keep_scale, B, H, dz,
                       dgamma, dbeta, dbias);
  PL_CHECK_LAUNCH("bn_small_bwd");
  return PL_OK;
}

// Row chunks of the two backward streaming passes over a [B][H] matrix (grid.y; one partial row per chunk).
// 32 rows per chunk, capped so that strips x chunks stays near 1024 workgroups (4 per CU) and never below 128:
// a 256-column map of the conv path (one strip, 131072 rows) ran 128 workgroups on 256 CUs with the old flat cap;
// 2048 made the two streaming passes no faster and the finalize kernel (serial over the chunks) twice as slow.
int bwd_row_chunks(int B, int H) {
  const int strips = (H + 255) / 256;
  int cap = 1024 / strips;
  if (cap < 128) cap = 128;
  int rc = (B + 31) / 32;
  if (rc > cap) rc = cap;
  return rc < 1 ? 1 : rc;
}

int launch_bn_bwd_reduce(const float* g, const uint64_t* bits, const float* z, const float* mean,
                         const float* rstd, float keep_scale, int B, int H, float* part_dy,
                         float* part_dyz, hipStream_t s, int Hc, float* part_amax, int rc, const float* join_g2,
                         float* join_dx) {
  dim3 grid((H + 255) / 256, rc > 0 ? rc : bwd_row_chunks(B, H));
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(NTHR), 0, s, g, bits, z, mean, rstd, keep_scale, B,
                     H, part_dy, part_dyz, Hc > 0 ? Hc : H, part_amax, join_g2, join_dx);
  PL_CHECK_LAUNCH("bn_bwd_reduce");
  return PL_OK;
}

int launch_bn_bwd_finalize(const float* part, int RC, int world, int rank, int B, int H,
                           const float* gamma, const float* rstd, float* coef, float* dgamma,
                           float* dbeta, hipStream_t s, const float* part_amax, int n_amax, float* dz_scale,
                           int eval_mode, int64_t rstride, int amax_world) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((H + RCOLS - 1) / RCOLS + (dz_scale ? 1 : 0)), dim3(NTHR), 0, s, part,
                     RC, world, rank, B, H, gamma, rstd, coef, dgamma, dbeta, part_amax, n_amax, dz_scale, eval_mode,
                     (long long)rstride, amax_world);
  PL_CHECK_LAUNCH("bn_bwd_finalize");
  return PL_OK;
}

int launch_bn_bwd_dz(const float* g, const uint64_t* bits, const float* z, const float* mean,
                     const float* rstd, const float* coef, float keep_scale, int bn, int B, int H,
                     float* dz, float* part_db, hipStream_t s, int Hc, const PlaneOut* planes, int rc) {
  dim3 grid((H + 255) / 256, rc > 0 ? rc : bwd_row_chunks(B, H));
  PlaneOut po = planes ? *planes : PlaneOut{nullptr, nullptr, 1.f, nullptr, 0};
  po.nt = (nontemporal_on() && (int64_t)B * H * 4 >= kNontemporalBytes) ? 1 : 0;
  if (!dz && !po.kind) PL_FAIL(PL_EINVAL, "bn_bwd_dz: nothing to write");
  hipLaunchKernelGGL(bn_bwd_dz_kernel, grid, dim3(NTHR), 0, s, g, bits, z, mean, rstd, coef, keep_scale,
                     bn, B, H, dz, part_db, Hc > 0 ? Hc : H, po);
  PL_CHECK_LAUNCH("bn_bwd_dz");
  return PL_OK;
}

bool nontemporal_on() {
  static const bool on = [] { const char* e = getenv("POSELIFT_NT"); return !(e && e[0] == '0'); }();
  return on;
}

int launch_split_planes(const float* x, int64_t n, const PlaneOut& out, hipStream_t s) {
  if (!x || !out.h || (out.kind == 2 && !out.l) || out.kind < 1 || out.kind > 2) PL_FAIL(PL_EINVAL, "split_planes: bad arguments");
  if ((n & 3) || !aligned16(x) || (reinterpret_cast<uintptr_t>(out.h) & 7) || (reinterpret_cast<uintptr_t>(out.l) & 7))
    PL_FAIL(PL_EINVAL, "split_planes: n %% 4 and alignment");
  const int64_t n4 = n >> 2;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, s, x, n4, out);
  PL_CHECK_LAUNCH("split_planes");
  return PL_OK;
}

int launch_reduce_slabs(const float* slabs, int nslab, int64_t n, float* out, hipStream_t s) {
  if (nslab >= 8 && n <= 65536) {   // many short slabs: parallelise over the slab index too
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(((int)n + RCOLS - 1) / RCOLS), dim3(NTHR), 0, s, slabs, nslab,
                       (int)n, out);
    PL_CHECK_LAUNCH("reduce_rows");
    return PL_OK;
  }
  const int vec = ((n & 3) == 0) && aligned16(slabs) && aligned16(out);
  const int64_t work = vec ? (n >> 2) : n;
  int blocks = (int)((work + NTHR - 1) / NTHR);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(NTHR), 0, s, slabs, nslab, n, out, vec);
  PL_CHECK_LAUNCH("reduce_slabs");
  return PL_OK;
}

int launch_reduce_slabs_bias(const float* slabs, int nslab, int rows, int cols, const float* bias,
                             float* out, hipStream_t s) {
  const int64_t n = (int64_t)rows * cols;
  int blocks = (int)((n + NTHR - 1) / NTHR);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(reduce_slabs_bias_kernel, dim3(blocks), dim3(NTHR), 0, s, slabs, nslab, n, cols, bias, out);
  PL_CHECK_LAUNCH("reduce_slabs_bias");
  return PL_OK;
}

int launch_reduce_rows_multi(const float* const* part, const int* R, const int* H, float* const* out, int njobs,
                             hipStream_t s, const int* kind, const int* transK, float loss_inv_n, uint64_t* loss_tick) {
  for (int base = 0; base < njobs; base += kMaxRowJobs) {
    RowJobs j = {};
    j.inv_n = loss_inv_n; j.tick = loss_tick;
    const int n = njobs - base < kMaxRowJobs ? njobs - base : kMaxRowJobs;
    int gx = 1;
    for (int k = 0; k < n; ++k) {
      j.part[k] = part[base + k]; j.out[k] = out[base + k]; j.R[k] = R[base + k]; j.H[k] = H[base + k];
      j.kind[k] = kind ? kind[base + k] : 0;
      j.transK[k] = transK ? transK[base + k] : 0;
      int need;
      if (j.kind[k] == 1) {
        if ((j.H[k] & 3) || !aligned16(j.part[k]) || !aligned16(j.out[k])) PL_FAIL(PL_EINVAL, "reduce_rows_multi: slab job alignment");
        need = ((j.H[k] >> 2) + NTHR - 1) / NTHR;
        if (need > 1024) need = 1024;
      } else if (j.kind[k] == 2) {
        need = 1;
      } else {
        need = (j.H[k] + RCOLS - 1) / RCOLS;
      }
      if (need > gx) gx = need;
    }
    hipLaunchKernelGGL(reduce_rows_multi_kernel, dim3(gx, n), dim3(NTHR), 0, s, j);
    PL_CHECK_LAUNCH("reduce_rows_multi");
  }
  return PL_OK;
}

int colsum_chunks(int rows) {
  int rc = (rows + 63) / 64;
  if (rc > 64) rc = 64;
  return rc < 1 ? 1 : rc;
}

int launch_colsum_partial(const float* X, int rows, int cols, float* part, hipStream_t s) {
  const int rc = colsum_chunks(rows);
  const int rpc = (rows + rc - 1) / rc;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(rc), dim3(NTHR), 0, s, X, rows, cols, rpc, part);
  PL_CHECK_LAUNCH("colsum_partial");
  return PL_OK;
}

int launch_bn_fold_eval(const float* bias, const float* gamma, const float* beta, const float* rm,
                        const float* rv, float eps, int bn, int H, float* scale, float* shift,
                        hipStream_t s) {
  hipLaunchKernelGGL(bn_fold_eval_kernel, dim3((H + NTHR - 1) / NTHR), dim3(NTHR), 0, s, bias, gamma, beta,
                     rm, rv, eps, bn, H, scale, shift);
  PL_CHECK_LAUNCH("bn_fold_eval");
  return PL_OK;
}

int launch_bn_eval_stats(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int H,
                         float* mean, float* rstd, float* scale, float* shift, hipStream_t s) {
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((H + NTHR - 1) / NTHR), dim3(NTHR), 0, s, gamma, beta, rm, rv, eps, H,
                     mean, rstd, scale, shift);
  PL_CHECK_LAUNCH("bn_eval_stats");
  return PL_OK;
}

int launch_fill(float* p, int64_t n, float v, hipStream_t s) {
  if (n <= 0) return PL_OK;
  int blocks = (int)((n + NTHR - 1) / NTHR);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(NTHR), 0, s, p, n, v);
  PL_CHECK_LAUNCH("fill");
  return PL_OK;
}

}  // namespace pl

// =====================================================================================
// C ABI: loss / metric / optimiser
// =====================================================================================
using namespace pl;

static int mse_blocks(int64_t n) {
  int64_t b = (n + NTHR * 4 - 1) / (NTHR * 4);
  if (b > 1024) b = 1024;
  return b < 1 ? 1 : (int)b;
}

extern "C" size_t pl_mse_scratch_bytes(int64_t n) { return (size_t)mse_blocks(n > 0 ? n : 1) * sizeof(float); }

namespace pl {
int mse_fwd_bwd_tick(const float* pred, const float* tgt, int64_t n, float grad_scale, float* dpred, float* loss_out,
                     void* scratch, uint64_t* tick, void* stream);
}

extern "C" int pl_mse_fwd_bwd(const float* pred, const float* tgt, int64_t n, float grad_scale,
                              float* dpred, float* loss_out, void* scratch, void* stream) {
  return mse_fwd_bwd_tick(pred, tgt, n, grad_scale, dpred, loss_out, scratch, nullptr, stream);
}

// the first half of mse_fwd_bwd_tick: dpred and the per-workgroup partial sums; the loss itself is summed later by a kind-2
// job of launch_reduce_rows_multi (R = mse_partials(n), inv_n = 1 / n)
int pl::mse_partial_only(const float* pred, const float* tgt, int64_t n, float grad_scale, float* dpred, void* scratch,
                         void* stream) {
  if (!pred || !tgt || !scratch) PL_FAIL(PL_EINVAL, "pl_mse_fwd_bwd: null pointer");
  if (n <= 0) PL_FAIL(PL_ESHAPE, "pl_mse_fwd_bwd: n = %lld", (long long)n);
  hipLaunchKernelGGL(mse_partial_kernel, dim3(mse_blocks(n)), dim3(NTHR), 0, (hipStream_t)stream, pred, tgt, n,
                     grad_scale * 2.0f / (float)n, dpred, (float*)scratch);
  PL_CHECK_LAUNCH("mse_partial");
  return PL_OK;
}
int pl::mse_partials(int64_t n) { return mse_blocks(n); }

// y = bias + slabs, dpred, MSE partials in one launch (mse_partial_from_slabs_kernel); false: shape not specialised
bool pl::mse_from_slabs_supported(int splits, int N) { return splits == 8 && (N == 51 || N == 34); }
int pl::mse_partial_from_slabs(const float* part, int splits, int B, int N, const float* bias, const float* tgt, float grad_scale,
                               float* y, float* dpred, void* scratch, void* stream) {
  if (!part || !tgt || !y || !dpred || !scratch) PL_FAIL(PL_EINVAL, "mse_partial_from_slabs: null pointer");
  if (!mse_from_slabs_supported(splits, N) || B < 1) PL_FAIL(PL_ESHAPE, "mse_partial_from_slabs: splits=%d N=%d", splits, N);
  const int64_t n = (int64_t)B * N;
  const float coef = grad_scale * 2.0f / (float)n;
  const dim3 grid(mse_blocks(n)), block(NTHR);
  if (N == 51) hipLaunchKernelGGL((mse_partial_from_slabs_kernel<8, 51>), grid, block, 0, (hipStream_t)stream, part, B, bias, tgt, coef, y, dpred, (float*)scratch);
  else hipLaunchKernelGGL((mse_partial_from_slabs_kernel<8, 34>), grid, block, 0, (hipStream_t)stream, part, B, bias, tgt, coef, y, dpred, (float*)scratch);
  PL_CHECK_LAUNCH("mse_partial_from_slabs");
  return PL_OK;
}

int pl::mse_fwd_bwd_tick(const float* pred, const float* tgt, int64_t n, float grad_scale, float* dpred,
                         float* loss_out, void* scratch, uint64_t* tick, void* stream) {
  if (!pred || !tgt || !loss_out || !scratch) PL_FAIL(PL_EINVAL, "pl_mse_fwd_bwd: null pointer");
  if (n <= 0) PL_FAIL(PL_ESHAPE, "pl_mse_fwd_bwd: n = %lld", (long long)n);
  hipStream_t s = (hipStream_t)stream;
  const int nb = mse_blocks(n);
  const float coef = grad_scale * 2.0f / (float)n;
  hipLaunchKernelGGL(mse_partial_kernel, dim3(nb), dim3(NTHR), 0, s, pred, tgt, n, coef, dpred,
                     (float*)scratch);
  PL_CHECK_LAUNCH("mse_partial");
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(NTHR), 0, s, (const float*)scratch, nb,
                     1.0f / (float)n, loss_out, tick);
  PL_CHECK_LAUNCH("mse_final");
  return PL_OK;
}

static int mpjpe_chunks(int64_t B) {
  int64_t c = (B + 63) / 64;
  if (c > 256) c = 256;
  return c < 1 ? 1 : (int)c;
}

extern "C" size_t pl_mpjpe_scratch_bytes(int64_t B, int64_t joints) {
  return (size_t)mpjpe_chunks(B) * (size_t)(joints > 0 ? joints : 1) * sizeof(float);
}

extern "C" int pl_mpjpe_accum(const float* pred, const float* tgt, int64_t B, int64_t joints,
                              float* metric, void* scratch, void* stream) {
  if (!pred || !tgt || !metric || !scratch) PL_FAIL(PL_EINVAL, "pl_mpjpe_accum: null pointer");
  if (B <= 0 || joints <= 0 || joints > 1024) PL_FAIL(PL_ESHAPE, "pl_mpjpe_accum: B=%lld joints=%lld", (long long)B, (long long)joints);
  hipStream_t s = (hipStream_t)stream;
  const int nc = mpjpe_chunks(B);
  const int rpc = (int)((B + nc - 1) / nc);
  hipLaunchKernelGGL(mpjpe_partial_kernel, dim3(nc), dim3(64), 0, s, pred, tgt, (int)B, (int)joints, rpc,
                     (float*)scratch);
  PL_CHECK_LAUNCH("mpjpe_partial");
  hipLaunchKernelGGL(mpjpe_final_kernel, dim3(1), dim3(64), 0, s, (const float*)scratch, nc, (int)joints,
                     metric);
  PL_CHECK_LAUNCH("mpjpe_final");
  return PL_OK;
}

static int adamw_launch(float* p, const float* g, float* m, float* v, int64_t n, const AdamWIn& in, void* stream) {
  if (!p || !g || !m || !v) PL_FAIL(PL_EINVAL, "pl_adamw_flat: null pointer");
  if (n <= 0 || in.t < (in.t_dev ? 0 : 1)) PL_FAIL(PL_ESHAPE, "pl_adamw_flat: n=%lld t=%lld", (long long)n, (long long)in.t);
  const int vec = aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v);
  int64_t work = vec ? (n >> 2) : n;
  int blocks = (int)((work + NTHR - 1) / NTHR);
  static const int cap = [] { const char* e = getenv("POSELIFT_ADAM_BLOCKS"); return e ? atoi(e) : 512; }();   // (same-box sweep, B = 64 step: 256 / 512 / 1024 / 2048 -> 0.1297 / 0.1289 / 0.1295 / 0.1304 ms; B = 4096: no difference)
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(NTHR), 0, (hipStream_t)stream, p, g, m, v, n, in, vec);
  PL_CHECK_LAUNCH("adamw");
  return PL_OK;
}

static AdamWIn adamw_in(float lr, float beta1, float beta2, float eps, float wd, float gscale, int64_t t,
                        const float* lr_dev, const uint64_t* t_dev) {
  AdamWIn in = {};
  in.lr = lr; in.beta1 = beta1; in.beta2 = beta2; in.eps = eps; in.wd = wd; in.gscale = gscale; in.t = t;
  in.lr_dev = lr_dev; in.t_dev = t_dev;
  return in;
}

extern "C" int pl_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int64_t t,
                             float grad_scale, void* stream) {
  return adamw_launch(p, g, m, v, n, adamw_in(lr, beta1, beta2, eps, weight_decay, grad_scale, t, nullptr, nullptr), stream);
}

extern "C" int pl_adamw_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev,
                                 float beta1, float beta2, float eps, float weight_decay, int64_t t_base,
                                 const uint64_t* t_dev, float grad_scale, void* stream) {
  if (!lr_dev || !t_dev) PL_FAIL(PL_EINVAL, "pl_adamw_flat_dev: null lr / t pointer");
  return adamw_launch(p, g, m, v, n, adamw_in(0.f, beta1, beta2, eps, weight_decay, grad_scale, t_base, lr_dev, t_dev), stream);
}

extern "C" int pl_adamw_flat_planes(float* p, const float* g, float* m, float* v, int64_t n, float lr, const float* lr_dev,
                                    float beta1, float beta2, float eps, float weight_decay, int64_t t,
                                    const uint64_t* t_dev, float grad_scale, const PLAdamWPlanes* planes, void* stream) {
  if ((lr_dev != nullptr) != (t_dev != nullptr)) PL_FAIL(PL_EINVAL, "pl_adamw_flat_planes: lr_dev and t_dev go together");
  AdamWIn in = adamw_in(lr, beta1, beta2, eps, weight_decay, grad_scale, t, lr_dev, t_dev);
  if (planes && planes->nseg > 0) {
    if (planes->nseg > PL_ADAMW_MAX_SEGS || (planes->kind != 1 && planes->kind != 2) || !(planes->scale > 0.f))
      PL_FAIL(PL_EINVAL, "pl_adamw_flat_planes: bad plane description");
    if (!(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v)) || (n & 3))
      PL_FAIL(PL_EINVAL, "pl_adamw_flat_planes: arenas must be 16-byte aligned, n %% 4 == 0");
    in.nseg = planes->nseg; in.kind = planes->kind; in.pscale = planes->scale;
    for (int q = 0; q < planes->nseg; ++q) {
      const PLAdamWSeg& sg = planes->seg[q];
      if (sg.offset < 0 || sg.numel <= 0 || (sg.offset & 3) || (sg.numel & 3) || sg.offset + sg.numel > n || !sg.h ||
          (planes->kind == 2 && !sg.l) || (reinterpret_cast<uintptr_t>(sg.h) & 7) || (reinterpret_cast<uintptr_t>(sg.l) & 7))
        PL_FAIL(PL_EINVAL, "pl_adamw_flat_planes: segment %d", q);
      in.seg_off[q] = sg.offset; in.seg_n[q] = sg.numel;
      in.seg_h[q] = static_cast<unsigned short*>(sg.h); in.seg_l[q] = static_cast<unsigned short*>(sg.l);
    }
  }
  return adamw_launch(p, g, m, v, n, in, stream);
}

// *counter += delta, one thread: the step counter of an optimizer whose step is replayed from a hipGraph (pl_adamw_flat_dev
// reads t = t_base + *t_dev; this ticks it behind the update)
__global__ void counter_add_kernel(uint64_t* counter, int64_t delta) { counter[0] = (uint64_t)((int64_t)counter[0] + delta); }
extern "C" int pl_counter_add(uint64_t* counter, int64_t delta, void* stream) {
  if (!counter) PL_FAIL(PL_EINVAL, "pl_counter_add: null counter");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, delta);
  PL_CHECK_LAUNCH("counter_add");
  return PL_OK;
}

extern "C" int pl_flip_pose(const float* in, float* out, int64_t B, int64_t joints, int64_t D, void* stream) {
  if (!in || !out || in == out) PL_FAIL(PL_EINVAL, "pl_flip_pose: null or aliased pointers");
  if (B <= 0 || joints != 17 || (D != 2 && D != 3)) PL_FAIL(PL_ESHAPE, "pl_flip_pose: expects (B, 17, 2|3)");
  const int64_t n = B * joints * D;
  hipLaunchKernelGGL(flip_pose_kernel, dim3((unsigned)((n + NTHR - 1) / NTHR)), dim3(NTHR), 0, (hipStream_t)stream,
                     in, out, n, (int)D);
  PL_CHECK_LAUNCH("flip_pose");
  return PL_OK;
}

extern "C" int pl_flip_tta_pack(const float* x, float* xx, int64_t B, int64_t joints, int64_t D, void* stream) {
  if (!x || !xx) PL_FAIL(PL_EINVAL, "pl_flip_tta_pack: null pointer");
  if (B <= 0 || joints != 17 || (D != 2 && D != 3)) PL_FAIL(PL_ESHAPE, "pl_flip_tta_pack: expects (B, 17, 2|3)");
  const int64_t n = B * joints * D;
  hipLaunchKernelGGL(flip_tta_pack_kernel, dim3((unsigned)((n + NTHR - 1) / NTHR)), dim3(NTHR), 0,
                     (hipStream_t)stream, x, xx, n, (int)D);
  PL_CHECK_LAUNCH("flip_tta_pack");
  return PL_OK;
}

extern "C" int pl_flip_tta_merge(const float* yy, float* y, int64_t B, int64_t joints, int64_t D, void* stream) {
  if (!yy || !y) PL_FAIL(PL_EINVAL, "pl_flip_tta_merge: null pointer");
  if (B <= 0 || joints != 17 || (D != 2 && D != 3)) PL_FAIL(PL_ESHAPE, "pl_flip_tta_merge: expects (B, 17, 2|3)");
  const int64_t n = B * joints * D;
  hipLaunchKernelGGL(flip_tta_merge_kernel, dim3((unsigned)((n + NTHR - 1) / NTHR)), dim3(NTHR), 0,
                     (hipStream_t)stream, yy, y, n, (int)D);
  PL_CHECK_LAUNCH("flip_tta_merge");
  return PL_OK;
}

extern "C" int pl_flip_pose_ex(const float* in, const float* addend, float* out, int64_t B, int64_t joints, int64_t D,
                               float x_offset, float scale, void* stream) {
  if (!in || !out || in == out || addend == out) PL_FAIL(PL_EINVAL, "pl_flip_pose_ex: null or aliased pointers");
  if (B <= 0 || joints != 17 || (D != 2 && D != 3)) PL_FAIL(PL_ESHAPE, "pl_flip_pose_ex: expects (B, 17, 2|3)");
  const int64_t n = B * joints * D;
  hipLaunchKernelGGL(flip_pose_ex_kernel, dim3((unsigned)((n + NTHR - 1) / NTHR)), dim3(NTHR), 0, (hipStream_t)stream,
                     in, addend, out, n, (int)D, x_offset, scale);
  PL_CHECK_LAUNCH("flip_pose_ex");
  return PL_OK;
}

extern "C" int pl_flip_w_nhwc(const float* in, float* out, int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
  if (!in || !out || in == out) PL_FAIL(PL_EINVAL, "pl_flip_w_nhwc: null or aliased pointers");
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || W > (1 << 30) || C > (1 << 30)) PL_FAIL(PL_ESHAPE, "pl_flip_w_nhwc: bad shape");
  const int64_t n = B * H * W * C;
  if ((n + NTHR - 1) / NTHR > 0x7fffffffLL) PL_FAIL(PL_ESHAPE, "pl_flip_w_nhwc: tensor too large");
  hipLaunchKernelGGL(flip_w_nhwc_kernel, dim3((unsigned)((n + NTHR - 1) / NTHR)), dim3(NTHR), 0, (hipStream_t)stream,
                     in, out, n, (int)W, (int)C);
  PL_CHECK_LAUNCH("flip_w_nhwc");
  return PL_OK;
}

extern "C" int pl_gather_rows2(const float* a, int64_t wa, const float* b, int64_t wb, const int64_t* idx,
                               int64_t n, int64_t table_rows, float* oa, float* ob, void* stream) {
  if (!a || !b || !idx || !oa || !ob) PL_FAIL(PL_EINVAL, "pl_gather_rows2: null pointer");
  if (n <= 0 || wa <= 0 || wb <= 0 || wa + wb > 4096 || table_rows <= 0)
    PL_FAIL(PL_ESHAPE, "pl_gather_rows2: n=%lld wa=%lld wb=%lld rows=%lld", (long long)n, (long long)wa,
            (long long)wb, (long long)table_rows);
  const int64_t total = n * (wa + wb);
  if (total > (int64_t)INT32_MAX * NTHR) PL_FAIL(PL_ESHAPE, "pl_gather_rows2: batch too large");
  hipLaunchKernelGGL(gather_rows2_kernel, dim3((unsigned)((total + NTHR - 1) / NTHR)), dim3(NTHR), 0,
                     (hipStream_t)stream, a, (int)wa, b, (int)wb, idx, n, oa, ob);
  PL_CHECK_LAUNCH("gather_rows2");
  return PL_OK;
}

extern "C" size_t pl_l1_scratch_bytes(int nterms) {
  return (size_t)(nterms > 0 ? nterms : 1) * L1_BLOCKS * sizeof(float);
}

extern "C" int pl_l1_terms_fwd_bwd(const PLL1Term* terms, int nterms, float grad_scale, float* losses,
                                   void* scratch, void* stream) {
  if (!terms || !losses || !scratch) PL_FAIL(PL_EINVAL, "pl_l1_terms_fwd_bwd: null pointer");
  if (nterms < 1 || nterms > PL_L1_MAX_TERMS) PL_FAIL(PL_EINVAL, "pl_l1_terms_fwd_bwd: nterms=%d", nterms);
  L1Terms T = {};
  for (int t = 0; t < nterms; ++t) {
    if (!terms[t].a || !terms[t].b || terms[t].n <= 0)
      PL_FAIL(PL_EINVAL, "pl_l1_terms_fwd_bwd: term %d has a null operand or n <= 0", t);
    T.t[t] = terms[t];
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(l1_partial_kernel, dim3(L1_BLOCKS, nterms), dim3(NTHR), 0, s, T, grad_scale, (float*)scratch);
  PL_CHECK_LAUNCH("l1_partial");
  hipLaunchKernelGGL(l1_final_kernel, dim3(nterms), dim3(64), 0, s, (const float*)scratch, T, losses);
  PL_CHECK_LAUNCH("l1_final");
  return PL_OK;
}

// =====================================================================================
// BatchNorm over the rows of a [rows][C] matrix as stand-alone entry points (the conv path: an NHWC feature
// map IS that matrix, BatchNorm2d = BatchNorm1d over its rows).  Same kernels as the lifter's layers.
// =====================================================================================
// rows per statistics group: 64 (the GEMM epilogue's unit) up to 16 K rows, then 256 -- a 131072-row map would
// otherwise hand the finalize kernel 2048 partials per column (38 us per layer)
// (1024 for maps so large that 256-row groups would leave > 1024 partials per column: the statistics kernel merges
//  its 64-row sub-groups itself, so longer groups only cost workgroup count)
static int bn_group_rows(int64_t rows) { return rows > 262144 ? 1024 : (rows > 16384 ? 256 : 64); }
static int bn_groups(int64_t rows) { const int gs = bn_group_rows(rows); return (int)((rows + gs - 1) / gs); }
// replicas of a narrow map: [rows][C] is worked on as [rows/R][R*C] (bn_colstats_kernel)
static int bn_replicas(int64_t rows, int64_t C) {
  int r = C <= 64 ? 4 : (C <= 128 ? 2 : 1);
  while (r > 1 && (rows % r != 0 || rows / r < 2)) r >>= 1;
  return r;
}

extern "C" size_t pl_bn_train_scratch_bytes(int64_t rows, int64_t C) {
  if (rows <= 0 || C <= 0) return 0;
  const int R = bn_replicas(rows, C);
  const int64_t rv = rows / R;                      // rows of the reshaped view
  size_t fwd = ((size_t)2 * R * bn_groups(rv) * C + 2 * (size_t)C) * sizeof(float);
  if (fwd < ((size_t)2 * 512 * C + 2 * (size_t)C) * sizeof(float)) fwd = ((size_t)2 * 512 * C + 2 * (size_t)C) * sizeof(float);
  const int rc = bwd_row_chunks((int)rv, (int)C * R);
  const size_t bwd = ((size_t)2 * R * rc * C + 3 * (size_t)C + (size_t)rc * R * C +
                      (size_t)2 * ((C * R + 255) / 256) * rc + 16) * sizeof(float);     // ... + the dz range maxima
  return fwd > bwd ? fwd : bwd;
}

// operand planes of a [n]-element tensor for the planes GEMM: mode PL_F16X3 -> [2][n] fp16 (h, l), PL_BF16 -> [n] bf16
int pl::plane_out_of(int mode, void* planes, int64_t n, float scale, const float* dyn, PlaneOut* po, const char* who) {
  *po = PlaneOut{nullptr, nullptr, scale, dyn, 0};
  if (!planes) return PL_OK;
  if (mode != PL_F16X3 && mode != PL_BF16) PL_FAIL(PL_EDTYPE, "%s: planes want PL_F16X3 or PL_BF16 (mode %d)", who, mode);
  if ((reinterpret_cast<uintptr_t>(planes) & 15) || (n & 7)) PL_FAIL(PL_EINVAL, "%s: planes misaligned", who);
  po->h = static_cast<unsigned short*>(planes);
  po->l = po->h + n;
  po->kind = mode == PL_F16X3 ? 2 : 1;
  return PL_OK;
}

extern "C" float pl_conv_act_plane_scale(void) { return kConvActPlaneScale; }

extern "C" int pl_planes_split(const float* x, int64_t n, int mode, float scale, void* planes, void* stream) {
  if (!x || !planes || n <= 0 || !(scale > 0.f)) PL_FAIL(PL_EINVAL, "pl_planes_split: bad arguments");
  PlaneOut po;
  PL_TRY(plane_out_of(mode, planes, n, scale, nullptr, &po, "pl_planes_split"));
  return launch_split_planes(x, n, po, (hipStream_t)stream);
}

// planes of a strided 4-D view: element (i0, i1, i2, i3) of the (contiguous) result is x[i0 s0 + i1 s1 + i2 s2 + i3 s3] -- x already
// points at element (0, 0, 0, 0) of the view, strides in elements and signed (a flipped axis: negative).  dims[3] % 4 == 0.
extern "C" int pl_planes_split_strided(const float* x, const int64_t* dims, const int64_t* strides, int mode, float scale,
                                       void* planes, void* stream) {
  if (!x || !dims || !strides || !planes || !(scale > 0.f)) PL_FAIL(PL_EINVAL, "pl_planes_split_strided: bad arguments");
  for (int i = 0; i < 4; ++i)
    if (dims[i] <= 0 || dims[i] > INT32_MAX) PL_FAIL(PL_ESHAPE, "pl_planes_split_strided: dims[%d] = %lld", i, (long long)dims[i]);
  if (dims[3] & 3) PL_FAIL(PL_ESHAPE, "pl_planes_split_strided: innermost extent %lld is not a multiple of 4", (long long)dims[3]);
  const int64_t n = dims[0] * dims[1] * dims[2] * dims[3];
  PlaneOut po;
  PL_TRY(plane_out_of(mode, planes, n, scale, nullptr, &po, "pl_planes_split_strided"));
  const Strided4 v = {(int)dims[1], (int)dims[2], (int)dims[3], strides[0], strides[1], strides[2], strides[3]};
  const int64_t n4 = n >> 2;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(split_planes_strided_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, v, n4, po);
  PL_CHECK_LAUNCH("split_planes_strided");
  return PL_OK;
}

extern "C" int pl_bn_train_fwd(const float* z, int64_t rows, int64_t C, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, int64_t* batches, int relu,
                               float* y, uint64_t* bits, float* mean, float* rstd, void* scratch, void* stream) {
  if (!y) PL_FAIL(PL_EINVAL, "pl_bn_train_fwd: null pointer");
  return pl_bn_train_fwd_ex(z, rows, C, gamma, beta, eps, momentum, running_mean, running_var, batches, relu, y, bits, mean,
                            rstd, scratch, nullptr, 0, nullptr, nullptr, stream);
}

// + y_planes (optional): the output also / only (y == NULL) as operand planes of the next 1x1 convolution's planes GEMM
extern "C" int pl_bn_train_fwd_ex(const float* z, int64_t rows, int64_t C, const float* gamma, const float* beta, float eps,
                                  float momentum, float* running_mean, float* running_var, int64_t* batches, int relu,
                                  float* y, uint64_t* bits, float* mean, float* rstd, void* scratch, void* y_planes,
                                  int planes_mode, const float* gemm_stat, const float* join, void* stream) {
  if (!z || !gamma || !beta || (!y && !y_planes) || !bits || !mean || !rstd || !scratch) PL_FAIL(PL_EINVAL, "pl_bn_train_fwd: null pointer");
  PlaneOut ypo;
  PL_TRY(plane_out_of(planes_mode, y_planes, rows * C, kConvActPlaneScale, nullptr, &ypo, "pl_bn_train_fwd_ex"));
  if (rows < 2 || rows > INT32_MAX || C <= 0 || (C & 3)) PL_FAIL(rows < 2 ? PL_EBATCH : PL_ESHAPE, "pl_bn_train_fwd: rows=%lld C=%lld (C %% 4 == 0, rows >= 2)", (long long)rows, (long long)C);
  hipStream_t s = (hipStream_t)stream;
  const int R = bn_replicas(rows, C);
  const int B = (int)(rows / R), H = (int)C * R, Hc = (int)C;       // the view the streaming kernels work on
  const int G = bn_groups(B), gs = bn_group_rows(B);
  float* stat = static_cast<float*>(scratch);
  float* scale = stat + (size_t)2 * R * G * Hc;
  float* shift = scale + Hc;
  if (gemm_stat) {
    // the convolution's GEMM epilogue already left (sum, M2) per 64-row group and column: [2][gemm_stat_groups(rows)][C]
    // -- no pass over z for the statistics; more than 512 groups are merged F at a time first
    const int G64 = gemm_stat_groups((int)rows);
    const float* st = gemm_stat;
    int Gf = G64, gsf = 64;
    if (G64 > 512) {
      const int F = (G64 + 511) / 512;
      Gf = (G64 + F - 1) / F; gsf = 64 * F;
      float* merged = stat;                              // (scratch >= 2 * 512 * C floats: pl_bn_train_scratch_bytes)
      scale = merged + (size_t)2 * Gf * Hc; shift = scale + Hc;
      hipLaunchKernelGGL(bn_merge_groups_kernel, dim3((Hc + NTHR - 1) / NTHR, Gf), dim3(NTHR), 0, s, gemm_stat, G64, (int)rows,
                         Hc, F, merged, Gf);
      PL_CHECK_LAUNCH("bn_merge_groups");
      st = merged;
    }
    PL_TRY(launch_bn_finalize(st, Gf, 1, (int)rows, Hc, gamma, beta, eps, momentum, running_mean, running_var, batches, mean,
                              rstd, scale, shift, s, gsf));
  } else {
    hipLaunchKernelGGL(bn_colstats_kernel, dim3((H + 255) / 256, G), dim3(NTHR), 0, s, z, B, H, gs, Hc, stat);
    PL_CHECK_LAUNCH("bn_colstats");
    PL_TRY(launch_bn_finalize(stat, G, R, B, Hc, gamma, beta, eps, momentum, running_mean, running_var, batches, mean, rstd,
                              scale, shift, s, gs));
  }
  const int strips = (H + 255) / 256;
  dim3 grid(strips, stream_rows_grid(B, strips));
  // join (optional): y = relu(bn(z) + join) in this one pass -- the Bottleneck's bn3 and residual join (needs relu != 0)
  if (join && !relu) PL_FAIL(PL_EINVAL, "pl_bn_train_fwd_ex: a join without its ReLU");
  hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(NTHR), 0, s, z, scale, shift, join, y, bits, B, H,
                     (relu ? 0 : 8) | (join ? 32 : 0), 0u, 1.0f, 0u, 0u, 0u, 0u, (const uint64_t*)nullptr, Hc, ypo,
                     (const uint64_t*)nullptr, 0u, BnFin{});
  PL_CHECK_LAUNCH("bn_apply");
  return PL_OK;
}

extern "C" int pl_bn_train_bwd(const float* dy, const uint64_t* bits, const float* z, const float* mean, const float* rstd,
                               const float* gamma, int64_t rows, int64_t C, float* dz, float* dgamma, float* dbeta,
                               void* scratch, void* stream) {
  if (!dz) PL_FAIL(PL_EINVAL, "pl_bn_train_bwd: null pointer");
  return pl_bn_train_bwd_ex(dy, bits, z, mean, rstd, gamma, rows, C, dz, dgamma, dbeta, scratch, nullptr, 0, nullptr, stream);
}

// + dz_planes (optional): dz also / only (dz == NULL) as operand planes of the 1x1 convolution's dgrad / wgrad planes GEMMs.
// PL_F16X3: the fp16 planes hold S * dz with S the power of two from the range bound of the lifter's layers
// (bn_bwd_finalize_kernel); dz_scale (device, 2 floats) receives {S, 1/S} -- the GEMMs take &dz_scale[1] as dyn_inv.
extern "C" int pl_bn_train_bwd_ex(const float* dy, const uint64_t* bits, const float* z, const float* mean, const float* rstd,
                                  const float* gamma, int64_t rows, int64_t C, float* dz, float* dgamma, float* dbeta,
                                  void* scratch, void* dz_planes, int planes_mode, float* dz_scale, void* stream) {
  if (!dy || !bits || !z || !mean || !rstd || !gamma || (!dz && !dz_planes) || !dgamma || !dbeta || !scratch)
    PL_FAIL(PL_EINVAL, "pl_bn_train_bwd: null pointer");
  if (rows < 2 || rows > INT32_MAX || C <= 0 || (C & 3)) PL_FAIL(PL_ESHAPE, "pl_bn_train_bwd: rows=%lld C=%lld", (long long)rows, (long long)C);
  const bool scaled = dz_planes && planes_mode == PL_F16X3;
  if (scaled && !dz_scale) PL_FAIL(PL_EINVAL, "pl_bn_train_bwd_ex: fp16 planes of dz need dz_scale");
  PlaneOut po;
  PL_TRY(plane_out_of(planes_mode, dz_planes, rows * C, 1.0f, scaled ? dz_scale : nullptr, &po, "pl_bn_train_bwd_ex"));
  hipStream_t s = (hipStream_t)stream;
  const int R = bn_replicas(rows, C);
  const int B = (int)(rows / R), H = (int)C * R, Hc = (int)C, RC = bwd_row_chunks(B, H);
  float* part = static_cast<float*>(scratch);
  float* coef = part + (size_t)2 * R * RC * Hc;
  float* part_db = coef + 3 * (size_t)Hc;
  float* amax = part_db + (size_t)RC * R * Hc;                      // [strips * RC][2] (pl_bn_train_scratch_bytes)
  const int n_amax = ((H + 255) / 256) * RC;
  PL_TRY(launch_bn_bwd_reduce(dy, bits, z, mean, rstd, 1.0f, B, H, part, part + (size_t)RC * Hc, s, Hc, scaled ? amax : nullptr));
  PL_TRY(launch_bn_bwd_finalize(part, RC, R, R > 1 ? -1 : 0, B, Hc, gamma, rstd, coef, dgamma, dbeta, s,
                                scaled ? amax : nullptr, n_amax, scaled ? dz_scale : nullptr));
  return launch_bn_bwd_dz(dy, bits, z, mean, rstd, coef, 1.0f, 1, B, H, dz, part_db, s, Hc, dz_planes ? &po : nullptr);
}

// Backward of the Bottleneck's bn3 + residual join (pl_bn_train_fwd_ex with `join`; Resnet.py:81-91) in three launches: dx = (g + g2)
// where the join's bitmap is set (the identity's gradient AND bn3's dy; g2 may be NULL) written by the same pass that takes
// BatchNorm-backward's column sums of it (no separate pl_mask_add_by_bits pass and no re-read of dx for the sums), the finalize,
// then dz (fp32 and / or planes as pl_bn_train_bwd_ex).  C >= 256 (no column replication: the bitmap is per (row, channel)).
extern "C" int pl_bn_join_bwd(const float* g, const float* g2, const uint64_t* bits, const float* z, const float* mean,
                              const float* rstd, const float* gamma, int64_t rows, int64_t C, float* dx, float* dz, float* dgamma,
                              float* dbeta, void* scratch, void* dz_planes, int planes_mode, float* dz_scale, void* stream) {
  if (!g || !bits || !z || !mean || !rstd || !gamma || !dx || (!dz && !dz_planes) || !dgamma || !dbeta || !scratch)
    PL_FAIL(PL_EINVAL, "pl_bn_join_bwd: null pointer");
  if (rows < 2 || rows > INT32_MAX || C <= 0 || (C & 3)) PL_FAIL(PL_ESHAPE, "pl_bn_join_bwd: rows=%lld C=%lld", (long long)rows, (long long)C);
  if (bn_replicas(rows, C) != 1) PL_FAIL(PL_ESHAPE, "pl_bn_join_bwd: C=%lld is narrower than one 256-column strip", (long long)C);
  const bool scaled = dz_planes && planes_mode == PL_F16X3;
  if (scaled && !dz_scale) PL_FAIL(PL_EINVAL, "pl_bn_join_bwd: fp16 planes of dz need dz_scale");
  PlaneOut po;
  PL_TRY(plane_out_of(planes_mode, dz_planes, rows * C, 1.0f, scaled ? dz_scale : nullptr, &po, "pl_bn_join_bwd"));
  hipStream_t s = (hipStream_t)stream;
  const int B = (int)rows, H = (int)C, RC = bwd_row_chunks(B, H);
  float* part = static_cast<float*>(scratch);
  float* coef = part + (size_t)2 * RC * H;
  float* part_db = coef + 3 * (size_t)H;
  float* amax = part_db + (size_t)RC * H;
  const int n_amax = ((H + 255) / 256) * RC;
  PL_TRY(launch_bn_bwd_reduce(g, bits, z, mean, rstd, 1.0f, B, H, part, part + (size_t)RC * H, s, H, scaled ? amax : nullptr, 0, g2, dx));
  PL_TRY(launch_bn_bwd_finalize(part, RC, 1, 0, B, H, gamma, rstd, coef, dgamma, dbeta, s, scaled ? amax : nullptr, n_amax,
                                scaled ? dz_scale : nullptr));
  return launch_bn_bwd_dz(dx, bits, z, mean, rstd, coef, 1.0f, 1, B, H, dz, part_db, s, H, dz_planes ? &po : nullptr);
}

extern "C" int pl_add_relu_fwd(const float* a, const float* b, int64_t rows, int64_t C, float* out, uint64_t* bits,
                               void* stream) {
  return pl_add_relu_fwd_ex(a, b, rows, C, out, bits, nullptr, 0, stream);
}

// + out_planes (optional): the block output also as operand planes (the next block's 1x1 convolutions read those)
extern "C" int pl_add_relu_fwd_ex(const float* a, const float* b, int64_t rows, int64_t C, float* out, uint64_t* bits,
                                  void* out_planes, int planes_mode, void* stream) {
  if (!a || !b || !out || !bits) PL_FAIL(PL_EINVAL, "pl_add_relu_fwd: null pointer");
  if (rows <= 0 || rows > INT32_MAX || C <= 0 || (C & 3)) PL_FAIL(PL_ESHAPE, "pl_add_relu_fwd: rows=%lld C=%lld", (long long)rows, (long long)C);
  PlaneOut po;
  PL_TRY(plane_out_of(planes_mode, out_planes, rows * C, kConvActPlaneScale, nullptr, &po, "pl_add_relu_fwd_ex"));
  const int strips = ((int)C + 255) / 256;
  dim3 grid(strips, stream_rows_grid((int)rows, strips));
  hipLaunchKernelGGL(add_relu_kernel, grid, dim3(NTHR), 0, (hipStream_t)stream, a, b, out, bits, (int)rows, (int)C, po);
  PL_CHECK_LAUNCH("add_relu");
  return PL_OK;
}

extern "C" int pl_mask_by_bits(const float* g, const uint64_t* bits, int64_t rows, int64_t C, float* dx, void* stream) {
  return pl_mask_add_by_bits(g, nullptr, bits, rows, C, dx, stream);
}

// dx = (g + g2) where the bitmap is set, else 0: the residual join's backward with the sum of the two gradients that
// reach the block output (through the next join and through the next block's convolutions) folded in (g2 may be NULL)
extern "C" int pl_mask_add_by_bits(const float* g, const float* g2, const uint64_t* bits, int64_t rows, int64_t C, float* dx,
                                   void* stream) {
  if (!g || !bits || !dx) PL_FAIL(PL_EINVAL, "pl_mask_by_bits: null pointer");
  if (rows <= 0 || rows > INT32_MAX || C <= 0 || (C & 3)) PL_FAIL(PL_ESHAPE, "pl_mask_by_bits: rows=%lld C=%lld", (long long)rows, (long long)C);
  const int strips = ((int)C + 255) / 256;
  dim3 grid(strips, stream_rows_grid((int)rows, strips));
  hipLaunchKernelGGL(mask_by_bits_kernel, grid, dim3(NTHR), 0, (hipStream_t)stream, g, bits, dx, (int)rows, (int)C, g2);
  PL_CHECK_LAUNCH("mask_by_bits");
  return PL_OK;
}
