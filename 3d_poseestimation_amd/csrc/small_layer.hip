// One hidden layer of the lifter per launch, for SMALL batches (B <= 64 rows: the reference's own batch_size = 64,
// phase1_lifting/train_1.py:194).
//
// At this size every kernel of the step is launch latency (4.5-9 us each for 256 KB of data) and a hidden layer is four of
// them forward (split-K GEMM, its reduce, BatchNorm statistics + apply) and four backward.  BatchNorm is what forces the
// boundaries -- its statistics need every row of a column -- so here a workgroup OWNS 16 columns for ALL rows:
//   forward  (small_fwd_kernel):  z = a W^T + b on its columns (K split over the workgroup's eight waves, exact-fp32 MFMA
//             v_mfma_f32_16x16x4_f32 fed from registers, partials through LDS in wave order), then -- the columns are
//             complete -- batch statistics, running-statistics update, scale/shift, ReLU, dropout, the residual add and the
//             ReLU & keep bitmap: Linear + BatchNorm1d + ReLU + Dropout (+ skip) of baselineModel.py:33-37 / 39-45 / 79-95.
//   backward (small_bwd_kernel):  g = dz W (+ skip gradient) on its columns = the incoming gradient of the layer BELOW, whose
//             BatchNorm backward (sum dy, sum dy zhat, the coefficients, dz, dgamma, dbeta, the bias gradient) follows in the
//             same workgroup: the autograd of the same modules.
// 64 workgroups at H = 1024; the MFMA work of one (2 MFLOP at 256 FLOP/clk) is 3.4 us, the operands come from L2.
//
// Bitmap of the layers produced here ("tile format", private to the small path: written by small_fwd_kernel, read by
// small_bwd_kernel and bn_small_bwd_kernel<.., true>): element (r, c) is bit (r & 15) * 4 + ((c >> 2) & 3) of word
// (c >> 4) * 16 + (r >> 4) * 4 + (c & 3) -- what a ballot over the epilogue's lanes yields.
#include <stdlib.h>

#include "bn_pieces.h"
#include "philox.h"
#include "pl_internal.h"

namespace pl {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NTHR = 512, NWAVE = 8, COLS = 16, ROWS = 64;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float comp(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// the workgroup's 16 columns: column blocks in XCD-contiguous runs (workgroup b runs on XCD b % 8), so that the two blocks
// sharing a 128-byte line of W (NN) fetch it into one L2
__device__ __forceinline__ int col_block(int b, int nb) {
  if (nb & 7) return b;
  return (b & 7) * (nb >> 3) + (b >> 3);
}

// part[wave][64 x 16] = A[:, wave's k range] * B[that range, c0 .. c0+15]; wave's range = STEPS * 32 k.
//   B_KS = false: B [N][K] k-contiguous (forward, W rows);  true: B [K][N] (backward: dz W, 64-byte segments of W's rows)
// The MFMA wants its 16 rows on adjacent lanes; memory has k on adjacent addresses.  (First version: every lane fetched the
// 32 bytes of its own row straight into its fragment -- 64 separate requests per instruction: 8.5 us of a 15 us kernel for
// 320 KB, measured with the MFMAs removed.)  So the global loads are row-contiguous -- eight lanes per 128-byte row piece, all
// of the wave's 40 in flight at once -- and each 32-k step passes through a wave-private LDS image [row][36 floats] (the pad
// makes both the b128 writes and the fragment reads conflict-free); no barrier: one wave's DS operations execute in order.
// Rows >= M read row M - 1 (valid memory); their results are never used.
constexpr int LDR = 36;                                   // floats per row of the step image
constexpr int STAGE = (ROWS + COLS) * LDR;                // per wave: 64 rows of A, 16 of B
template <int STEPS, bool B_KS, int ABL = 0>
__device__ __forceinline__ void contract(const float* __restrict__ A, int lda, int M, const float* __restrict__ Bm, int ldb,
                                         int c0, float* __restrict__ part, float* __restrict__ stage) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int kb = wave * STEPS * 32;
  float* img = stage + wave * STAGE;
  float4 ga[STEPS][8], gb[STEPS][2];
  {
    const int rr = lane >> 3, q4 = 4 * (lane & 7);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
      for (int j = 0; j < 8; ++j) ga[s][j] = ld4(A + (size_t)min(8 * j + rr, M - 1) * lda + kb + 32 * s + q4);
      if (!B_KS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) gb[s][j] = ld4(Bm + (size_t)(c0 + 8 * j + rr) * ldb + kb + 32 * s + q4);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) gb[s][j] = ld4(Bm + (size_t)(kb + 32 * s + 16 * j + (lane >> 2)) * ldb + c0 + 4 * (lane & 3));
      }
      __builtin_amdgcn_sched_barrier(0);                  // (all 40 loads leave before anything below, in step order: one round trip)
    }
  }
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    {
      const int rr = lane >> 3, q4 = 4 * (lane & 7);
#pragma unroll
      for (int j = 0; j < 8; ++j) st4(img + (8 * j + rr) * LDR + q4, ga[s][j]);
      if (!B_KS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) st4(img + (ROWS + 8 * j + rr) * LDR + q4, gb[s][j]);
      } else {
        // a lane holds four columns of one k: transposed into the [column][k] image
        const int kr = lane >> 2, cq = 4 * (lane & 3);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          img[(ROWS + cq + 0) * LDR + 16 * j + kr] = gb[s][j].x;
          img[(ROWS + cq + 1) * LDR + 16 * j + kr] = gb[s][j].y;
          img[(ROWS + cq + 2) * LDR + 16 * j + kr] = gb[s][j].z;
          img[(ROWS + cq + 3) * LDR + 16 * j + kr] = gb[s][j].w;
        }
      }
    }
    float4 a[4][2], b[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a[t][0] = ld4(img + (16 * t + i) * LDR + 8 * kq);
      a[t][1] = ld4(img + (16 * t + i) * LDR + 8 * kq + 4);
    }
    b[0] = ld4(img + (ROWS + i) * LDR + 8 * kq);
    b[1] = ld4(img + (ROWS + i) * LDR + 8 * kq + 4);
    if (ABL == 2) {        // timing only: everything but the MFMAs
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t][0] += a[t][0].x + a[t][1].x + b[0].x; acc[t][1] += a[t][0].y + a[t][1].y + b[1].y;
        acc[t][2] += a[t][0].z + a[t][1].z; acc[t][3] += a[t][0].w + a[t][1].w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(a[t][e >> 2], e & 3), comp(b[e >> 2], e & 3), acc[t], 0, 0, 0);
    }
  }
  float* mine = part + wave * (ROWS * COLS);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) mine[(16 * t + 4 * kq + v) * COLS + i] = acc[t][v];
}

// the epilogue's element of thread tid < 256: row tid >> 2 (= 16 wave + (lane >> 2)), columns c0 + 4 (tid & 3) .. + 3
__device__ __forceinline__ float4 gather_part(const float* __restrict__ part) {
  const int tid = threadIdx.x;
  float4 v = ld4(part + tid * 4);
#pragma unroll
  for (int w = 1; w < NWAVE; ++w) {
    const float4 u = ld4(part + w * (ROWS * COLS) + tid * 4);
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  return v;
}

// column sums over the 64 rows of NV float4 values per thread (threads >= 256 pass zeros and get garbage): the wave's 16 rows
// by butterfly over lanes l ^ 4, 8, 16, 32, the four epilogue waves through LDS in wave order
template <int NV>
__device__ __forceinline__ void colsum(float4 (&v)[NV], float4 (*sm)[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cq = lane & 3;
#pragma unroll
  for (int n = 0; n < NV; ++n)
#pragma unroll
    for (int o = 4; o <= 32; o <<= 1) {
      v[n].x += __shfl_xor(v[n].x, o); v[n].y += __shfl_xor(v[n].y, o);
      v[n].z += __shfl_xor(v[n].z, o); v[n].w += __shfl_xor(v[n].w, o);
    }
  if (wave < 4 && lane < 4) {
#pragma unroll
    for (int n = 0; n < NV; ++n) sm[n * 4 + wave][lane] = v[n];
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    float4 t = sm[n * 4][cq];
#pragma unroll
    for (int w = 1; w < 4; ++w) { const float4 x = sm[n * 4 + w][cq]; t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w; }
    v[n] = t;
  }
  __syncthreads();
}

struct FwdArgs {
  const float *a, *W, *bias, *gamma, *beta, *resid;
  float *rm, *rv, *mean, *rstd, *z, *act;
  int64_t* nbt;
  uint64_t* bits;
  const uint64_t *inject, *step_dev;
  float eps, momentum, kscale;
  int B, H, K, mode;
  uint32_t thr, k0, k1, c3, layer, seed_hi;
};

template <int STEPS, int ABL = 0>
__global__ __launch_bounds__(NTHR) void small_fwd_kernel(FwdArgs p) {
  __shared__ float part[NWAVE * ROWS * COLS];
  __shared__ float stage[NWAVE * STAGE];
  __shared__ float4 sm[4][4];
  const int blk = col_block(blockIdx.x, gridDim.x);
  const int c0 = blk * COLS;
  if (ABL != 1) contract<STEPS, false, ABL>(p.a, p.K, p.B, p.W, p.K, c0, part, stage);
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = tid >> 2, c = c0 + 4 * (tid & 3);
  const bool live = tid < 256 && r < p.B;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 z = zero, ga = zero, be = zero, rv = zero;
  if (tid < 256) {
    z = gather_part(part);
    const float4 b = ld4(p.bias + c);
    z.x += b.x; z.y += b.y; z.z += b.z; z.w += b.w;
    ga = ld4(p.gamma + c); be = ld4(p.beta + c);
    if (p.resid && live) rv = ld4(p.resid + (size_t)r * p.H + c);
  }
  if (!live) z = zero;
  if (live) st4(p.z + (size_t)r * p.H + c, z);
  const float Bt = (float)p.B;
  float4 s[1] = {z};
  colsum<1>(s, sm);
  const float4 mean = make_float4(s[0].x / Bt, s[0].y / Bt, s[0].z / Bt, s[0].w / Bt);
  float4 q[1] = {zero};
  if (live) {
    const float dx = z.x - mean.x, dy = z.y - mean.y, dz = z.z - mean.z, dw = z.w - mean.w;
    q[0] = make_float4(dx * dx, dy * dy, dz * dz, dw * dw);
  }
  colsum<1>(q, sm);
  if (tid >= 256) return;
  const float var[4] = {q[0].x / Bt, q[0].y / Bt, q[0].z / Bt, q[0].w / Bt};
  const float mu[4] = {mean.x, mean.y, mean.z, mean.w};
  const float g4[4] = {ga.x, ga.y, ga.z, ga.w}, b4[4] = {be.x, be.y, be.z, be.w};
  float rs[4], sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rs[j] = 1.0f / sqrtf(var[j] + p.eps);
    sc[j] = g4[j] * rs[j];
    sh[j] = bn_shift_of(b4[j], mu[j], sc[j]);
  }
  if (tid < 4) {
    st4(p.mean + c, mean);
    st4(p.rstd + c, make_float4(rs[0], rs[1], rs[2], rs[3]));
    if (p.rm) {
      float4 m = ld4(p.rm + c), v = ld4(p.rv + c);
      bn_running_update(m.x, v.x, mu[0], var[0], Bt, p.momentum);
      bn_running_update(m.y, v.y, mu[1], var[1], Bt, p.momentum);
      bn_running_update(m.z, v.z, mu[2], var[2], Bt, p.momentum);
      bn_running_update(m.w, v.w, mu[3], var[3], Bt, p.momentum);
      st4(p.rm + c, m); st4(p.rv + c, v);
    }
  }
  if (p.nbt && blockIdx.x == 0 && tid == 0) p.nbt[0] += 1;
  // ReLU, dropout, bitmap, residual (element order of bn_apply_row, elementwise.hip)
  uint32_t c3 = p.c3, k1 = p.k1;
  if (p.step_dev) {
    const uint64_t step = (((uint64_t)k1 << 32) | c3) + p.step_dev[0];
    c3 = (uint32_t)step;
    k1 = p.seed_hi ^ (uint32_t)(step >> 32);
  }
  const int mode = p.mode & 7;
  const bool norelu = (p.mode & 8) != 0;
  const float zv[4] = {z.x, z.y, z.z, z.w};
  float y[4];
  bool keep[4] = {true, true, true, true};
#pragma unroll
  for (int j = 0; j < 4; ++j) y[j] = fmaf(zv[j], sc[j], sh[j]);
  if (live) {
    if (mode == 1) {
      const uint64_t g = ((uint64_t)r * (uint64_t)p.H + (uint64_t)c) >> 2;
      const Philox4 u = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), p.layer, c3, p.k0, k1);
#pragma unroll
      for (int j = 0; j < 4; ++j) keep[j] = u.v[j] >= p.thr;
    } else if (mode == 2) {
      const int wpr = (((p.H + 255) >> 8) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) keep[j] = (p.inject[(size_t)r * wpr + (c >> 8) * 4 + j] >> ((c >> 2) & 63)) & 1ull;
    } else if (mode == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) keep[j] = false;
    }
  }
  float o[4];
  uint64_t word = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool on = live && keep[j] && (norelu || y[j] > 0.f);
    o[j] = on ? y[j] * p.kscale : 0.f;
    const uint64_t bal = __ballot(on);
    if (lane == j) word = bal;
  }
  if (lane < 4) p.bits[(size_t)blk * 16 + wave * 4 + lane] = word;
  if (live) st4(p.act + (size_t)r * p.H + c, make_float4(o[0] + rv.x, o[1] + rv.y, o[2] + rv.z, o[3] + rv.w));
}

// The weight gradient of the same layer rides in the backward launch as extra workgroups (it needs dz and the layer's input,
// both complete when the launch starts): dW [N][Kin] = dz^T a over the B <= 64 rows, one 128 x 64 tile per workgroup; wave w
// owns rows 32 (w & 3) .. + 31 and columns 32 (w >> 2) .. + 31 of the tile: 2 x 2 MFMA tiles, sixteen 4-row steps.
__device__ __forceinline__ void dw_tile(const float* __restrict__ dz, const float* __restrict__ a, float* __restrict__ dW, int B,
                                        int N, int Kin, int tile) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int tpr = Kin >> 6;                              // tiles per row of tiles
  const int n0 = (tile / tpr) * 128 + 32 * (wave & 3), k0 = (tile % tpr) * 64 + 32 * (wave >> 2);
  float av[16][2], bv[16][2];
#pragma unroll
  for (int st = 0; st < 16; ++st) {
    const int r = 4 * st + kq;
    const int rc = min(r, B - 1);
    const float* pa = dz + (size_t)rc * N + n0 + i;
    const float* pb = a + (size_t)rc * Kin + k0 + i;
    av[st][0] = pa[0]; av[st][1] = pa[16];
    bv[st][0] = pb[0]; bv[st][1] = pb[16];
    if (r >= B) av[st][0] = av[st][1] = 0.f;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int st = 0; st < 16; ++st)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st][x], bv[st][y], acc[x][y], 0, 0, 0);
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 4; ++v) dW[(size_t)(n0 + 16 * x + 4 * kq + v) * Kin + k0 + 16 * y + i] = acc[x][y][v];
}

struct BwdArgs {
  const float *a_in;                      // the layer's input [B][H] and
  float* dW;                              //   its weight gradient [K][H] = dz^T a_in (extra workgroups), or NULL
  int nblk_dx;                            // workgroups of the dX part (H / 16)
  const float *dz, *W, *addend;           // g = dz W (+ addend) on the workgroup's columns
  float* gout;                            // g is kept here when something later reads it (the skip gradient), or NULL
  // the layer below
  const float *z, *mean, *rstd, *gamma;
  const uint64_t* bits;
  float *dz_lo, *dgamma, *dbeta, *dbias;
  float kscale;
  int B, H, K, rowbits;                   // rowbits: the lower layer's bitmap is in the row format of bn_apply_row
};

template <int STEPS>
__global__ __launch_bounds__(NTHR) void small_bwd_kernel(BwdArgs p) {
  __shared__ float part[NWAVE * ROWS * COLS];
  __shared__ float stage[NWAVE * STAGE];
  __shared__ float4 sm[12][4];
  if ((int)blockIdx.x >= p.nblk_dx) {       // (workgroup-uniform)
    dw_tile(p.dz, p.a_in, p.dW, p.B, p.K, p.H, blockIdx.x - p.nblk_dx);
    return;
  }
  const int blk = col_block(blockIdx.x, p.nblk_dx);
  const int c0 = blk * COLS;
  contract<STEPS, true>(p.dz, p.K, p.B, p.W, p.H, c0, part, stage);
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = tid >> 2, c = c0 + 4 * (tid & 3);
  const bool live = tid < 256 && r < p.B;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 g = zero, zl = zero, mu = zero, rs = zero, ga = zero;
  uint64_t bw[4] = {0, 0, 0, 0};
  int bit = lane;
  if (tid < 256) {
    mu = ld4(p.mean + c); rs = ld4(p.rstd + c); ga = ld4(p.gamma + c);
    if (live) {
      zl = ld4(p.z + (size_t)r * p.H + c);
      if (p.rowbits) {
        const uint64_t* q = p.bits + (size_t)r * (((p.H + 255) >> 8) * 4) + (c >> 8) * 4;
        const ulonglong2 b01 = *reinterpret_cast<const ulonglong2*>(q), b23 = *reinterpret_cast<const ulonglong2*>(q + 2);
        bw[0] = b01.x; bw[1] = b01.y; bw[2] = b23.x; bw[3] = b23.y;
        bit = (c >> 2) & 63;
      } else {
        const uint64_t* q = p.bits + (size_t)blk * 16 + wave * 4;
        const ulonglong2 b01 = *reinterpret_cast<const ulonglong2*>(q), b23 = *reinterpret_cast<const ulonglong2*>(q + 2);
        bw[0] = b01.x; bw[1] = b01.y; bw[2] = b23.x; bw[3] = b23.y;
      }
    }
    g = gather_part(part);
    if (p.addend && live) { const float4 t = ld4(p.addend + (size_t)r * p.H + c); g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w; }
    if (p.gout && live) st4(p.gout + (size_t)r * p.H + c, g);
  }
  float4 dv = zero, zh = zero;
  if (live) {
    dv.x = ((bw[0] >> bit) & 1ull) ? g.x * p.kscale : 0.f;
    dv.y = ((bw[1] >> bit) & 1ull) ? g.y * p.kscale : 0.f;
    dv.z = ((bw[2] >> bit) & 1ull) ? g.z * p.kscale : 0.f;
    dv.w = ((bw[3] >> bit) & 1ull) ? g.w * p.kscale : 0.f;
    zh = make_float4((zl.x - mu.x) * rs.x, (zl.y - mu.y) * rs.y, (zl.z - mu.z) * rs.z, (zl.w - mu.w) * rs.w);
  }
  float4 s[3];
  s[0] = dv;
  s[1] = make_float4(dv.x * zh.x, dv.y * zh.y, dv.z * zh.z, dv.w * zh.w);
  s[2] = zh;
  colsum<3>(s, sm);
  if (tid >= 256) return;
  const float Bt = (float)p.B;
  const float4 s1 = s[0], s2 = s[1], sz = s[2];
  const float4 k0 = make_float4(ga.x * rs.x, ga.y * rs.y, ga.z * rs.z, ga.w * rs.w);
  const float4 k1 = make_float4(s1.x / Bt, s1.y / Bt, s1.z / Bt, s1.w / Bt);
  const float4 k2 = make_float4(s2.x / Bt, s2.y / Bt, s2.z / Bt, s2.w / Bt);
  if (tid < 4) {
    st4(p.dgamma + c, s2); st4(p.dbeta + c, s1);
    // (the bias in front of BatchNorm: true gradient 0; formed from the three sums as in bn_small_bwd_kernel)
    float4 db;
    db.x = k0.x * ((s1.x - Bt * k1.x) - k2.x * sz.x); db.y = k0.y * ((s1.y - Bt * k1.y) - k2.y * sz.y);
    db.z = k0.z * ((s1.z - Bt * k1.z) - k2.z * sz.z); db.w = k0.w * ((s1.w - Bt * k1.w) - k2.w * sz.w);
    st4(p.dbias + c, db);
  }
  if (!live) return;
  float4 d;
  d.x = k0.x * (dv.x - k1.x - zh.x * k2.x); d.y = k0.y * (dv.y - k1.y - zh.y * k2.y);
  d.z = k0.z * (dv.z - k1.z - zh.z * k2.z); d.w = k0.w * (dv.w - k1.w - zh.w * k2.w);
  st4(p.dz_lo + (size_t)r * p.H + c, d);
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

// shapes the two kernels take: B rows in one tile, 16-column blocks, K split over eight waves in 32-k steps
bool small_layer_ok(int B, int H, int K) {
  static const bool off = [] { const char* e = getenv("POSELIFT_SMALL_LAYER"); return e && e[0] == '0'; }();   // =0: same-box A/B
  if (off || B < 2 || B > ROWS || (H & 15) || H < 16) return false;
  const int steps = K / (NWAVE * 32);
  return K % (NWAVE * 32) == 0 && (steps == 1 || steps == 2 || steps == 4);
}

int launch_small_layer_fwd(const float* a, const float* W, const float* bias, const float* gamma, const float* beta, float eps,
                           float momentum, float* rm, float* rv, int64_t* nbt, float* mean, float* rstd, const float* resid,
                           float* z, float* act, uint64_t* bits, int B, int H, int K, float pdrop, uint64_t seed, uint64_t step,
                           int layer, const uint64_t* inject_keep, hipStream_t s, const uint64_t* step_dev) {
  if (!small_layer_ok(B, H, K)) PL_FAIL(PL_ESHAPE, "small_layer_fwd: B=%d H=%d K=%d", B, H, K);
  if (!a || !W || !bias || !gamma || !beta || !mean || !rstd || !z || !act || !bits || (rm != nullptr) != (rv != nullptr))
    PL_FAIL(PL_EINVAL, "small_layer_fwd: bad arguments");
  if (!al16(a) || !al16(W) || !al16(bias) || !al16(gamma) || !al16(beta) || !al16(mean) || !al16(rstd) || !al16(z) || !al16(act) ||
      !al16(resid) || !al16(rm) || !al16(rv) || !al16(bits))
    PL_FAIL(PL_EINVAL, "small_layer_fwd: 16-byte alignment");
  FwdArgs p = {};
  p.a = a; p.W = W; p.bias = bias; p.gamma = gamma; p.beta = beta; p.resid = resid;
  p.rm = rm; p.rv = rv; p.mean = mean; p.rstd = rstd; p.z = z; p.act = act; p.nbt = nbt; p.bits = bits;
  p.inject = inject_keep; p.step_dev = step_dev; p.eps = eps; p.momentum = momentum;
  p.B = B; p.H = H; p.K = K;
  p.mode = 0; p.kscale = 1.f;
  if (pdrop >= 1.f) p.mode = 3;
  else if (pdrop > 0.f) { p.mode = inject_keep ? 2 : 1; p.kscale = 1.0f / (1.0f - pdrop); }
  p.thr = dropout_threshold(pdrop);
  p.k0 = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
  p.k1 = step_dev ? (uint32_t)(step >> 32) : p.seed_hi ^ (uint32_t)(step >> 32);
  p.c3 = (uint32_t)step; p.layer = (uint32_t)layer;
  const dim3 grid(H / COLS), block(NTHR);
  void* prof = prof_begin_flops(2.0 * B * H * K, s);
  switch (K / (NWAVE * 32)) {
    case 1: hipLaunchKernelGGL(small_fwd_kernel<1>, grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL(small_fwd_kernel<2>, grid, block, 0, s, p); break;
    default: {
      static const int abl = [] { const char* e = getenv("POSELIFT_SL_ABL"); return e ? atoi(e) : 0; }();   // timing only
      if (abl == 1) hipLaunchKernelGGL((small_fwd_kernel<4, 1>), grid, block, 0, s, p);
      else if (abl == 2) hipLaunchKernelGGL((small_fwd_kernel<4, 2>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((small_fwd_kernel<4, 0>), grid, block, 0, s, p);
      break;
    }
  }
  prof_end(prof, s);
  PL_CHECK_LAUNCH("small_layer_fwd");
  return PL_OK;
}

int launch_small_layer_bwd(const float* dz, const float* W, const float* addend, float* gout, int B, int H, int K,
                           const float* z_lo, const uint64_t* bits_lo, bool rowbits, const float* mean_lo, const float* rstd_lo,
                           const float* gamma_lo, float kscale, float* dz_lo, float* dgamma, float* dbeta, float* dbias,
                           hipStream_t s, const float* a_in, float* dW) {
  if (!small_layer_ok(B, H, K)) PL_FAIL(PL_ESHAPE, "small_layer_bwd: B=%d H=%d K=%d", B, H, K);
  if (!dz || !W || !z_lo || !bits_lo || !mean_lo || !rstd_lo || !gamma_lo || !dz_lo || !dgamma || !dbeta || !dbias || dz_lo == dz)
    PL_FAIL(PL_EINVAL, "small_layer_bwd: bad arguments");
  if (!al16(dz) || !al16(W) || !al16(addend) || !al16(gout) || !al16(z_lo) || !al16(bits_lo) || !al16(mean_lo) || !al16(rstd_lo) ||
      !al16(gamma_lo) || !al16(dz_lo) || !al16(dgamma) || !al16(dbeta) || !al16(dbias))
    PL_FAIL(PL_EINVAL, "small_layer_bwd: 16-byte alignment");
  BwdArgs p = {};
  p.dz = dz; p.W = W; p.addend = addend; p.gout = gout; p.z = z_lo; p.mean = mean_lo; p.rstd = rstd_lo; p.gamma = gamma_lo;
  p.bits = bits_lo; p.dz_lo = dz_lo; p.dgamma = dgamma; p.dbeta = dbeta; p.dbias = dbias; p.kscale = kscale;
  p.B = B; p.H = H; p.K = K; p.rowbits = rowbits ? 1 : 0;
  p.nblk_dx = H / COLS;
  int extra = 0;
  if (dW) {
    if (!a_in || !al16(a_in) || !al16(dW) || (K & 127) || (H & 63)) PL_FAIL(PL_EINVAL, "small_layer_bwd: weight-gradient part (K=%d H=%d)", K, H);
    p.a_in = a_in; p.dW = dW;
    extra = (K / 128) * (H / 64);
  }
  const dim3 grid(H / COLS + extra), block(NTHR);
  void* prof = prof_begin_flops(2.0 * B * H * K * (dW ? 2 : 1), s);
  switch (K / (NWAVE * 32)) {
    case 1: hipLaunchKernelGGL(small_bwd_kernel<1>, grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL(small_bwd_kernel<2>, grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL(small_bwd_kernel<4>, grid, block, 0, s, p); break;
  }
  prof_end(prof, s);
  PL_CHECK_LAUNCH("small_layer_bwd");
  return PL_OK;
}

}  // namespace pl
