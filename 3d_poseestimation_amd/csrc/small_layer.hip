// One hidden layer of the lifter per launch, for SMALL batches (B <= 64 rows: the reference's own batch_size = 64,
// phase1_lifting/train_1.py:194) -- and, in evaluation, for every batch off the tile grid (M <= 512).
//
// At this size every kernel of the step is launch latency (4.5-9 us each for 256 KB of data) and a hidden layer was four of
// them forward (split-K GEMM, its reduce, BatchNorm statistics + apply) and four backward.  BatchNorm is what forces the
// boundaries -- its statistics need every row of a column -- so here a workgroup OWNS 16 columns for ALL rows:
//   small_first_fwd_kernel   the first layer (34 / 51 inputs: contraction on the vector unit), then the forward tail
//   small_fwd_kernel         z = a W^T + b on its columns -- K split over the workgroup's eight waves, row-contiguous loads
//                            through a wave-private LDS image, exact-fp32 v_mfma_f32_16x16x4_f32 or (PL_F16X3) three
//                            v_mfma_f32_16x16x32_f16 on fp16 planes, wave partials through LDS in wave order -- then, the
//                            columns being complete, the forward tail: batch statistics, running statistics, scale / shift,
//                            ReLU, dropout, the residual add, the ReLU & keep bitmap (Linear + BatchNorm1d + ReLU + Dropout
//                            (+ skip), baselineModel.py:33-37 / 39-45 / 79-95); evaluation: the fold on the running
//                            statistics instead, grid also over 64-row blocks; the last layer also leaves the slabs of the
//                            output Linear
//   small_mse_kernel         y = bias + slabs, MSE forward + backward (or, evaluation, y alone)
//   small_top_bwd_kernel     g = dy W2, BatchNorm backward of the top hidden layer, dW2, db2, loss, step counter
//   small_bwd_kernel         three roles by workgroup: g = dz W (+ skip gradient) on its columns and the BatchNorm backward
//                            of the layer BELOW (+ the first layer's weight gradient); the layer's own weight gradient
//                            dz^T a, one 128 x 64 tile each; a slice of the AdamW step
// (the autograd of the same modules, train_1.py:95-96).  64 column workgroups at H = 1024: column ownership caps the
// parallelism, and every one of them reads all of the activations -- what a launch waits for is its 320 KB of operands through
// one CU's vector memory unit and LDS, then (exact fp32) 3.4 us of MFMA issue.  DESIGN.md 3.7 has the measurements.
//
// Bitmap of the layers produced here ("tile format", private to the small path: written by the forward tail, read by the
// backward tail and by bn_small_bwd_kernel<.., true>): element (r, c) is bit (r & 15) * 4 + ((c >> 2) & 3) of word
// (c >> 4) * 16 + (r >> 4) * 4 + (c & 3) -- what a ballot over the tail's lanes yields.
#include <stdlib.h>

#include <algorithm>

#include "adamw.h"
#include "bn_pieces.h"
#include "philox.h"
#include "pl_internal.h"
#include "plane_store.h"

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
constexpr int LDH = 40;                                   // halves per row of one PLANE's step image (f16x3 form below)
// per wave: 64 rows of A, 16 of B -- fp32 rows of LDR floats, or two fp16 planes of LDH halves each (the larger of the two)
constexpr int STAGE = (ROWS + COLS) * (LDR > LDH ? LDR : LDH);
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

// The same contraction on fp16 operand planes (PL_F16X3 descriptors, forward): A arrives as the two planes the producing
// launch's tail wrote (h = fp16(x), l = fp16((x - h) 2048): plane_store.h), B = W is split on the fly (16 rows per workgroup:
// 32 elements per thread), and a 32-k step is three v_mfma_f32_16x16x32_f16 per row tile on two accumulators -- main = Ah Bh,
// low = Ah Bl + Al Bh, result (main + low / 2048) / (S_a S_w) -- 192 MFMA cycles per step instead of the 1,024 of the exact
// fp32 form: the 3.8 us of MFMA issue that a forward launch was waiting for becomes 0.7.  The fp32-grade arithmetic of the
// bench's large batches (gemm_planes.h), not exact fp32: only descriptors that ask for it take it.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4h __attribute__((ext_vector_type(4)));
template <int STEPS>
__device__ __forceinline__ void contract_f16(const unsigned short* __restrict__ Ah, size_t a_plane, int lda, int M,
                                             const float* __restrict__ Bm, int ldb, int c0, float w_scale, float out_scale,
                                             float* __restrict__ part, float* __restrict__ stage) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int kb = wave * STEPS * 32;
  _Float16* img = reinterpret_cast<_Float16*>(stage + wave * STAGE);
  _Float16* iAh = img;
  _Float16* iAl = img + ROWS * LDH;
  _Float16* iBh = img + 2 * ROWS * LDH;
  _Float16* iBl = iBh + COLS * LDH;
  uint4 gah[STEPS][4], gal[STEPS][4];
  float4 gb[STEPS][2];
  {
    const int rr = lane >> 2, q8 = 8 * (lane & 3);          // A planes: four lanes per 64-byte row piece, 16 rows per instruction
    const int br = lane >> 3, bq = 4 * (lane & 7);          // B fp32: eight lanes per 128-byte row piece
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned short* q = Ah + (size_t)min(16 * j + rr, M - 1) * lda + kb + 32 * s + q8;
        gah[s][j] = *reinterpret_cast<const uint4*>(q);
        gal[s][j] = *reinterpret_cast<const uint4*>(q + a_plane);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) gb[s][j] = ld4(Bm + (size_t)(c0 + 8 * j + br) * ldb + kb + 32 * s + bq);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  f32x4 acc0[4], acc1[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { acc0[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    {
      const int rr = lane >> 2, q8 = 8 * (lane & 3);
      const int br = lane >> 3, bq = 4 * (lane & 7);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<uint4*>(iAh + (16 * j + rr) * LDH + q8) = gah[s][j];
        *reinterpret_cast<uint4*>(iAl + (16 * j + rr) * LDH + q8) = gal[s][j];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float a[4] = {gb[s][j].x * w_scale, gb[s][j].y * w_scale, gb[s][j].z * w_scale, gb[s][j].w * w_scale};
        f16x4h hh, ll;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hh[e] = (_Float16)a[e];
          ll[e] = (_Float16)((a[e] - (float)hh[e]) * 2048.0f);
        }
        *reinterpret_cast<f16x4h*>(iBh + (8 * j + br) * LDH + bq) = hh;
        *reinterpret_cast<f16x4h*>(iBl + (8 * j + br) * LDH + bq) = ll;
      }
    }
    f16x8 ah[4], al[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ah[t] = *reinterpret_cast<const f16x8*>(iAh + (16 * t + i) * LDH + 8 * kq);
      al[t] = *reinterpret_cast<const f16x8*>(iAl + (16 * t + i) * LDH + 8 * kq);
    }
    const f16x8 bh = *reinterpret_cast<const f16x8*>(iBh + i * LDH + 8 * kq);
    const f16x8 bl = *reinterpret_cast<const f16x8*>(iBl + i * LDH + 8 * kq);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc0[t], 0, 0, 0);
      acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acc1[t], 0, 0, 0);
      acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acc1[t], 0, 0, 0);
    }
  }
  float* mine = part + wave * (ROWS * COLS);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v)
      mine[(16 * t + 4 * kq + v) * COLS + i] = fmaf(acc1[t][v], 1.0f / 2048.0f, acc0[t][v]) * out_scale;
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

// dst[idx] = src(idx) for idx < n, by the whole workgroup: BATCH loads per thread requested before the first is stored (an
// element loop keeps ONE 4-byte load in flight per thread: 5-7 dependent round trips for the 2-3 K floats staged here)
template <int BATCH, class F>
__device__ __forceinline__ void fill_lds(float* dst, int n, F src) {
  for (int base = threadIdx.x; base < n; base += BATCH * NTHR) {
    float v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int idx = base + u * NTHR;
      v[u] = src(min(idx, n - 1));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int idx = base + u * NTHR;
      if (idx < n) dst[idx] = v[u];
    }
  }
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
  // the last hidden layer of a fused train step: this workgroup's share of the output Linear, ypart[blk][B][64] (columns < O) =
  // act[:, its 16 columns] W2[:, those columns]^T -- the slabs small_mse_kernel adds up (no launch for the output layer)
  const float* W2;
  float* ypart;
  int O;
  // evaluation (model.eval(), train_1.py:112-126): BatchNorm on the running statistics, Dropout the identity, nothing saved.
  // No statistic ties the rows together, so the grid also runs over 64-row blocks (blockIdx.y; Mtot rows in all): every row
  // sees the same contraction order whatever the batch, i.e. the same bits.
  int eval, Mtot;
  // PL_F16X3 descriptors: the input as fp16 planes [rows][K] (h, then l a_plane elements behind: what the launch before left in
  // outp), this layer's output also as planes for the next launch (outp; NULL: nobody reads them)
  const unsigned short* ap;
  unsigned short* outp;
  size_t a_plane, o_plane;
  // the Linear alone, for training batches of 65 ... 512 rows (stats = 1): z = a W^T + b and the partial BatchNorm statistics a
  // tile GEMM's epilogue emits (per 64-row group: column sums, and sums of squares about the group mean -- stat_sum / stat_m2
  // [groups][H], zeros for groups past the last row; NULL: none); the grid runs over the groups like an evaluation's
  int stats;
  float *stat_sum, *stat_m2;
};

// the launch's row block: blockIdx.y * 64 .. + 63 of Mtot rows (evaluation; training launches have one block of B rows)
__device__ __forceinline__ FwdArgs row_block(FwdArgs p) {
  if (!p.eval && !p.stats) { p.Mtot = p.B; return p; }
  const int r0 = blockIdx.y * ROWS;
  p.B = min(ROWS, p.Mtot - r0);
  if (p.stats) p.z += (size_t)r0 * p.H;
  p.a += (size_t)r0 * p.K;
  if (p.ap) p.ap += (size_t)r0 * p.K;
  if (p.outp) p.outp += (size_t)r0 * p.H;
  if (p.resid) p.resid += (size_t)r0 * p.H;
  p.act += (size_t)r0 * p.H;
  if (p.ypart) p.ypart += (size_t)r0 * 64;
  return p;
}

// everything behind the Linear: z (bias NOT yet added; thread tid < 256 holds row tid >> 2, columns c .. c + 3 of block blk)
// what the tail reads from memory besides z: requested BEFORE the contraction, so that its round trip is not the tail's
struct FwdPre { float4 bias, ga, be, rv, rmean, rvar; };
__device__ __forceinline__ FwdPre fwd_prefetch(const FwdArgs& p, int c) {
  const int tid = threadIdx.x, r = tid >> 2;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  FwdPre q = {zero, zero, zero, zero, zero, zero};
  if (tid < 256) {
    q.bias = ld4(p.bias + c);
    if (!p.stats) { q.ga = ld4(p.gamma + c); q.be = ld4(p.beta + c); }
    if (p.resid && r < p.B) q.rv = ld4(p.resid + (size_t)r * p.H + c);
    if (p.eval) { q.rmean = ld4(p.rm + c); q.rvar = ld4(p.rv + c); }
  }
  return q;
}
// the evaluation tail: y = relu(z s + t) (+ residual) with s = gamma / sqrt(running_var + eps), t = (bias - running_mean) s +
// beta -- bn_fold_eval_kernel's fold and the GEMM epilogue's order (elementwise.hip, gemm_epilogue.h)
__device__ __forceinline__ float4 eval_tail(const FwdArgs& p, float4 z, const FwdPre& q, int c) {
  const int tid = threadIdx.x, r = tid >> 2;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid >= 256 || r >= p.B) return zero;
  const float sx = q.ga.x * (1.0f / sqrtf(q.rvar.x + p.eps)), sy = q.ga.y * (1.0f / sqrtf(q.rvar.y + p.eps));
  const float sz = q.ga.z * (1.0f / sqrtf(q.rvar.z + p.eps)), sw = q.ga.w * (1.0f / sqrtf(q.rvar.w + p.eps));
  float4 y;
  y.x = fmaxf(fmaf(z.x, sx, fmaf(q.bias.x - q.rmean.x, sx, q.be.x)), 0.f) + q.rv.x;
  y.y = fmaxf(fmaf(z.y, sy, fmaf(q.bias.y - q.rmean.y, sy, q.be.y)), 0.f) + q.rv.y;
  y.z = fmaxf(fmaf(z.z, sz, fmaf(q.bias.z - q.rmean.z, sz, q.be.z)), 0.f) + q.rv.z;
  y.w = fmaxf(fmaf(z.w, sw, fmaf(q.bias.w - q.rmean.w, sw, q.be.w)), 0.f) + q.rv.w;
  st4(p.act + (size_t)r * p.H + c, y);
  if (p.outp) store_planes4(PlaneDst{p.outp, p.outp + p.o_plane, kActPlaneScale, 2, 0}, (size_t)r * p.H + c, y);
  return y;
}
// Returns the thread's four outputs (zeros outside the batch and for threads >= 256).
__device__ __forceinline__ float4 fwd_tail(const FwdArgs& p, float4 z, const FwdPre& pre, int blk, int c, float4 (*sm)[4]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = tid >> 2;
  const bool live = tid < 256 && r < p.B;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 ga = pre.ga, be = pre.be, rv = pre.rv;
  z.x += pre.bias.x; z.y += pre.bias.y; z.z += pre.bias.z; z.w += pre.bias.w;
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
  if (tid >= 256) return zero;
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
  const float4 out = make_float4(o[0] + rv.x, o[1] + rv.y, o[2] + rv.z, o[3] + rv.w);
  if (live) {
    st4(p.act + (size_t)r * p.H + c, out);
    if (p.outp) store_planes4(PlaneDst{p.outp, p.outp + p.o_plane, kActPlaneScale, 2, 0}, (size_t)r * p.H + c, out);
  }
  return live ? out : zero;
}

// the workgroup's slab of the output Linear (FwdArgs.ypart): hs = the act tile [64][16] from the threads' registers, ws = the
// 16 columns of W2 [O][17] (odd stride: lanes walk o); thread t handles (row idx >> 6, output idx & 63), an fmaf chain over the
// 16 columns in order.  Every thread of the workgroup calls it.
__device__ __forceinline__ void head_slab(const FwdArgs& p, float4 out, int blk, int c0, float* hs, float* ws) {
  const int tid = threadIdx.x;
  if (tid < 256) st4(hs + tid * 4, out);
  {
    float v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = min(tid + u * NTHR, p.O * COLS - 1);
      v[u] = p.W2[(size_t)(idx >> 4) * p.H + c0 + (idx & 15)];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + u * NTHR;
      if (idx < p.O * COLS) ws[(idx >> 4) * 17 + (idx & 15)] = v[u];
    }
  }
  __syncthreads();
  float* mine = p.ypart + (size_t)blk * p.Mtot * 64;
  for (int idx = tid; idx < p.B * 64; idx += NTHR) {
    const int r = idx >> 6, o = idx & 63;
    if (o >= p.O) continue;
    float acc = 0.f;
#pragma unroll
    for (int cc = 0; cc < COLS; ++cc) acc = fmaf(hs[r * COLS + cc], ws[o * 17 + cc], acc);
    mine[idx] = acc;
  }
}

// the tail of the Linear-alone form: z (+ bias) stored, the group's partial statistics (every thread of the workgroup calls it)
__device__ __forceinline__ void stats_tail(const FwdArgs& p, float4 z, const FwdPre& pre, int c, float4 (*sm)[4]) {
  const int tid = threadIdx.x, r = tid >> 2;
  const bool live = tid < 256 && r < p.B;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  z.x += pre.bias.x; z.y += pre.bias.y; z.z += pre.bias.z; z.w += pre.bias.w;
  if (!live) z = zero;
  if (live) st4(p.z + (size_t)r * p.H + c, z);
  if (!p.stat_sum) return;                                   // (kernel-uniform)
  float4 s[1] = {z};
  colsum<1>(s, sm);
  const float cnt = (float)p.B;
  float4 q[1] = {zero};
  if (live) {
    const float dx = z.x - s[0].x / cnt, dy = z.y - s[0].y / cnt, dz = z.z - s[0].z / cnt, dw = z.w - s[0].w / cnt;
    q[0] = make_float4(dx * dx, dy * dy, dz * dz, dw * dw);
  }
  colsum<1>(q, sm);
  if (tid < 4) {
    st4(p.stat_sum + (size_t)blockIdx.y * p.H + c, s[0]);
    st4(p.stat_m2 + (size_t)blockIdx.y * p.H + c, q[0]);
  }
}

template <int STEPS, int ABL = 0>
__global__ __launch_bounds__(NTHR) void small_fwd_kernel(FwdArgs p_in) {
  __shared__ float part[NWAVE * ROWS * COLS];
  __shared__ float stage[NWAVE * STAGE];
  __shared__ float4 sm[4][4];
  const FwdArgs p = row_block(p_in);
  const int blk = col_block(blockIdx.x, gridDim.x);
  const int c0 = blk * COLS;
  const int tid = threadIdx.x;
  const int c = c0 + 4 * (tid & 3);
  if (p.stats && p.B <= 0) {                                 // a statistics group past the last row: zeros (workgroup-uniform)
    if (p.stat_sum && tid < 4) {
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
      st4(p.stat_sum + (size_t)blockIdx.y * p.H + c, zero);
      st4(p.stat_m2 + (size_t)blockIdx.y * p.H + c, zero);
    }
    return;
  }
  const FwdPre pre = fwd_prefetch(p, c);
  if (ABL == 0 && p.ap)
    contract_f16<STEPS>(p.ap, p.a_plane, p.K, p.B, p.W, p.K, c0, kWeightPlaneScale, 1.0f / (kActPlaneScale * kWeightPlaneScale), part,
                        stage);
  else if (ABL != 1) contract<STEPS, false, ABL>(p.a, p.K, p.B, p.W, p.K, c0, part, stage);
  __syncthreads();
  float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 256) z = gather_part(part);
  if (p.stats) { stats_tail(p, z, pre, c, sm); return; }
  const float4 out = p.eval ? eval_tail(p, z, pre, c) : fwd_tail(p, z, pre, blk, c, sm);
  if (p.ypart) {
    if (p.eval) __syncthreads();                             // (every thread is done with part)
    head_slab(p, out, blk, c0, part, stage);                 // (part and stage are free behind the barriers of fwd_tail)
  }
}

// The FIRST hidden layer (LinearModel.w1: in_dim = 34 or 51 inputs, baselineModel.py:76,90-94): the contraction is 34 steps,
// so x (all rows) and the workgroup's 16 rows of W1 sit in LDS and a thread forms its four outputs on the vector unit (an fmaf
// chain over k in index order); the tail is the other layers'.  K <= kFirstMaxK.
constexpr int kFirstMaxK = 256;                        // (64 + 16) rows of K floats in the stage array
__global__ __launch_bounds__(NTHR) void small_first_fwd_kernel(FwdArgs p_in) {
  __shared__ float stage[NWAVE * STAGE];
  __shared__ float4 sm[4][4];
  const FwdArgs p = row_block(p_in);
  static_assert((ROWS + COLS) * kFirstMaxK <= NWAVE * STAGE, "stage array");
  const int blk = col_block(blockIdx.x, gridDim.x);
  const int c0 = blk * COLS;
  const int tid = threadIdx.x, K = p.K;
  const FwdPre pre = fwd_prefetch(p, c0 + 4 * (tid & 3));
  float* xs = stage;
  float* ws = stage + ROWS * K;
  {
    const float* __restrict__ xa = p.a;
    const float* __restrict__ wa = p.W + (size_t)c0 * K;
    const int nx = p.B * K;
    fill_lds<8>(xs, ROWS * K, [&](int idx) { return idx < nx ? xa[idx] : 0.f; });
    fill_lds<2>(ws, COLS * K, [&](int idx) { return wa[idx]; });
  }
  __syncthreads();
  const int c = c0 + 4 * (tid & 3);
  float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 256) {
    const float* xr = xs + (tid >> 2) * K;
    const float* w0 = ws + 4 * (tid & 3) * K;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int k = 0; k < K; ++k) {
      const float xv = xr[k];
      a0 = fmaf(xv, w0[k], a0); a1 = fmaf(xv, w0[K + k], a1); a2 = fmaf(xv, w0[2 * K + k], a2); a3 = fmaf(xv, w0[3 * K + k], a3);
    }
    z = make_float4(a0, a1, a2, a3);
  }
  const float4 out = p.eval ? eval_tail(p, z, pre, c) : fwd_tail(p, z, pre, blk, c, sm);
  if (p.ypart) {                                             // (num_stage = 0: the first layer is the last)
    if (p.eval) __syncthreads();                             // (every thread is done with xs / ws)
    head_slab(p, out, blk, c0, stage, stage + ROWS * COLS);
  }
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

// BatchNorm backward of one layer on the workgroup's columns, from the layer's incoming gradient g (thread tid < 256: row
// tid >> 2, columns c .. c + 3 of block blk): sum dy, sum dy zhat, the coefficients, dz (stored, and returned: zeros outside the
// batch), dgamma, dbeta and the gradient of the bias in front of the BatchNorm.  Every thread of the workgroup calls it.
struct BnLo {
  const float *z, *mean, *rstd, *gamma;
  const uint64_t* bits;
  float *dz, *dgamma, *dbeta, *dbias;
  float kscale;
  int B, H, rowbits;                      // rowbits: the bitmap is in the row format of bn_apply_row (elementwise.hip)
};
// what the tail reads from memory besides g: requested before the contraction that produces g
struct BnPre { float4 zl, mu, rs, ga; uint64_t bw[4]; int bit; };
__device__ __forceinline__ BnPre bn_prefetch(const BnLo& p, int blk, int c) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = tid >> 2;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  BnPre q = {zero, zero, zero, zero, {0, 0, 0, 0}, lane};
  if (tid < 256) {
    q.mu = ld4(p.mean + c); q.rs = ld4(p.rstd + c); q.ga = ld4(p.gamma + c);
    if (r < p.B) {
      q.zl = ld4(p.z + (size_t)r * p.H + c);
      const uint64_t* w = p.bits + (size_t)blk * 16 + wave * 4;
      if (p.rowbits) {
        w = p.bits + (size_t)r * (((p.H + 255) >> 8) * 4) + (c >> 8) * 4;
        q.bit = (c >> 2) & 63;
      }
      const ulonglong2 b01 = *reinterpret_cast<const ulonglong2*>(w), b23 = *reinterpret_cast<const ulonglong2*>(w + 2);
      q.bw[0] = b01.x; q.bw[1] = b01.y; q.bw[2] = b23.x; q.bw[3] = b23.y;
    }
  }
  return q;
}
__device__ __forceinline__ float4 bnbwd_tail(const BnLo& p, float4 g, const BnPre& pre, int blk, int c, float4 (*sm)[4]) {
  const int tid = threadIdx.x;
  const int r = tid >> 2;
  const bool live = tid < 256 && r < p.B;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 zl = pre.zl, mu = pre.mu, rs = pre.rs, ga = pre.ga;
  const uint64_t bw[4] = {pre.bw[0], pre.bw[1], pre.bw[2], pre.bw[3]};
  const int bit = pre.bit;
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
  float4 d = zero;
  if (live) {
    d.x = k0.x * (dv.x - k1.x - zh.x * k2.x); d.y = k0.y * (dv.y - k1.y - zh.y * k2.y);
    d.z = k0.z * (dv.z - k1.z - zh.z * k2.z); d.w = k0.w * (dv.w - k1.w - zh.w * k2.w);
    st4(p.dz + (size_t)r * p.H + c, d);
  }
  return d;
}

struct BwdArgs {
  const float *a_in;                      // the layer's input [B][H] and
  float* dW;                              //   its weight gradient [K][H] = dz^T a_in (extra workgroups), or NULL
  int nblk_dx;                            // workgroups of the dX part (H / 16)
  const float *dz, *W, *addend;           // g = dz W (+ addend) on the workgroup's columns
  float* gout;                            // g is kept here when something later reads it (the skip gradient), or NULL
  BnLo lo;                                // the layer below
  // the layer below is the FIRST layer: its weight gradient dW1 [H][K1] = dz_lo^T x follows in the same workgroup, or NULL
  const float* x1;
  float* dW1;
  int K1;
  int B, H, K;
  // a slice of the AdamW step (parameters whose gradients earlier launches finished and which nothing reads any more) on
  // nblk_adam further workgroups behind the weight-gradient ones: HBM streaming beside the latency-bound layer work
  AdamWRide adam;
  int nblk_dw, nblk_adam;
};

// dW [16 columns of this workgroup][K1] = d^T x over the rows: d (this workgroup's dz of the first layer, from the threads'
// registers) and x through LDS; thread o handles (column o / K1, input o % K1), an fmaf chain over the rows in order
__device__ __forceinline__ void first_wgrad(float4 d, const float* __restrict__ x, float* __restrict__ dW1, int B, int K1, int c0,
                                            float* ds, float* xs) {
  const int tid = threadIdx.x;
  if (tid < 256) st4(ds + tid * 4, d);                   // [row][16]
  fill_lds<8>(xs, B * K1, [&](int idx) { return x[idx]; });
  __syncthreads();
  for (int o = tid; o < COLS * K1; o += NTHR) {
    const int cc = o / K1, k = o - cc * K1;
    float acc = 0.f;
    for (int r = 0; r < B; ++r) acc = fmaf(ds[r * COLS + cc], xs[r * K1 + k], acc);
    dW1[(size_t)(c0 + cc) * K1 + k] = acc;
  }
}

template <int STEPS>
__global__ __launch_bounds__(NTHR) void small_bwd_kernel(BwdArgs p) {
  __shared__ float part[NWAVE * ROWS * COLS];
  __shared__ float stage[NWAVE * STAGE];
  __shared__ float4 sm[12][4];
  if ((int)blockIdx.x >= p.nblk_dx + p.nblk_dw) {       // (workgroup-uniform)
    const AdamWK k = adamw_consts(p.adam);
    adamw_span(p.adam.p, p.adam.g, p.adam.m, p.adam.v, p.adam.n >> 2,
               (int64_t)(blockIdx.x - p.nblk_dx - p.nblk_dw) * NTHR + threadIdx.x, (int64_t)p.nblk_adam * NTHR, k);
    return;
  }
  if ((int)blockIdx.x >= p.nblk_dx) {
    dw_tile(p.dz, p.a_in, p.dW, p.B, p.K, p.H, blockIdx.x - p.nblk_dx);
    return;
  }
  const int blk = col_block(blockIdx.x, p.nblk_dx);
  const int c0 = blk * COLS;
  const int tid = threadIdx.x;
  const int r = tid >> 2, c = c0 + 4 * (tid & 3);
  const bool live = tid < 256 && r < p.B;
  const BnPre pre = bn_prefetch(p.lo, blk, c);
  float4 add = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.addend && live) add = ld4(p.addend + (size_t)r * p.H + c);
  contract<STEPS, true>(p.dz, p.K, p.B, p.W, p.H, c0, part, stage);
  __syncthreads();
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 256) {
    g = gather_part(part);
    g.x += add.x; g.y += add.y; g.z += add.z; g.w += add.w;
    if (p.gout && live) st4(p.gout + (size_t)r * p.H + c, g);
  }
  const float4 d = bnbwd_tail(p.lo, g, pre, blk, c, sm);
  // (part and stage are free: every thread is past the barriers of bnbwd_tail)
  if (p.dW1) first_wgrad(d, p.x1, p.dW1, p.B, p.K1, c0, part, stage);
}

// The TOP of the backward pass: the output Linear (LinearModel.w2, out_dim = 51 or 34 outputs) and the BatchNorm backward of
// the last hidden layer, from dy [B][O]:  g = dy W2 on the workgroup's 16 columns (O steps on the vector unit, dy and the W2
// columns in LDS; kept in gout: it is also the skip gradient of the last residual block), the BatchNorm backward of the top
// layer, dW2[:, columns] = dy^T h, and -- workgroup 0 -- the output bias gradient = column sums of dy.   O <= 64.
struct TopArgs {
  const float *dy, *W2, *h;
  float *gout, *dW2, *db2;
  BnLo lo;
  int B, H, O;
  // the fused train step: the MSE loss = inv_n * (sum of the np partials, in order), and the device step counter ticks here
  // (every dropout kernel of the forward ran before this launch, AdamW runs after it) -- mse_final_kernel's job
  const float* mpart;
  float* loss;
  uint64_t* tick;
  float inv_n;
  int np;
};
__global__ __launch_bounds__(NTHR) void small_top_bwd_kernel(TopArgs p) {
  __shared__ float stage[ROWS * 64 + 64 * COLS + ROWS * COLS];
  __shared__ float4 sm[12][4];
  const int blk = col_block(blockIdx.x, gridDim.x);
  const int c0 = blk * COLS;
  const int tid = threadIdx.x, O = p.O;
  const BnPre pre = bn_prefetch(p.lo, blk, c0 + 4 * (tid & 3));
  float* dys = stage;                    // [64][O]
  float* ws = stage + ROWS * 64;         // [O][16]
  float* hs = ws + 64 * COLS;            // [64][16]
  {
    // (everything the workgroup stages is requested before the first LDS store: 8 + 2 + 2 loads per thread, one round trip)
    const int nd = p.B * O;
    float vd[8], vw[2], vh[2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + u * NTHR;
      vd[u] = idx < nd ? p.dy[idx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int iw = min(tid + u * NTHR, O * COLS - 1), ih = tid + u * NTHR;
      vw[u] = p.W2[(size_t)(iw >> 4) * p.H + c0 + (iw & 15)];
      vh[u] = (ih >> 4) < p.B ? p.h[(size_t)(ih >> 4) * p.H + c0 + (ih & 15)] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + u * NTHR;
      if (idx < ROWS * O) dys[idx] = vd[u];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + u * NTHR;
      if (idx < O * COLS) ws[idx] = vw[u];
      hs[idx] = vh[u];
    }
  }
  __syncthreads();
  const int r = tid >> 2, c = c0 + 4 * (tid & 3);
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 256) {
    const float* dr = dys + r * O;
    const float* w0 = ws + 4 * (tid & 3);
    for (int o = 0; o < O; ++o) {
      const float dv = dr[o];
      const float4 wv = ld4(w0 + o * COLS);
      g.x = fmaf(dv, wv.x, g.x); g.y = fmaf(dv, wv.y, g.y); g.z = fmaf(dv, wv.z, g.z); g.w = fmaf(dv, wv.w, g.w);
    }
    if (r < p.B) st4(p.gout + (size_t)r * p.H + c, g);
  } else {
    // (waves 4-7 have no element of the tail: the weight-gradient columns are theirs, beside the g loop of waves 0-3)
    for (int o = tid - 256; o < O * COLS; o += 256) {
      const int oo = o >> 4, cc = o & 15;
      float acc = 0.f;
      for (int rr = 0; rr < p.B; ++rr) acc = fmaf(dys[rr * O + oo], hs[rr * COLS + cc], acc);
      p.dW2[(size_t)oo * p.H + c0 + cc] = acc;
    }
  }
  (void)bnbwd_tail(p.lo, g, pre, blk, c, sm);
  if (blockIdx.x == 0 && tid < O) {
    float acc = 0.f;
    for (int rr = 0; rr < p.B; ++rr) acc += dys[rr * O + tid];
    p.db2[tid] = acc;
  }
  if (p.loss && blockIdx.x == 0 && tid == 64) {
    float acc = 0.f;
    for (int i = 0; i < p.np; ++i) acc += p.mpart[i];
    p.loss[0] = acc * p.inv_n;
    if (p.tick) p.tick[0] += 1;
  }
}

// y = bias + the NS slabs of the output Linear (FwdArgs.ypart: [NS][B][64]), dpred = coef (y - t), and per workgroup the
// partial sum of (y - t)^2: the MSE forward + backward of the fused train step at small batch (nn.MSELoss, train_1.py:64-66).
// Four lanes per element, a quarter of the slabs each (all of a lane's loads in flight: one round trip), combined in a fixed
// order: (q0 + q1) + (q2 + q3).  64 elements per workgroup.
constexpr int kMseElems = 64;
__global__ __launch_bounds__(256) void small_mse_kernel(const float* __restrict__ ypart, int NS, int B, int O,
                                                        const float* __restrict__ bias, const float* __restrict__ tgt, float coef,
                                                        float* __restrict__ y, float* __restrict__ dpred, float* __restrict__ mpart) {
  __shared__ float red[4];
  const int n = B * O, tid = threadIdx.x;
  const int i = blockIdx.x * kMseElems + (tid >> 2), q = tid & 3;
  const bool ok = i < n;
  const int r = ok ? i / O : 0, o = ok ? i - r * O : 0;
  const size_t slab = (size_t)B * 64;
  const int per = (NS + 3) >> 2;
  const float* src = ypart + (size_t)r * 64 + o + (size_t)(q * per) * slab;
  float a = 0.f;
  for (int s0 = 0; s0 < per; s0 += 16) {
    float u[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) u[j] = (ok && s0 + j < per && q * per + s0 + j < NS) ? src[(size_t)(s0 + j) * slab] : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) a += u[j];
  }
  a += __shfl_xor(a, 1);
  a += __shfl_xor(a, 2);
  float acc = 0.f;
  if (ok && q == 0) {
    if (bias) a += bias[o];
    y[i] = a;
    if (tgt) {
      const float d = a - tgt[i];
      acc = d * d;
      dpred[i] = d * coef;
    }
  }
  if (!tgt) return;                                          // (evaluation: y is all there is; kernel-uniform)
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) acc += __shfl_xor(acc, w);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) mpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

// shapes the layer kernels take: B rows in one tile, 16-column blocks, K split over eight waves in 32-k steps
bool small_layer_ok(int B, int H, int K) {
  static const bool off = [] { const char* e = getenv("POSELIFT_SMALL_LAYER"); return e && e[0] == '0'; }();   // =0: same-box A/B
  if (off || B < 2 || B > ROWS || (H & 15) || H < 16) return false;
  const int steps = K / (NWAVE * 32);
  return K % (NWAVE * 32) == 0 && (steps == 1 || steps == 2 || steps == 4);
}
// ... and the first layer (K = in_dim inputs) / the output layer (O = out_dim outputs) beside such hidden layers
bool small_first_ok(int K) {
  static const bool off = [] { const char* e = getenv("POSELIFT_SMALL_ENDS"); return e && e[0] == '0'; }();    // =0: same-box A/B
  return !off && K >= 1 && K <= kFirstMaxK;
}
bool small_top_ok(int O) {
  static const bool off = [] { const char* e = getenv("POSELIFT_SMALL_ENDS"); return e && e[0] == '0'; }();
  return !off && O >= 1 && O <= 64;
}

// the output Linear of an evaluation forward from the slabs the last hidden layer's launch left: y = bias + sum of slabs
int launch_small_out(const float* ypart, int NS, int M, int O, const float* bias, float* y, hipStream_t s) {
  if (!ypart || !y || NS < 1 || M < 1 || !small_top_ok(O)) PL_FAIL(PL_EINVAL, "small_out: bad arguments");
  hipLaunchKernelGGL(small_mse_kernel, dim3((M * O + kMseElems - 1) / kMseElems), dim3(256), 0, s, ypart, NS, M, O, bias,
                     (const float*)nullptr, 0.f, y, (float*)nullptr, (float*)nullptr);
  PL_CHECK_LAUNCH("small_out");
  return PL_OK;
}

// hidden layer l of an evaluation forward of M rows (any M: the grid runs over 64-row blocks): act = relu(bn_eval(a W^T + b))
// (+ resid); first: the K-input first layer; ypart != NULL: + the output Linear's slabs [H / 16][M][64]
int launch_small_layer_eval(const float* a, const float* W, const float* bias, const float* gamma, const float* beta, float eps,
                            const float* rm, const float* rv, const float* resid, float* act, int M, int H, int K, hipStream_t s,
                            bool first, const float* W2, float* ypart, int O, const unsigned short* a_planes,
                            unsigned short* out_planes) {
  if (first ? !(small_layer_ok(2, H, H) && small_first_ok(K)) : !small_layer_ok(2, H, K))
    PL_FAIL(PL_ESHAPE, "small_layer_eval: H=%d K=%d first=%d", H, K, (int)first);
  if (!a || !W || !bias || !gamma || !beta || !rm || !rv || !act || M < 1) PL_FAIL(PL_EINVAL, "small_layer_eval: bad arguments");
  if ((!first && (!al16(a) || !al16(W))) || !al16(bias) || !al16(gamma) || !al16(beta) || !al16(rm) || !al16(rv) || !al16(act) ||
      !al16(resid))
    PL_FAIL(PL_EINVAL, "small_layer_eval: 16-byte alignment");
  FwdArgs p = {};
  p.a = a; p.W = W; p.bias = bias; p.gamma = gamma; p.beta = beta; p.resid = resid;
  p.rm = const_cast<float*>(rm); p.rv = const_cast<float*>(rv); p.act = act; p.eps = eps;
  p.B = ROWS; p.H = H; p.K = K; p.eval = 1; p.Mtot = M;
  if (ypart) {
    if (!W2 || !small_top_ok(O)) PL_FAIL(PL_EINVAL, "small_layer_eval: output-layer slabs (O=%d)", O);
    p.W2 = W2; p.ypart = ypart; p.O = O;
  }
  if ((a_planes && (first || !al16(a_planes))) || !al16(out_planes)) PL_FAIL(PL_EINVAL, "small_layer_eval: operand planes");
  p.ap = a_planes; p.a_plane = (size_t)M * K; p.outp = out_planes; p.o_plane = (size_t)M * H;
  const dim3 grid(H / COLS, (M + ROWS - 1) / ROWS), block(NTHR);
  void* prof = prof_begin_flops(2.0 * M * H * K, s);
  if (first) hipLaunchKernelGGL(small_first_fwd_kernel, grid, block, 0, s, p);
  else switch (K / (NWAVE * 32)) {
    case 1: hipLaunchKernelGGL(small_fwd_kernel<1>, grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL(small_fwd_kernel<2>, grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL((small_fwd_kernel<4, 0>), grid, block, 0, s, p); break;
  }
  prof_end(prof, s);
  PL_CHECK_LAUNCH("small_layer_eval");
  return PL_OK;
}

// z [M][H] = a W^T + bias for M <= 512 rows (any M; one 64-row block x 16 columns per workgroup) and, stat_sum != NULL, the
// groups partial BatchNorm statistics of a tile GEMM's epilogue ([groups][H] each; groups >= ceil(M / 64): the rest zeros).
// a_planes != NULL: the input as fp16 planes [M][K] (h, l), contraction as three fp16 MFMAs per product; else a fp32, exact.
int launch_small_linear_stats(const float* a, const unsigned short* a_planes, const float* W, const float* bias, float* z, int M,
                              int H, int K, float* stat_sum, float* stat_m2, int groups, hipStream_t s) {
  if (!small_layer_ok(2, H, K) || M < 1 || groups < (M + ROWS - 1) / ROWS)
    PL_FAIL(PL_ESHAPE, "small_linear_stats: M=%d H=%d K=%d groups=%d", M, H, K, groups);
  if ((!a && !a_planes) || !W || !bias || !z || (stat_sum != nullptr) != (stat_m2 != nullptr))
    PL_FAIL(PL_EINVAL, "small_linear_stats: bad arguments");
  if (!al16(a) || !al16(a_planes) || !al16(W) || !al16(bias) || !al16(z) || !al16(stat_sum) || !al16(stat_m2))
    PL_FAIL(PL_EINVAL, "small_linear_stats: 16-byte alignment");
  FwdArgs p = {};
  p.a = a; p.W = W; p.bias = bias; p.z = z; p.B = ROWS; p.H = H; p.K = K; p.Mtot = M; p.stats = 1;
  p.stat_sum = stat_sum; p.stat_m2 = stat_m2;
  p.ap = a_planes; p.a_plane = (size_t)M * K;
  if (!a) p.a = reinterpret_cast<const float*>(a_planes);     // (never read: the planes form is taken)
  const dim3 grid(H / COLS, stat_sum ? groups : (M + ROWS - 1) / ROWS), block(NTHR);
  void* prof = prof_begin_flops(2.0 * M * H * K, s);
  switch (K / (NWAVE * 32)) {
    case 1: hipLaunchKernelGGL(small_fwd_kernel<1>, grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL(small_fwd_kernel<2>, grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL((small_fwd_kernel<4, 0>), grid, block, 0, s, p); break;
  }
  prof_end(prof, s);
  PL_CHECK_LAUNCH("small_linear_stats");
  return PL_OK;
}

int launch_small_layer_fwd(const float* a, const float* W, const float* bias, const float* gamma, const float* beta, float eps,
                           float momentum, float* rm, float* rv, int64_t* nbt, float* mean, float* rstd, const float* resid,
                           float* z, float* act, uint64_t* bits, int B, int H, int K, float pdrop, uint64_t seed, uint64_t step,
                           int layer, const uint64_t* inject_keep, hipStream_t s, const uint64_t* step_dev, bool first,
                           const float* W2, float* ypart, int O, const unsigned short* a_planes, unsigned short* out_planes) {
  if (first ? !(small_layer_ok(B, H, H) && small_first_ok(K)) : !small_layer_ok(B, H, K))
    PL_FAIL(PL_ESHAPE, "small_layer_fwd: B=%d H=%d K=%d first=%d", B, H, K, (int)first);
  if (!a || !W || !bias || !gamma || !beta || !mean || !rstd || !z || !act || !bits || (rm != nullptr) != (rv != nullptr))
    PL_FAIL(PL_EINVAL, "small_layer_fwd: bad arguments");
  if ((!first && (!al16(a) || !al16(W))) || !al16(bias) || !al16(gamma) || !al16(beta) || !al16(mean) || !al16(rstd) || !al16(z) ||
      !al16(act) || !al16(resid) || !al16(rm) || !al16(rv) || !al16(bits))
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
  if (ypart) {
    if (!W2 || !small_top_ok(O)) PL_FAIL(PL_EINVAL, "small_layer_fwd: output-layer slabs (O=%d)", O);
    p.W2 = W2; p.ypart = ypart; p.O = O;
  }
  if ((a_planes && (first || !al16(a_planes))) || !al16(out_planes)) PL_FAIL(PL_EINVAL, "small_layer_fwd: operand planes");
  p.ap = a_planes; p.a_plane = (size_t)B * K; p.outp = out_planes; p.o_plane = (size_t)B * H;
  const dim3 grid(H / COLS), block(NTHR);
  void* prof = prof_begin_flops(2.0 * B * H * K, s);
  if (first) {
    hipLaunchKernelGGL(small_first_fwd_kernel, grid, block, 0, s, p);
  } else {
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
  }
  prof_end(prof, s);
  PL_CHECK_LAUNCH("small_layer_fwd");
  return PL_OK;
}

namespace {
int fill_lo(BnLo& lo, const SmallBnLayer& b, int B, int H, float kscale, float* dz_lo, const char* who) {
  if (!b.z || !b.bits || !b.mean || !b.rstd || !b.gamma || !dz_lo || !b.dgamma || !b.dbeta || !b.dbias)
    PL_FAIL(PL_EINVAL, "%s: bad arguments (layer below)", who);
  if (!al16(b.z) || !al16(b.bits) || !al16(b.mean) || !al16(b.rstd) || !al16(b.gamma) || !al16(dz_lo) || !al16(b.dgamma) ||
      !al16(b.dbeta) || !al16(b.dbias))
    PL_FAIL(PL_EINVAL, "%s: 16-byte alignment (layer below)", who);
  lo.z = b.z; lo.mean = b.mean; lo.rstd = b.rstd; lo.gamma = b.gamma; lo.bits = b.bits; lo.dz = dz_lo;
  lo.dgamma = b.dgamma; lo.dbeta = b.dbeta; lo.dbias = b.dbias; lo.kscale = kscale; lo.B = B; lo.H = H; lo.rowbits = b.rowbits ? 1 : 0;
  return PL_OK;
}
}  // namespace

int launch_small_layer_bwd(const float* dz, const float* W, const float* addend, float* gout, int B, int H, int K,
                           const SmallBnLayer& below, float kscale, float* dz_lo, hipStream_t s, const float* a_in, float* dW,
                           const float* x1, float* dW1, int K1, const AdamWRide* adam) {
  if (!small_layer_ok(B, H, K)) PL_FAIL(PL_ESHAPE, "small_layer_bwd: B=%d H=%d K=%d", B, H, K);
  if (!dz || !W || dz_lo == dz) PL_FAIL(PL_EINVAL, "small_layer_bwd: bad arguments");
  if (!al16(dz) || !al16(W) || !al16(addend) || !al16(gout)) PL_FAIL(PL_EINVAL, "small_layer_bwd: 16-byte alignment");
  BwdArgs p = {};
  PL_TRY(fill_lo(p.lo, below, B, H, kscale, dz_lo, "small_layer_bwd"));
  p.dz = dz; p.W = W; p.addend = addend; p.gout = gout;
  p.B = B; p.H = H; p.K = K;
  p.nblk_dx = H / COLS;
  int extra = 0;
  if (dW) {
    if (!a_in || !al16(a_in) || !al16(dW) || (K & 127) || (H & 63)) PL_FAIL(PL_EINVAL, "small_layer_bwd: weight-gradient part (K=%d H=%d)", K, H);
    p.a_in = a_in; p.dW = dW;
    extra = (K / 128) * (H / 64);
  }
  if (dW1) {
    if (!x1 || !small_first_ok(K1)) PL_FAIL(PL_EINVAL, "small_layer_bwd: first-layer weight gradient (K1=%d)", K1);
    p.x1 = x1; p.dW1 = dW1; p.K1 = K1;
  }
  p.nblk_dw = extra;
  if (adam && adam->n > 0) {
    if (!adam->p || !adam->g || !adam->m || !adam->v || (adam->n & 3) || !al16(adam->p) || !al16(adam->g) || !al16(adam->m) ||
        !al16(adam->v) || (adam->lr_dev != nullptr) != (adam->t_dev != nullptr))
      PL_FAIL(PL_EINVAL, "small_layer_bwd: AdamW slice");
    p.adam = *adam;
    // ~2 K float4 per workgroup pass: 1 M parameters on 64 workgroups
    static const int cap = [] { const char* e = getenv("POSELIFT_SMALL_ADAM_BLOCKS"); return e ? atoi(e) : 64; }();   // (A/B)
    p.nblk_adam = (int)std::min<int64_t>(cap, (adam->n / 4 + NTHR - 1) / NTHR);
    extra += p.nblk_adam;
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

int small_mse_partials(int B, int O) { return (B * O + kMseElems - 1) / kMseElems; }

int launch_small_mse(const float* ypart, int NS, int B, int O, const float* bias, const float* tgt, float grad_scale, float* y,
                     float* dpred, float* mpart, hipStream_t s) {
  if (!ypart || !tgt || !y || !dpred || !mpart || NS < 1 || B < 1 || B > ROWS || !small_top_ok(O))
    PL_FAIL(PL_EINVAL, "small_mse: bad arguments");
  const int np = small_mse_partials(B, O);
  const float coef = grad_scale * 2.0f / (float)(B * O);
  hipLaunchKernelGGL(small_mse_kernel, dim3(np), dim3(256), 0, s, ypart, NS, B, O, bias, tgt, coef, y, dpred, mpart);
  PL_CHECK_LAUNCH("small_mse");
  return PL_OK;
}

int launch_small_top_bwd(const float* dy, const float* W2, const float* h, int B, int H, int O, float* gout, float* dW2,
                         float* db2, const SmallBnLayer& top, float kscale, float* dz_top, hipStream_t s, const float* mpart,
                         int np, float inv_n, float* loss, uint64_t* tick) {
  if (!small_layer_ok(B, H, H) || !small_top_ok(O)) PL_FAIL(PL_ESHAPE, "small_top_bwd: B=%d H=%d O=%d", B, H, O);
  if (!dy || !W2 || !h || !gout || !dW2 || !db2 || !al16(gout) || !al16(h)) PL_FAIL(PL_EINVAL, "small_top_bwd: bad arguments");
  TopArgs p = {};
  PL_TRY(fill_lo(p.lo, top, B, H, kscale, dz_top, "small_top_bwd"));
  p.dy = dy; p.W2 = W2; p.h = h; p.gout = gout; p.dW2 = dW2; p.db2 = db2; p.B = B; p.H = H; p.O = O;
  if (loss) {
    if (!mpart || np < 1) PL_FAIL(PL_EINVAL, "small_top_bwd: loss partials");
    p.mpart = mpart; p.np = np; p.inv_n = inv_n; p.loss = loss; p.tick = tick;
  }
  hipLaunchKernelGGL(small_top_bwd_kernel, dim3(H / COLS), dim3(NTHR), 0, s, p);
  PL_CHECK_LAUNCH("small_top_bwd");
  return PL_OK;
}

}  // namespace pl
