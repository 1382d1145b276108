// The two skinny Linear layers of the lifter: 34 -> H (LinearModel.w1, baselineModel.py:67,90) and
// H -> 51 (LinearModel.w2, :77,100), forward and backward.  One side of each GEMM is only 34 / 51
// wide, so a 128x128 LDS-tiled MFMA tile is mostly padding (the generic kernel's guarded edge path
// takes 24-32 us per call).  Each of these ops is really ONE pass over a (B x H) activation
// (16.8 MB, ~3.5 us at HBM rate) with ~0.3-0.5 GFLOP attached.
//
// Design: MFMA straight from registers, no LDS operand tiles, one wavefront = one independent
// task (a 32-column strip x a 64-row group), 2048 tasks = 8 per CU.  v_mfma_f32_32x32x2_f32
// wants A[i][k], B[k][j] with i/j = lane&31 and k = lane>>5: the wide tensor is always the
// operand whose 32 lanes run along its contiguous (column) dimension, so its loads and stores
// are 128-byte segments; the narrow operand is small enough to live in L1/L2.
// (Two earlier VALU designs -- narrow operand through scalar loads, then through broadcast LDS
// reads -- ran 26-67 us per call: one wave per SIMD with 34-52-long dependent FMA chains.
// Round 2 tried wide_out with 128 x 128 workgroup tiles, both operands staged in LDS and the columns dealt
// round-robin over four MFMA tiles so that a lane stores 16 bytes on 512-byte row segments: 15.9 / 15.6 us against
// 13.9 / 11.1 us for this form -- these kernels are bound by the ~3 us of fp32 MFMA issue per wave plus un-overlapped
// load and store latency at one wave per SIMD, not by the store pattern.  Not kept.)
// (Also tried after the staging fix in skinny_wide_out_kernel: all 96 loads of a wide_in wave, and all 48 16-byte loads of a
// narrow_out wave, requested in ONE batch instead of two -- 15.8 / 14.3 / 12.9 us against 13.6 / 11.1 / 12.3 us for two
// batches: past ~48 requests per wave the queue, not the round trip, is what a wave waits for; and the saved z of the BNR
// epilogue requested before the staging -- 14.8 against 14.6 us; and, again after the staging fix, the workgroup's 64 x 128
// output block through LDS into 16-byte stores on 512-byte row segments -- 10.3 / 14.9 us against 10.0 / 14.5 us.  Not kept.)
// (Round 3, after csrc/small_layer.hip found per-lane row fetches at a quarter of the coalesced rate: narrow_out with a
// workgroup's 64 rows x 128 k of h and of W arriving as 512-byte row pieces and passing through LDS into 16x16x4 MFMA
// fragments -- 15.2 us against 11.6 us for the form below; wide_out's 128 weight columns ([H][K] case) staged through LDS
// instead of 17 per-lane 4-byte fetches -- 9.35 against 9.2 us.  These kernels sit on their fp32 MFMA issue (0.54 GFLOP at
// 157 TF = 3.4 us chip-wide) plus one exposed round trip, not on the request pattern.  Not kept.)
#include "pl_internal.h"

namespace pl {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NTHR = 256;
constexpr int RG = 64;     // rows per task = one 64-row BatchNorm statistics group (2048 tasks at B=4096:
                           // 8 waves per CU; with 128-row tasks the kernels were load-latency bound)

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// out[r][c] = bias[c] + sum_k X[r][k] * w_c[k]      X: [B][K] narrow, out: [B][H] wide
//   WT = false: w_c[k] = W[c*K + k]   (forward of the input layer: W is [H][K])
//   WT = true : w_c[k] = W[k*H + c]   (backward of the output layer: g = dy W5, W5 is [K][H])
// STATS: BatchNorm partial statistics (sum, M2 about the group mean) per 64-row group.
// The workgroup's four waves take four neighbouring column strips of the SAME row group, so the
// 64 narrow rows are staged once in LDS (coalesced) and read back as MFMA A fragments.
// BNR (g = dy W5 only): BatchNorm-backward pass 1 of the top hidden layer on the block just produced, as the planes GEMM's
// epilogue does for the layers below (GemmArgs::bnr_*): per 64-row group and column sum dy and sum dy*zhat
// (dy = g * bit * kscale), per (group, 32-column strip) max |dy| and max |zhat|.  Whole groups only (B % 64 == 0).
struct SkinnyBnr {
  const float* z; const uint64_t* bits; const float* mean; const float* rstd; float kscale;
  float* part_dy; float* part_dyz; float* amax;
};

template <int K, bool WT, bool STATS, bool BNR = false>
__global__ __launch_bounds__(NTHR) void skinny_wide_out_kernel(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ out, int B, int H, float* __restrict__ stat_sum, float* __restrict__ stat_m2, SkinnyBnr bn) {
  constexpr int KS = (K + 1) / 2;            // MFMA steps, 2 k each
  constexpr int KL = 2 * KS + 1;             // LDS row stride (odd: conflict-free column reads)
  __shared__ float xs[RG * KL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int strips4 = H / 128;               // workgroups per row group (4 strips of 32 each)
  const int rg = blockIdx.x / strips4;
  const int c = ((blockIdx.x % strips4) * 4 + wave) * 32 + j;
  const int r_base = rg * RG;
  // the lane's weight column first: its loads are in flight while the narrow rows are staged (they do not depend on the LDS)
  float wb[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 2 * s + h;
    wb[s] = (k < K) ? (WT ? W[(size_t)k * H + c] : W[(size_t)c * K + k]) : 0.f;
  }
  const float b = bias ? bias[c] : 0.f;
  if (r_base + RG <= B) {
    // the 64 narrow rows are ONE contiguous run of 64 K floats (16-byte aligned: 64 K * 4 is a multiple of 16): three or four
    // 16-byte loads per thread, all requested before the first LDS write.  (The element loop below kept one 4-byte load in
    // flight per thread: nine -- K = 51: thirteen -- dependent round trips before the first MFMA, most of the kernel's time.)
    constexpr int N4 = RG * K / 4, IT = (N4 + NTHR - 1) / NTHR;
    static_assert((RG * K) % 4 == 0, "64 rows of K floats are whole float4s");
    const float4* __restrict__ src = reinterpret_cast<const float4*>(X + (size_t)r_base * K);
    float4 v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) v[i] = src[min((int)threadIdx.x + i * NTHR, N4 - 1)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int idx = threadIdx.x + i * NTHR;
      if (idx < N4) {
        const float q[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int e = idx * 4 + jj, r = e / K, k = e - r * K;
          xs[r * KL + k] = q[jj];
        }
      }
    }
  } else {
    for (int e = threadIdx.x; e < RG * K; e += NTHR) {
      const int r = e / K, k = e - r * K;
      xs[r * KL + k] = (r_base + r < B) ? X[(size_t)(r_base + r) * K + k] : 0.f;
    }
  }
  if (K & 1)
    for (int r = threadIdx.x; r < RG; r += NTHR) xs[r * KL + K] = 0.f;
  __syncthreads();
#pragma unroll 1
  for (int g2 = 0; g2 < RG / 64; ++g2) {
    const int g0 = r_base + g2 * 64;
    if (g0 >= B) break;
    f32x16 acc[2];
    float xa[2][KS];                         // all A fragments first: one LDS round trip, not one per MFMA
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      const float* xr = xs + (g2 * 64 + t * 32 + j) * KL + h;
#pragma unroll
      for (int s = 0; s < KS; ++s) xa[t][s] = xr[2 * s];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[0][s], wb[s], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[1][s], wb[s], acc[1], 0, 0, 0);
    }
    float ssum = 0.f;
    if (g0 + 64 <= B) {                      // whole group: 32 unguarded 128-byte row stores per half-wave
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[t][r] + b;
          acc[t][r] = v;
          out[(size_t)(g0 + t * 32 + acc_row(r, h)) * H + c] = v;
          ssum += v;
        }
    } else {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = g0 + t * 32 + acc_row(r, h);
        const float v = acc[t][r] + b;
        acc[t][r] = v;
        if (row < B) {
          out[(size_t)row * H + c] = v;
          ssum += v;
        }
      }
    }
    if (BNR) {
      const int wpr = ((H + 255) >> 8) * 4;
      const uint64_t* bw = bn.bits + (size_t)(c >> 8) * 4 + (c & 3);
      const int bit = (c & 255) >> 2;
      const float mu = bn.mean[c], rs = bn.rstd[c];
      float zz[2][16];
      uint64_t ww[2][16];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const size_t row = (size_t)(g0 + t * 32 + acc_row(r, h));
          zz[t][r] = bn.z[row * H + c];
          ww[t][r] = bw[row * wpr];
        }
      float s1 = 0.f, s2 = 0.f, mxd = 0.f, mxz = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float dv = ((ww[t][r] >> bit) & 1ull) ? acc[t][r] * bn.kscale : 0.f;
          const float zh = (zz[t][r] - mu) * rs;
          s1 += dv;
          s2 = fmaf(dv, zh, s2);
          mxd = fmaxf(mxd, fabsf(dv)); mxz = fmaxf(mxz, fabsf(zh));
        }
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      if (h == 0) {
        bn.part_dy[(size_t)(g0 >> 6) * H + c] = s1;
        bn.part_dyz[(size_t)(g0 >> 6) * H + c] = s2;
      }
      if (bn.amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mxd = fmaxf(mxd, __shfl_xor(mxd, o)); mxz = fmaxf(mxz, __shfl_xor(mxz, o)); }
        if (lane == 0) {
          float* q = bn.amax + ((size_t)(g0 >> 6) * (H >> 5) + (c >> 5)) * 2;
          q[0] = mxd; q[1] = mxz;
        }
      }
    }
    if (STATS) {
      ssum += __shfl_xor(ssum, 32);
      const int cnt = min(64, B - g0);
      const float mean = ssum / (float)cnt;
      float m2 = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float dv = acc[t][r] - mean;
          if (g0 + t * 32 + acc_row(r, h) < B) m2 = fmaf(dv, dv, m2);
        }
      m2 += __shfl_xor(m2, 32);
      if (h == 0) {
        stat_sum[(size_t)(g0 >> 6) * H + c] = ssum;
        stat_m2[(size_t)(g0 >> 6) * H + c] = m2;
      }
    }
  }
}

// part[chunk][k][c] = sum_{r in chunk} X[r][k] * D[r][c]     X: [B][K] narrow, D: [B][H] wide
// (dW0^T = x^T dz0 and dW5 = dy^T h).  MFMA rows = k (two 32-row tiles; rows >= K compute garbage that
// is never stored), columns = a 32-wide strip of D, contraction over 64 rows per wavefront: both
// operands are read in 128-byte row segments.  The workgroup's four wavefronts take four consecutive
// 64-row groups of the SAME strip and add their accumulators through LDS in a fixed order, so a
// chunk is 256 rows: a quarter of the partial-sum traffic of one partial per wavefront (it was as
// large as the read of D itself), and the combine kernel walks 16 partials instead of 64 at B = 4096.
constexpr int IN_CHUNK = 4 * RG;

// xsum (optional): xsum[chunk][k] = sum over the chunk's rows of X[r][k] -- the bias gradient of the output layer is the
// column sum of dy, and the first strip's workgroup of every chunk has all of dy's rows in its A fragments anyway.
template <int K>
__global__ __launch_bounds__(NTHR) void skinny_wide_in_kernel(const float* __restrict__ X,
                                                              const float* __restrict__ D, int B, int H,
                                                              float* __restrict__ part, float* __restrict__ xsum) {
  __shared__ float red[3][32 * 64];          // accumulators of waves 1..3
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int strips = H / 32;
  const int ch = blockIdx.x / strips;
  const int c = (blockIdx.x % strips) * 32 + j;
  const int r_base = ch * IN_CHUNK + wave * RG;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int k1 = min(32 + j, K - 1);         // second k tile: clamp instead of masking (rows >= K are dropped)
  const bool do_xsum = xsum != nullptr && (blockIdx.x % strips) == 0;      // (workgroup-uniform)
  float xs0 = 0.f, xs1 = 0.f;                // this lane's rows (parity h) of columns j and 32 + j
  if (r_base + RG <= B) {                    // whole group: no guards
    // Loads in explicit batches of 16 row pairs (48 per wave in flight), pinned above their MFMAs: left to
    // itself hipcc keeps ONE iteration's three loads in flight and waits for them before every MFMA
    // pair -- 32 dependent HBM round trips per wave, 17 us for a 16.8 MB pass.
    const float* __restrict__ dp = D + (size_t)(r_base + h) * H + c;
    const float* __restrict__ xp = X + (size_t)(r_base + h) * K;
#pragma unroll
    for (int b0 = 0; b0 < RG / 2; b0 += 16) {
      float d[16], a0[16], a1[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        d[s] = dp[(size_t)2 * (b0 + s) * H];
        a0[s] = xp[2 * (b0 + s) * K + j];
        a1[s] = xp[2 * (b0 + s) * K + k1];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], d[s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], d[s], acc[1], 0, 0, 0);
      }
      if (do_xsum) {
#pragma unroll
        for (int s = 0; s < 16; ++s) { xs0 += a0[s]; xs1 += a1[s]; }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll 4
    for (int s = 0; s < RG / 2; ++s) {
      const int r = r_base + 2 * s + h;
      const bool ok = r < B;
      const float d = ok ? D[(size_t)r * H + c] : 0.f;
      const float a0 = ok ? X[(size_t)r * K + j] : 0.f;
      const float a1 = ok ? X[(size_t)r * K + k1] : 0.f;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, d, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, d, acc[1], 0, 0, 0);
      xs0 += a0; xs1 += a1;
    }
  }
  if (do_xsum) {
    // rows of both parities, then the four waves in wave order (through the tail of `red`, unused until the barrier below)
    xs0 += __shfl_xor(xs0, 32); xs1 += __shfl_xor(xs1, 32);
    float* xr = &red[2][32 * 64 - 4 * 64];
    if (h == 0) { xr[wave * 64 + j] = xs0; xr[wave * 64 + 32 + j] = xs1; }
    __syncthreads();
    if (wave == 0 && h == 0) {
      const float t0 = ((xr[j] + xr[64 + j]) + xr[128 + j]) + xr[192 + j];
      const float t1 = ((xr[32 + j] + xr[64 + 32 + j]) + xr[128 + 32 + j]) + xr[192 + 32 + j];
      xsum[(size_t)ch * K + j] = t0;
      if (32 + j < K) xsum[(size_t)ch * K + 32 + j] = t1;
    }
    __syncthreads();
  }
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave - 1][(t * 16 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = (t * 16 + r) * 64 + lane;
        const float v = ((acc[t][r] + red[0][q]) + red[1][q]) + red[2][q];
        const int k = t * 32 + acc_row(r, h);
        if (k < K) part[((size_t)ch * K + k) * H + c] = v;
      }
  }
}

// out = sum over chunks of part[chunk][i], i = k*H + c; TRANS writes out[c*K + k] (dW0 is [H][K]).
// block = 16 outputs x 16 chunk-parts, fixed-order combine through LDS.
template <bool TRANS>
__global__ __launch_bounds__(NTHR) void skinny_reduce_kernel(const float* __restrict__ part, int nchunk,
                                                             int K, int H, float* __restrict__ out) {
  __shared__ float red[16][16];
  const int il = threadIdx.x & 15, p = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + il;
  const int n = K * H;
  float a = 0.f;
  if (i < n)
    for (int s = p; s < nchunk; s += 16) a += part[(size_t)s * n + i];
  red[p][il] = a;
  __syncthreads();
  if (p == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][il];
    if (TRANS) {
      const int k = i / H, c = i - k * H;
      out[(size_t)c * K + k] = t;
    } else {
      out[i] = t;
    }
  }
}

// part[split][r][n] = sum_{k in split} h[r][k] * W[n][k]   h: [B][H] wide (read once), n < N <= 64
// MFMA rows = 32 rows of h, columns = two 32-wide tiles of output features, contraction over a
// 1/splits slice of H.  Both operands are k-contiguous: a lane fetches 4 consecutive k with one
// 16-byte load and feeds 4 MFMAs (any k order shared by A and B is a valid contraction order).
__global__ __launch_bounds__(NTHR) void skinny_narrow_out_kernel(const float* __restrict__ hmat,
                                                                 const float* __restrict__ W, int B, int H,
                                                                 int N, int splits, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int task = blockIdx.x * 4 + wave;
  const int rt = task / splits, sp = task - rt * splits;
  if (rt * 32 >= B) return;
  const int kper = H / splits;
  const int row = min(rt * 32 + j, B - 1);
  const float* __restrict__ ap = hmat + (size_t)row * H + sp * kper + 4 * h;
  const float* __restrict__ b0p = W + (size_t)min(j, N - 1) * H + sp * kper + 4 * h;
  const float* __restrict__ b1p = W + (size_t)min(32 + j, N - 1) * H + sp * kper + 4 * h;
  // columns >= N read a clamped (valid) row of W and compute values the combine kernel never reads
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // loads in explicit batches of 8 k-steps (24 x 16 B per lane in flight), pinned above their MFMAs:
  // with guarded one-step-ahead loads every iteration was a dependent memory round trip (19.6 us)
#define PL_NO_MFMA8(A, B0, B1)                                                          \
  do {                                                                                  \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((A).x, (B0).x, acc[0], 0, 0, 0);      \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((A).x, (B1).x, acc[1], 0, 0, 0);      \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((A).y, (B0).y, acc[0], 0, 0, 0);      \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((A).y, (B1).y, acc[1], 0, 0, 0);      \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((A).z, (B0).z, acc[0], 0, 0, 0);      \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((A).z, (B1).z, acc[1], 0, 0, 0);      \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((A).w, (B0).w, acc[0], 0, 0, 0);      \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((A).w, (B1).w, acc[1], 0, 0, 0);      \
  } while (0)
  int kb = 0;
  for (; kb + 64 <= kper; kb += 64) {
    float4 a[8], b0[8], b1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      a[q] = *reinterpret_cast<const float4*>(ap + kb + 8 * q);
      b0[q] = *reinterpret_cast<const float4*>(b0p + kb + 8 * q);
      b1[q] = *reinterpret_cast<const float4*>(b1p + kb + 8 * q);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 8; ++q) PL_NO_MFMA8(a[q], b0[q], b1[q]);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (; kb < kper; kb += 8) {               // k slices that are not a multiple of 64 (generic hidden sizes)
    const float4 a = *reinterpret_cast<const float4*>(ap + kb);
    const float4 b0 = *reinterpret_cast<const float4*>(b0p + kb);
    const float4 b1 = *reinterpret_cast<const float4*>(b1p + kb);
    PL_NO_MFMA8(a, b0, b1);
  }
#undef PL_NO_MFMA8
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int orow = rt * 32 + acc_row(r, h);
      if (orow < B) part[((size_t)sp * B + orow) * 64 + t * 32 + j] = acc[t][r];
    }
}

// y[r][n] = bias[n] + sum_s part[s][r][n]   (n < N of the 64 padded columns)
__global__ __launch_bounds__(NTHR) void skinny_narrow_out_reduce_kernel(const float* __restrict__ part,
                                                                        int splits, int B, int N,
                                                                        const float* __restrict__ bias,
                                                                        float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * NTHR + threadIdx.x;      // i = r*64 + n
  if (i >= (int64_t)B * 64) return;
  const int n = (int)(i & 63);
  if (n >= N) return;
  float a = bias ? bias[n] : 0.f;
  for (int s = 0; s < splits; ++s) a += part[(size_t)s * B * 64 + i];
  y[(size_t)(i >> 6) * N + n] = a;
}

}  // namespace

bool skinny_supported(int K, int H) { return (K == 34 || K == 51) && H % 128 == 0; }
int skinny_chunks(int B) { return (B + RG - 1) / RG; }           // row tasks (wide_out; upper bound for wide_in)
int skinny_in_chunks(int B) { return (B + IN_CHUNK - 1) / IN_CHUNK; }
int skinny_stat_groups(int B) { return (B + 63) / 64; }

// forward of the input layer (+ BN statistics)  /  g = dy W5 (transposed weights, no bias)
int launch_skinny_wide_out(const float* X, const float* W, const float* bias, float* out, int B, int K,
                           int H, bool w_transposed, float* stat_sum, float* stat_m2, hipStream_t s, const GemmArgs* bnr) {
  if (!skinny_supported(K, H)) PL_FAIL(PL_ESHAPE, "skinny_wide_out: K=%d H=%d not specialised", K, H);
  dim3 grid(skinny_chunks(B) * (H / 128)), block(NTHR);
  SkinnyBnr bn = {};
  if (bnr && bnr->bnr_z) {
    if (!w_transposed || stat_sum || (B & 63)) PL_FAIL(PL_EINVAL, "skinny_wide_out: BatchNorm-backward epilogue only on g = dy W, whole groups");
    bn = SkinnyBnr{bnr->bnr_z, bnr->bnr_bits, bnr->bnr_mean, bnr->bnr_rstd, bnr->bnr_kscale, bnr->bnr_part_dy, bnr->bnr_part_dyz,
                   bnr->bnr_amax};
    if (K == 34) hipLaunchKernelGGL((skinny_wide_out_kernel<34, true, false, true>), grid, block, 0, s, X, W, bias, out, B, H, stat_sum, stat_m2, bn);
    else hipLaunchKernelGGL((skinny_wide_out_kernel<51, true, false, true>), grid, block, 0, s, X, W, bias, out, B, H, stat_sum, stat_m2, bn);
    PL_CHECK_LAUNCH("skinny_wide_out");
    return PL_OK;
  }
#define PL_SK(KK, WT, ST) hipLaunchKernelGGL((skinny_wide_out_kernel<KK, WT, ST>), grid, block, 0, s, X, W, bias, out, B, H, stat_sum, stat_m2, bn)
  const bool st = stat_sum != nullptr;
  if (K == 34 && !w_transposed) { if (st) PL_SK(34, false, true); else PL_SK(34, false, false); }
  else if (K == 34) { if (st) PL_SK(34, true, true); else PL_SK(34, true, false); }
  else if (K == 51 && !w_transposed) { if (st) PL_SK(51, false, true); else PL_SK(51, false, false); }
  else { if (st) PL_SK(51, true, true); else PL_SK(51, true, false); }
#undef PL_SK
  PL_CHECK_LAUNCH("skinny_wide_out");
  return PL_OK;
}

// out = X^T D, X [B][K], D [B][H]; out is [K][H] or, transposed, [H][K]; part: chunks*K*H floats
int launch_skinny_wide_in(const float* X, const float* D, float* out, int B, int K, int H,
                          bool out_transposed, float* part, hipStream_t s, bool reduce, float* xsum) {
  if (!skinny_supported(K, H)) PL_FAIL(PL_ESHAPE, "skinny_wide_in: K=%d H=%d not specialised", K, H);
  const int nc = skinny_in_chunks(B);
  dim3 grid(nc * (H / 32)), block(NTHR);
  if (K == 34) hipLaunchKernelGGL((skinny_wide_in_kernel<34>), grid, block, 0, s, X, D, B, H, part, xsum);
  else hipLaunchKernelGGL((skinny_wide_in_kernel<51>), grid, block, 0, s, X, D, B, H, part, xsum);
  PL_CHECK_LAUNCH("skinny_wide_in");
  if (!reduce) return PL_OK;
  const int n = K * H;
  if (out_transposed)
    hipLaunchKernelGGL((skinny_reduce_kernel<true>), dim3((n + 15) / 16), block, 0, s, part, nc, K, H, out);
  else
    hipLaunchKernelGGL((skinny_reduce_kernel<false>), dim3((n + 15) / 16), block, 0, s, part, nc, K, H, out);
  PL_CHECK_LAUNCH("skinny_reduce");
  return PL_OK;
}

int skinny_narrow_out_splits(int B, int H) {
  int s = 8;
  while (s > 1 && (H % (8 * s) != 0)) s >>= 1;
  return s;
}
bool skinny_narrow_out_supported(int H, int N) { return N <= 64 && H % 8 == 0; }
size_t skinny_narrow_out_part_floats(int B, int H) { return (size_t)skinny_narrow_out_splits(B, H) * B * 64; }

int launch_skinny_narrow_out(const float* h, const float* W, const float* bias, float* y, int B, int H,
                             int N, float* part, hipStream_t s, bool reduce) {
  if (!skinny_narrow_out_supported(H, N)) PL_FAIL(PL_ESHAPE, "skinny_narrow_out: H=%d N=%d", H, N);
  const int splits = skinny_narrow_out_splits(B, H);
  const int tasks = ((B + 31) / 32) * splits;
  hipLaunchKernelGGL(skinny_narrow_out_kernel, dim3((tasks + 3) / 4), dim3(NTHR), 0, s, h, W, B, H, N, splits,
                     part);
  PL_CHECK_LAUNCH("skinny_narrow_out");
  if (!reduce) return PL_OK;                  // the caller combines the slabs (mse_partial_from_slabs)
  const int64_t n = (int64_t)B * 64;
  hipLaunchKernelGGL(skinny_narrow_out_reduce_kernel, dim3((unsigned)((n + NTHR - 1) / NTHR)), dim3(NTHR), 0, s,
                     part, splits, B, N, bias, y);
  PL_CHECK_LAUNCH("skinny_narrow_out_reduce");
  return PL_OK;
}

}  // namespace pl
