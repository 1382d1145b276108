// Thin GEMMs: the 1024-wide Linears of the lifter at SMALL batch (ragged M <= 512 rows: the reference's own batch_size = 64,
// phase1_lifting/train_1.py:194, and the ragged last batch of its DataLoader, :51).
//
// At M = 64 the 128x128-tile kernels have 8 workgroups for 256 CUs and walk K = 1024 serially (gemm_f32.hip's guarded
// edge kernel: 155-160 us per GEMM, 1.26 ms of a 1.51 ms train step at B = 64).  The work itself is nothing -- 0.13 GFLOP
// and one pass over W (4 MB) -- so here the contraction is split over the chip instead: a wavefront owns a 64-row x
// 32-column x (K / splits) piece, feeds exact-fp32 MFMAs (v_mfma_f32_32x32x2_f32: the edge kernel's arithmetic, so small
// batches keep fp32 products) straight from registers, and writes one slab; a second small kernel adds the slabs in a
// fixed order and applies the epilogue of gemm_epilogue.h (bias, skip-gradient addend, eval-BN fold, ReLU, residual,
// training-mode BatchNorm partial statistics per 64-row group).  Replaces the same ATen addmm calls as gemm_f32.hip
// (reference phase1_lifting/baselineModel.py:33,39 and their autograd).
#include <stdlib.h>

#include "pl_internal.h"

namespace pl {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NTHR = 256;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// part[z][M][N] = A[:, slice z] B[slice z, :]
//   B_KS = false (NT): B is [N][K], k-contiguous -- a lane fetches 4 consecutive k of its row of A and of its row of B with
//          one 16-byte load each and feeds 4 MFMAs (any k order shared by both operands is a valid contraction order);
//   B_KS = true  (NN): B is [K][N] -- the same 4 k of A against 4 rows of B, 128-byte segments along n.
// Rows >= M / columns >= N read a clamped (valid) row and compute values that are never stored.
template <bool B_KS>
__global__ __launch_bounds__(NTHR) void thin_gemm_kernel(GemmArgs p, int splits, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int strips = (p.N + 31) >> 5, rbs = (p.M + 63) >> 6;
  const int task = blockIdx.x * 4 + wave;
  const int z = task % splits, t2 = task / splits;
  const int cs = t2 % strips, rb = t2 / strips;
  if (rb >= rbs) return;
  const int kper = p.K / splits, k0 = z * kper;
  const int r0 = min(rb * 64 + j, p.M - 1), r1 = min(rb * 64 + 32 + j, p.M - 1);
  const int col = min(cs * 32 + j, p.N - 1);
  const float* __restrict__ ap0 = p.A + (size_t)r0 * p.lda + k0 + 4 * h;
  const float* __restrict__ ap1 = p.A + (size_t)r1 * p.lda + k0 + 4 * h;
  const float* __restrict__ bp = B_KS ? p.B + (size_t)(k0 + 4 * h) * p.ldb + col : p.B + (size_t)col * p.ldb + k0 + 4 * h;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#define PL_THIN_MFMA4(A0, A1, BV)                                                       \
  do {                                                                                  \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((A0).x, (BV).x, acc[0], 0, 0, 0);     \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((A1).x, (BV).x, acc[1], 0, 0, 0);     \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((A0).y, (BV).y, acc[0], 0, 0, 0);     \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((A1).y, (BV).y, acc[1], 0, 0, 0);     \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((A0).z, (BV).z, acc[0], 0, 0, 0);     \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((A1).z, (BV).z, acc[1], 0, 0, 0);     \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((A0).w, (BV).w, acc[0], 0, 0, 0);     \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((A1).w, (BV).w, acc[1], 0, 0, 0);     \
  } while (0)
  auto load_b = [&](int kb) {
    if (!B_KS) return ld4(bp + kb);
    const float* q = bp + (size_t)kb * p.ldb;
    return make_float4(q[0], q[p.ldb], q[2 * (size_t)p.ldb], q[3 * (size_t)p.ldb]);
  };
  int kb = 0;
  // batches of 4 steps (32 k): every load of the batch is issued before its MFMAs
  for (; kb + 32 <= kper; kb += 32) {
    float4 a0[4], a1[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a0[q] = ld4(ap0 + kb + 8 * q);
      a1[q] = ld4(ap1 + kb + 8 * q);
      b[q] = load_b(kb + 8 * q);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) PL_THIN_MFMA4(a0[q], a1[q], b[q]);
  }
  for (; kb < kper; kb += 8) {
    const float4 a0 = ld4(ap0 + kb), a1 = ld4(ap1 + kb), b = load_b(kb);
    PL_THIN_MFMA4(a0, a1, b);
  }
#undef PL_THIN_MFMA4
  const int oc = cs * 32 + j;
  if (oc < p.N) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb * 64 + t * 32 + acc_row(r, h);
        if (row < p.M) part[((size_t)z * p.M + row) * p.N + oc] = acc[t][r];
      }
  }
}

// C = epilogue(sum_z part[z]) on a 64-row statistics group x 16-column strip per workgroup (N / 16 workgroups per group:
// the slabs are the whole traffic of this kernel and want every CU pulling).  Lane: float4 column lane & 3, row
// 16 wave + (lane >> 2); its SPLITS slab loads are all in flight together and are added in slab order.
// Element order as gemm_epilogue: bias, addend, eval-BN fold, ReLU, residual, ReLU; statistics on the final values.
template <int SPLITS>
__global__ __launch_bounds__(NTHR) void thin_reduce_kernel(GemmArgs p, const float* __restrict__ part) {
  __shared__ float4 sm[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cq = lane & 3, lr = wave * 16 + (lane >> 2);
  const int c = blockIdx.x * 16 + cq * 4;
  const int g0 = blockIdx.y * 64;
  const int cnt = max(0, min(64, p.M - g0));       // (statistics groups past the last row: zeros, as the tile kernels write)
  const bool live = c < p.N && lr < cnt;
  const size_t MN = (size_t)p.M * p.N;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const int r = g0 + lr;
    const float* q = part + (size_t)r * p.N + c;
    float4 u[SPLITS];
#pragma unroll
    for (int z = 0; z < SPLITS; ++z) u[z] = ld4(q + (size_t)z * MN);
    a = u[0];
#pragma unroll
    for (int z = 1; z < SPLITS; ++z) { a.x += u[z].x; a.y += u[z].y; a.z += u[z].z; a.w += u[z].w; }
    const size_t o = (size_t)r * p.ldc + c;
    if (p.bias) { const float4 b = ld4(p.bias + c); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    if (p.addend) { const float4 t = ld4(p.addend + o); a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
    if (p.col_scale) {
      const float4 sc = ld4(p.col_scale + c), sh = ld4(p.col_shift + c);
      a.x = fmaf(a.x, sc.x, sh.x); a.y = fmaf(a.y, sc.y, sh.y); a.z = fmaf(a.z, sc.z, sh.z); a.w = fmaf(a.w, sc.w, sh.w);
    }
    if (p.relu == 1) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    if (p.resid) { const float4 t = ld4(p.resid + o); a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
    if (p.relu == 2) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    st4(p.C + o, a);
  }
  if (!p.stat_sum) return;                     // (kernel-uniform)
  // column sums over the group's rows: the wave's 16 rows by butterfly (lanes l ^ 4, 8, 16, 32 share a column), the four
  // waves through LDS in wave order
  auto colsum = [&](float4 v) {
#pragma unroll
    for (int o = 4; o <= 32; o <<= 1) {
      v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); v.z += __shfl_xor(v.z, o); v.w += __shfl_xor(v.w, o);
    }
    if (lane < 4) sm[wave][lane] = v;
    __syncthreads();
    float4 t = sm[0][cq];
#pragma unroll
    for (int w = 1; w < 4; ++w) { const float4 x = sm[w][cq]; t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w; }
    __syncthreads();
    return t;
  };
  const float4 tot = colsum(a);
  const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
  float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const float dx = a.x - tot.x * inv, dy = a.y - tot.y * inv, dz = a.z - tot.z * inv, dw = a.w - tot.w * inv;
    d = make_float4(dx * dx, dy * dy, dz * dz, dw * dw);
  }
  const float4 m2 = colsum(d);
  if (threadIdx.x < 4 && c < p.N) {
    const size_t o = (size_t)blockIdx.y * p.N + c;
    st4(p.stat_sum + o, tot);
    st4(p.stat_m2 + o, m2);
  }
}

}  // namespace

// largest M taken (POSELIFT_THIN_MAX_M overrides: same-box sweeps)
int thin_gemm_max_m() {
  static const int m = [] { const char* e = getenv("POSELIFT_THIN_MAX_M"); return e ? atoi(e) : 512; }();
  return m;
}

// K slices: a function of (N, K) only -- the contraction order, hence every output bit, is the same for every batch size
// that takes this path.  Enough wave tasks at ONE 64-row block for half the chip's SIMDs, slices of >= 32 k, <= 16 slabs.
int thin_gemm_splits(int N, int K) {
  const int strips = (N + 31) / 32;
  int s = 1;
  while (s < 16 && strips * s < 512 && K % (8 * 2 * s) == 0 && K / (2 * s) >= 32) s *= 2;
  return s;
}

size_t thin_gemm_scratch_floats(int M, int N, int K) { return (size_t)thin_gemm_splits(N, K) * M * N; }

bool thin_gemm_ok(GemmLayout layout, const GemmArgs& a) {
  if (layout != kNT && layout != kNN) return false;
  if (a.split_k > 1 || a.conv_cin || a.bnr_z) return false;
  if (a.M < 1 || a.M > thin_gemm_max_m() || a.N < 32 || (a.N & 3) || a.K < 8 || (a.K & 7)) return false;
  if ((a.lda & 3) || (a.ldc & 3) || (layout == kNT && (a.ldb & 3))) return false;
  if ((a.stat_sum != nullptr) != (a.stat_m2 != nullptr)) return false;
  if ((a.col_scale != nullptr) != (a.col_shift != nullptr)) return false;
  uintptr_t al = reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.C) | reinterpret_cast<uintptr_t>(a.bias) |
                 reinterpret_cast<uintptr_t>(a.addend) | reinterpret_cast<uintptr_t>(a.resid) |
                 reinterpret_cast<uintptr_t>(a.col_scale) | reinterpret_cast<uintptr_t>(a.col_shift) |
                 reinterpret_cast<uintptr_t>(a.stat_sum) | reinterpret_cast<uintptr_t>(a.stat_m2);
  if (layout == kNT) al |= reinterpret_cast<uintptr_t>(a.B);
  return (al & 15) == 0;
}

int launch_gemm_thin(GemmLayout layout, const GemmArgs& a, float* scratch, size_t scratch_floats, hipStream_t s) {
  if (!thin_gemm_ok(layout, a)) PL_FAIL(PL_ESHAPE, "gemm_thin: unsupported problem %dx%dx%d (layout %d)", a.M, a.N, a.K, (int)layout);
  const int splits = thin_gemm_splits(a.N, a.K);
  if (!scratch || scratch_floats < (size_t)splits * a.M * a.N || (reinterpret_cast<uintptr_t>(scratch) & 15))
    PL_FAIL(PL_EWORKSPACE, "gemm_thin: scratch of %zu floats needed", (size_t)splits * a.M * a.N);
  const int tasks = ((a.M + 63) / 64) * ((a.N + 31) / 32) * splits;
  void* prof = prof_begin_flops(2.0 * a.M * a.N * a.K, s);
  if (layout == kNT) hipLaunchKernelGGL(thin_gemm_kernel<false>, dim3((tasks + 3) / 4), dim3(NTHR), 0, s, a, splits, scratch);
  else hipLaunchKernelGGL(thin_gemm_kernel<true>, dim3((tasks + 3) / 4), dim3(NTHR), 0, s, a, splits, scratch);
  PL_CHECK_LAUNCH("gemm_thin");
  const dim3 rgrid((a.N + 15) / 16, a.stat_sum ? gemm_stat_groups(a.M) : (a.M + 63) / 64);
  switch (splits) {
    case 1: hipLaunchKernelGGL(thin_reduce_kernel<1>, rgrid, dim3(NTHR), 0, s, a, scratch); break;
    case 2: hipLaunchKernelGGL(thin_reduce_kernel<2>, rgrid, dim3(NTHR), 0, s, a, scratch); break;
    case 4: hipLaunchKernelGGL(thin_reduce_kernel<4>, rgrid, dim3(NTHR), 0, s, a, scratch); break;
    case 8: hipLaunchKernelGGL(thin_reduce_kernel<8>, rgrid, dim3(NTHR), 0, s, a, scratch); break;
    default: hipLaunchKernelGGL(thin_reduce_kernel<16>, rgrid, dim3(NTHR), 0, s, a, scratch); break;
  }
  prof_end(prof, s);
  PL_CHECK_LAUNCH("gemm_thin_reduce");
  return PL_OK;
}

}  // namespace pl
