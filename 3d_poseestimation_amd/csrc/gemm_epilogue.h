// Register epilogue shared by the fp32-operand GEMM (gemm_f32.hip) and the planes GEMM (gemm_planes.hip): same
// accumulator layout (4 wavefronts 2x2, 64x64 each as 2x2 MFMA tiles of 32x32), same bias / eval-BN fold / ReLU /
// residual / training-mode BatchNorm partial statistics, so the two main loops are interchangeable bit for bit.
#pragma once
#include "pl_internal.h"

namespace pl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- epilogue (shared by every main-loop variant) ------------------------------------------
// (Tried: transposing each wave's 64x64 block through LDS and storing 16 x dwordx4 instead of 64 x dword
//  per lane -- same-box A/B: 54.2 vs 53.3 us per forward GEMM, i.e. the extra barrier and LDS round trip
//  cost more than the narrower store issue saves at this size.  Not kept.)
// acc[a][b][r]: row = m0 + wm*64 + a*32 + (r&3) + 8*(r>>2) + 4*h, col = n0 + wn*32*NB + b*32 + i
template <bool EDGE, int NB>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, float* __restrict__ C, f32x16 (&acc)[2][NB],
                                              const int m0, const int n0, const int wm, const int wn,
                                              const int i, const int h) {
  constexpr int WCOLS = 32 * NB;
  const int rbase = m0 + wm * 64 + 4 * h;
  const bool plain = p.split_k > 1;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int col = n0 + wn * WCOLS + b * 32 + i;
    const bool cok = !EDGE || col < p.N;
    float bias = 0.f, scale = 1.f, shift = 0.f;
    if (!plain && cok) {
      if (p.bias) bias = p.bias[col];
      if (p.col_scale) { scale = p.col_scale[col]; shift = p.col_shift[col]; }
    }
    float ssum = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + a * 32 + (r & 3) + 8 * (r >> 2);
        float v = acc[a][b][r];
        if (!plain) {
          v += bias;
          if (p.addend && cok && (!EDGE || row < p.M)) v += p.addend[(size_t)row * p.ldc + col];
          if (p.col_scale) v = fmaf(v, scale, shift);
          if (p.relu == 1) v = fmaxf(v, 0.f);
          if (p.resid && cok && (!EDGE || row < p.M)) v += p.resid[(size_t)row * p.ldc + col];
          if (p.relu == 2) v = fmaxf(v, 0.f);
        }
        acc[a][b][r] = v;
        if (!EDGE || row < p.M) ssum += v;
      }
    if (!plain && p.stat_sum) {
      // column statistics over this wavefront's 64 rows (both lane halves)
      ssum += __shfl_xor(ssum, 32);
      const int g0 = m0 + wm * 64;
      const int cnt = max(0, min(64, p.M - g0));
      const float mean = cnt > 0 ? ssum / (float)cnt : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + a * 32 + (r & 3) + 8 * (r >> 2);
          const float d = acc[a][b][r] - mean;
          if (!EDGE || row < p.M) m2 = fmaf(d, d, m2);
        }
      m2 += __shfl_xor(m2, 32);
      if (h == 0 && cok) {
        const size_t o = (size_t)(g0 >> 6) * p.N + col;
        p.stat_sum[o] = ssum;
        p.stat_m2[o] = m2;
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + a * 32 + (r & 3) + 8 * (r >> 2);
        if (cok && (!EDGE || row < p.M)) C[(size_t)row * p.ldc + col] = acc[a][b][r];
      }
  }
}

}  // namespace pl
