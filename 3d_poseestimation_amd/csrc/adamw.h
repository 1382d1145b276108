// The flat AdamW step's device pieces, shared by the optimizer kernel (elementwise.hip) and the small-batch backward launches
// that carry a slice of the step (small_layer.hip): torch.optim.AdamW's single-tensor update order
// (phase1_lifting/train_1.py:39 optimizer, :89 step).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/poselift.h"

namespace pl {

struct AdamWK {
  float decay;       // 1 - lr*wd
  float one_m_b1, b2, one_m_b2;
  float step_size;   // lr / (1 - b1^t)
  float bc2_sqrt;    // sqrt(1 - b2^t)
  float eps, gscale;
};

// What the step is given; the per-step constants are derived ON THE DEVICE (every thread, once: two double pow),
// so that a captured graph advances them by itself: t = t_base + *t_dev, lr = *lr_dev when the pointers are set
// (the eager call passes them by value through the same code, hence the same bits).
struct AdamWIn {
  float lr, beta1, beta2, eps, wd, gscale;
  int64_t t;
  const float* lr_dev;
  const uint64_t* t_dev;
  // GEMM operand planes of the 1024-wide weight matrices, refreshed by the step that changes them (PLAdamWPlanes):
  // the forward then needs no split pass of its own
  int nseg, kind;
  float pscale;
  int64_t seg_off[PL_ADAMW_MAX_SEGS], seg_n[PL_ADAMW_MAX_SEGS];
  unsigned short* seg_h[PL_ADAMW_MAX_SEGS];
  unsigned short* seg_l[PL_ADAMW_MAX_SEGS];
};

// a slice [p, p + n) of the step carried by another kernel's spare workgroups (small_layer.hip): same fields, same constants
struct AdamWRide {
  float *p, *m, *v;
  const float* g;
  int64_t n;                 // floats; n % 4 == 0, the four pointers 16-byte aligned
  float lr, beta1, beta2, eps, wd, gscale;
  int64_t t;
  const float* lr_dev;
  const uint64_t* t_dev;
};

template <class A>
__device__ __forceinline__ AdamWK adamw_consts(const A& a) {
  const double lr = a.lr_dev ? (double)a.lr_dev[0] : (double)a.lr;
  const double t = (double)(a.t + (a.t_dev ? (int64_t)a.t_dev[0] : 0));
  const double bc1 = 1.0 - pow((double)a.beta1, t);
  const double bc2 = 1.0 - pow((double)a.beta2, t);
  AdamWK k;
  k.decay = (float)(1.0 - lr * (double)a.wd);
  k.one_m_b1 = (float)(1.0 - (double)a.beta1);
  k.b2 = a.beta2;
  k.one_m_b2 = (float)(1.0 - (double)a.beta2);
  k.step_size = (float)(lr / bc1);
  k.bc2_sqrt = (float)sqrt(bc2);
  k.eps = a.eps;
  k.gscale = a.gscale;
  return k;
}

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, const AdamWK& k) {
  g *= k.gscale;
  p *= k.decay;
  m = m + (g - m) * k.one_m_b1;
  v = v * k.b2 + (k.one_m_b2 * g) * g;
  const float denom = sqrtf(v) / k.bc2_sqrt + k.eps;
  p = p - k.step_size * (m / denom);
}


// p, m, v (and g read) of elements [4 i0, n) in float4 steps of `stride` threads: the loop of adamw_kernel without the planes
__device__ __forceinline__ void adamw_span(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                           float* __restrict__ v, int64_t n4, int64_t i0, int64_t stride, const AdamWK& k) {
  for (int64_t i = i0; i < n4; i += stride) {
    float4 pv = *reinterpret_cast<const float4*>(p + 4 * i), mv = *reinterpret_cast<const float4*>(m + 4 * i);
    float4 vv = *reinterpret_cast<const float4*>(v + 4 * i);
    const float4 gv = *reinterpret_cast<const float4*>(g + 4 * i);
    adamw_one(pv.x, gv.x, mv.x, vv.x, k);
    adamw_one(pv.y, gv.y, mv.y, vv.y, k);
    adamw_one(pv.z, gv.z, mv.z, vv.z, k);
    adamw_one(pv.w, gv.w, mv.w, vv.w, k);
    *reinterpret_cast<float4*>(p + 4 * i) = pv;
    *reinterpret_cast<float4*>(m + 4 * i) = mv;
    *reinterpret_cast<float4*>(v + 4 * i) = vv;
  }
}

}  // namespace pl
