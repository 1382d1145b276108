// Writers / readers of GEMM operand planes for the streaming kernels (PlaneOut, pl_internal.h): the kernel that produces a
// tensor stores it as the planes the planes GEMM stages by LDS-DMA.
#pragma once
#include "pl_internal.h"

namespace pl {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
struct PlaneDst { unsigned short* h; unsigned short* l; float scale; int kind; int nt; };

__device__ __forceinline__ PlaneDst plane_dst(const PlaneOut& o) {
  PlaneDst d = {o.h, o.l, o.scale, o.kind, o.nt};
  if (o.kind == 2 && o.dyn) d.scale = o.dyn[0];
  return d;
}

__device__ __forceinline__ void store_planes4(const PlaneDst& d, size_t off, float4 v) {
  if (d.kind == 2) {
    const float a[4] = {v.x * d.scale, v.y * d.scale, v.z * d.scale, v.w * d.scale};
    f16x4 hh, ll;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hh[j] = (_Float16)a[j];
      ll[j] = (_Float16)((a[j] - (float)hh[j]) * 2048.0f);
    }
    if (d.nt) {
      __builtin_nontemporal_store(hh, reinterpret_cast<f16x4*>(d.h + off));
      __builtin_nontemporal_store(ll, reinterpret_cast<f16x4*>(d.l + off));
    } else {
      *reinterpret_cast<f16x4*>(d.h + off) = hh;
      *reinterpret_cast<f16x4*>(d.l + off) = ll;
    }
  } else if (d.kind == 1) {
    bf16x4 q;
    q[0] = (__bf16)v.x; q[1] = (__bf16)v.y; q[2] = (__bf16)v.z; q[3] = (__bf16)v.w;
    if (d.nt) __builtin_nontemporal_store(q, reinterpret_cast<bf16x4*>(d.h + off));
    else *reinterpret_cast<bf16x4*>(d.h + off) = q;
  }
}

typedef float f32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_nt(float* p, float4 v, int nt) {
  if (nt) {
    const f32x4s q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, reinterpret_cast<f32x4s*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}

// x[off .. off+3] back from its planes, times inv (1 / S): kind 2 -> (h + l / 2048) * inv, kind 1 -> the bf16 values
__device__ __forceinline__ float4 load_planes4(int kind, const unsigned short* h, const unsigned short* l, size_t off, float inv) {
  if (kind == 2) {
    const f16x4 hh = *reinterpret_cast<const f16x4*>(h + off), ll = *reinterpret_cast<const f16x4*>(l + off);
    return make_float4(fmaf((float)ll[0], 1.0f / 2048.0f, (float)hh[0]) * inv, fmaf((float)ll[1], 1.0f / 2048.0f, (float)hh[1]) * inv,
                       fmaf((float)ll[2], 1.0f / 2048.0f, (float)hh[2]) * inv, fmaf((float)ll[3], 1.0f / 2048.0f, (float)hh[3]) * inv);
  }
  const bf16x4 q = *reinterpret_cast<const bf16x4*>(h + off);
  return make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
}

}  // namespace pl
