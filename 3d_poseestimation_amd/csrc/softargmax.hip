// Fused softmax + integral soft-argmax.
//
// Replaces the tail of Model_3D.forward (phase4_joined/Model.py:94-133: norm_heatmap softmax over
// D*H*W voxels per joint, the no-op renormalisation, three marginal sums, arange expectation,
// (c/dim - 0.5)*2) and of Model_2D.forward (phase5_loop/Model_2d.py:96-134: depth 1, c/dim).
// The reference materialises the normalised (B,17,64,64,64) heat-map 3-4 times (17.8 MB per
// frame each); here one pass streams the logits once with an online softmax that carries the
// three coordinate expectations along, and the backward streams them once more:
//   forward : read logits                      -> coords (B,J,3|2) + per-(b,j) {max, sum, Ex, Ey, Ez}
//   backward: read logits, write dlogits       dl = p * sum_a g_a * c_a * (idx_a - E_a)
// HBM-bound: 4 B read per voxel forward, 4 B read + 4 B written backward.
#include "pl_internal.h"
#include "plane_store.h"

namespace pl {
namespace {

constexpr int NTHR = 256;

struct Acc { float m, s, x, y, z; };

__device__ __forceinline__ void merge(Acc& a, const Acc& b) {
  const float m = fmaxf(a.m, b.m);
  const float fa = (a.m == -INFINITY) ? 0.f : __expf(a.m - m);
  const float fb = (b.m == -INFINITY) ? 0.f : __expf(b.m - m);
  a.s = a.s * fa + b.s * fb;
  a.x = a.x * fa + b.x * fb;
  a.y = a.y * fa + b.y * fb;
  a.z = a.z * fa + b.z * fb;
  a.m = m;
}

// one workgroup per (batch, joint) heat-map of n = D*H*W logits; W % 4 == 0
__global__ __launch_bounds__(NTHR) void softargmax_fwd_kernel(const float* __restrict__ logits, int D,
                                                              int H, int W, int centred,
                                                              float* __restrict__ coords, int ncoord,
                                                              float* __restrict__ stats) {
  __shared__ Acc sm[NTHR / 64];
  const int n4 = (D * H * W) >> 2;
  const float4* __restrict__ src = reinterpret_cast<const float4*>(logits + (size_t)blockIdx.x * D * H * W);
  Acc a = {-INFINITY, 0.f, 0.f, 0.f, 0.f};
  const int w4 = W >> 2;
  for (int i = threadIdx.x; i < n4; i += NTHR) {
    const float4 v = src[i];
    const int wq = i % w4, hd = i / w4;
    const float w0 = (float)(wq * 4), hh = (float)(hd % H), dd = (float)(hd / H);
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    if (mx > a.m) {                      // rescale what this lane has so far
      const float f = (a.m == -INFINITY) ? 0.f : __expf(a.m - mx);
      a.s *= f; a.x *= f; a.y *= f; a.z *= f;
      a.m = mx;
    }
    const float e0 = __expf(v.x - a.m), e1 = __expf(v.y - a.m), e2 = __expf(v.z - a.m), e3 = __expf(v.w - a.m);
    const float es = (e0 + e1) + (e2 + e3);
    a.s += es;
    a.x += e0 * w0 + e1 * (w0 + 1.f) + e2 * (w0 + 2.f) + e3 * (w0 + 3.f);
    a.y = fmaf(es, hh, a.y);
    a.z = fmaf(es, dd, a.z);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Acc b;
    b.m = __shfl_xor(a.m, o); b.s = __shfl_xor(a.s, o); b.x = __shfl_xor(a.x, o);
    b.y = __shfl_xor(a.y, o); b.z = __shfl_xor(a.z, o);
    merge(a, b);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sm[wave] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    Acc t = sm[0];
    for (int w = 1; w < NTHR / 64; ++w) merge(t, sm[w]);
    const float inv = 1.0f / t.s;
    const float ex = t.x * inv, ey = t.y * inv, ez = t.z * inv;
    float* st = stats + (size_t)blockIdx.x * 5;
    st[0] = t.m; st[1] = t.s; st[2] = ex; st[3] = ey; st[4] = ez;
    float* c = coords + (size_t)blockIdx.x * ncoord;
    if (centred) {
      c[0] = (ex / (float)W - 0.5f) * 2.f;
      c[1] = (ey / (float)H - 0.5f) * 2.f;
      if (ncoord > 2) c[2] = (ez / (float)D - 0.5f) * 2.f;
    } else {
      c[0] = ex / (float)W;
      c[1] = ey / (float)H;
      if (ncoord > 2) c[2] = ez / (float)D;
    }
  }
}

// The same forward on NHWC logits [B][H*W][J*64] (depth 64 = one lane per depth slice): what the conv
// path's final 1x1 convolution writes, so Model_3D inference needs no NHWC -> NCHW pass (1.1 GB read + written
// at B = 64).  One workgroup per (batch, joint); a wavefront reads one pixel's 64 depths (256 B) per load.
__global__ __launch_bounds__(NTHR) void softargmax_nhwc_fwd_kernel(const float* __restrict__ logits, int J, int H,
                                                                   int W, float* __restrict__ coords,
                                                                   float* __restrict__ stats) {
  __shared__ Acc sm[NTHR / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / J, j = blockIdx.x - b * J;
  const int P = H * W, C = J * 64;
  const float* __restrict__ src = logits + (size_t)b * P * C + j * 64 + lane;
  Acc a = {-INFINITY, 0.f, 0.f, 0.f, 0.f};
  for (int p0 = wave; p0 < P; p0 += 4 * (NTHR / 64)) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = p0 + q * (NTHR / 64);
      v[q] = p < P ? src[(size_t)p * C] : -INFINITY;
    }
    const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    if (mx > a.m) {
      const float f = (a.m == -INFINITY) ? 0.f : __expf(a.m - mx);
      a.s *= f; a.x *= f; a.y *= f; a.z *= f;
      a.m = mx;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = p0 + q * (NTHR / 64);
      if (p < P) {
        const float e = __expf(v[q] - a.m);
        a.s += e;
        a.x = fmaf(e, (float)(p % W), a.x);
        a.y = fmaf(e, (float)(p / W), a.y);
      }
    }
  }
  a.z = a.s * (float)lane;                   // this lane's depth index, weight = its whole mass
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Acc t;
    t.m = __shfl_xor(a.m, o); t.s = __shfl_xor(a.s, o); t.x = __shfl_xor(a.x, o);
    t.y = __shfl_xor(a.y, o); t.z = __shfl_xor(a.z, o);
    merge(a, t);
  }
  if (lane == 0) sm[wave] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    Acc t = sm[0];
    for (int w = 1; w < NTHR / 64; ++w) merge(t, sm[w]);
    const float inv = 1.0f / t.s;
    const float ex = t.x * inv, ey = t.y * inv, ez = t.z * inv;
    float* st = stats + (size_t)blockIdx.x * 5;
    st[0] = t.m; st[1] = t.s; st[2] = ex; st[3] = ey; st[4] = ez;
    float* c = coords + (size_t)blockIdx.x * 3;
    c[0] = (ex / (float)W - 0.5f) * 2.f;
    c[1] = (ey / (float)H - 0.5f) * 2.f;
    c[2] = (ez / 64.f - 0.5f) * 2.f;
  }
}

// grid = (chunks, BJ): dlogit = p * (gx*cx*(w-Ex) + gy*cy*(h-Ey) + gz*cz*(d-Ez))
__global__ __launch_bounds__(NTHR) void softargmax_bwd_kernel(const float* __restrict__ logits,
                                                              const float* __restrict__ stats,
                                                              const float* __restrict__ gcoords, int D, int H,
                                                              int W, int centred, int ncoord,
                                                              float* __restrict__ dlogits) {
  const int bj = blockIdx.y;
  const int n4 = (D * H * W) >> 2;
  const float* st = stats + (size_t)bj * 5;
  const float m = st[0], inv = 1.0f / st[1], ex = st[2], ey = st[3], ez = st[4];
  const float k = centred ? 2.f : 1.f;
  const float* g = gcoords + (size_t)bj * ncoord;
  const float gx = g[0] * k / (float)W, gy = g[1] * k / (float)H, gz = ncoord > 2 ? g[2] * k / (float)D : 0.f;
  const float4* __restrict__ src = reinterpret_cast<const float4*>(logits + (size_t)bj * D * H * W);
  float4* __restrict__ dst = reinterpret_cast<float4*>(dlogits + (size_t)bj * D * H * W);
  const int w4 = W >> 2;
  for (int i = blockIdx.x * NTHR + threadIdx.x; i < n4; i += gridDim.x * NTHR) {
    const float4 v = src[i];
    const int wq = i % w4, hd = i / w4;
    const float w0 = (float)(wq * 4) - ex;
    const float base = gy * ((float)(hd % H) - ey) + gz * ((float)(hd / H) - ez);
    float4 o;
    o.x = __expf(v.x - m) * inv * fmaf(gx, w0, base);
    o.y = __expf(v.y - m) * inv * fmaf(gx, w0 + 1.f, base);
    o.z = __expf(v.z - m) * inv * fmaf(gx, w0 + 2.f, base);
    o.w = __expf(v.w - m) * inv * fmaf(gx, w0 + 3.f, base);
    dst[i] = o;
  }
}

// backward on the NHWC layout: one float4 = four consecutive depths of one (pixel, joint); pure streaming
// (4 B read + 4 B written per voxel, every access a whole 16-byte vector of a contiguous row)
__global__ __launch_bounds__(NTHR) void softargmax_nhwc_bwd_kernel(const float* __restrict__ logits,
                                                                   const float* __restrict__ stats,
                                                                   const float* __restrict__ gcoords, int J, int H,
                                                                   int W, int64_t n4, float* __restrict__ dlogits,
                                                                   PlaneOut po) {
  const int64_t t = (int64_t)blockIdx.x * NTHR + threadIdx.x;
  if (t >= n4) return;
  const PlaneDst pd = plane_dst(po);
  const int c4 = J * 16;                           // float4s per pixel
  const int q = (int)(t % c4);
  const int64_t bp = t / c4;
  const int P = H * W;
  const int p = (int)(bp % P);
  const int64_t b = bp / P;
  const int j = q >> 4, d0 = (q & 15) * 4;
  const float* st = stats + ((size_t)b * J + j) * 5;
  const float* g = gcoords + ((size_t)b * J + j) * 3;
  const float m = st[0], inv = 1.0f / st[1];
  const float gx = g[0] * 2.f / (float)W, gy = g[1] * 2.f / (float)H, gz = g[2] * 2.f / 64.f;
  const float base = gx * ((float)(p % W) - st[2]) + gy * ((float)(p / W) - st[3]);
  const float z0 = (float)d0 - st[4];
  const float4 v = reinterpret_cast<const float4*>(logits)[t];
  float4 o;
  o.x = __expf(v.x - m) * inv * fmaf(gz, z0, base);
  o.y = __expf(v.y - m) * inv * fmaf(gz, z0 + 1.f, base);
  o.z = __expf(v.z - m) * inv * fmaf(gz, z0 + 2.f, base);
  o.w = __expf(v.w - m) * inv * fmaf(gz, z0 + 3.f, base);
  if (dlogits) reinterpret_cast<float4*>(dlogits)[t] = o;
  if (pd.kind) store_planes4(pd, (size_t)t * 4, o);        // the final convolution's gradient GEMMs read planes
}

int check_dims(int64_t BJ, int64_t D, int64_t H, int64_t W, int ncoord, const char* who) {
  if (BJ <= 0 || D <= 0 || H <= 0 || W <= 0 || (W & 3) || D * H * W > (int64_t)1 << 30 || BJ > 65535 * 64)
    PL_FAIL(PL_ESHAPE, "%s: bad dims BJ=%lld D=%lld H=%lld W=%lld (W %% 4 == 0)", who, (long long)BJ, (long long)D,
            (long long)H, (long long)W);
  if (ncoord != 2 && ncoord != 3) PL_FAIL(PL_ESHAPE, "%s: ncoord=%d", who, ncoord);
  if (ncoord == 2 && D != 1) PL_FAIL(PL_ESHAPE, "%s: 2 coordinates need depth 1", who);
  return PL_OK;
}

}  // namespace
}  // namespace pl

using namespace pl;

extern "C" int pl_softargmax_fwd(const float* logits, int64_t BJ, int64_t D, int64_t H, int64_t W, int ncoord,
                                 int centred, float* coords, float* stats, void* stream) {
  if (!logits || !coords || !stats) PL_FAIL(PL_EINVAL, "pl_softargmax_fwd: null pointer");
  if (reinterpret_cast<uintptr_t>(logits) & 15) PL_FAIL(PL_EINVAL, "pl_softargmax_fwd: logits not 16-byte aligned");
  PL_TRY(check_dims(BJ, D, H, W, ncoord, "pl_softargmax_fwd"));
  hipLaunchKernelGGL(softargmax_fwd_kernel, dim3((unsigned)BJ), dim3(NTHR), 0, (hipStream_t)stream, logits, (int)D,
                     (int)H, (int)W, centred, coords, ncoord, stats);
  PL_CHECK_LAUNCH("softargmax_fwd");
  return PL_OK;
}

extern "C" int pl_softargmax_bwd(const float* logits, const float* stats, const float* gcoords, int64_t BJ,
                                 int64_t D, int64_t H, int64_t W, int ncoord, int centred, float* dlogits,
                                 void* stream) {
  if (!logits || !stats || !gcoords || !dlogits) PL_FAIL(PL_EINVAL, "pl_softargmax_bwd: null pointer");
  if ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15)
    PL_FAIL(PL_EINVAL, "pl_softargmax_bwd: tensors not 16-byte aligned");
  PL_TRY(check_dims(BJ, D, H, W, ncoord, "pl_softargmax_bwd"));
  if (BJ > 65535) PL_FAIL(PL_ESHAPE, "pl_softargmax_bwd: BJ=%lld > 65535 (split the batch)", (long long)BJ);
  const int64_t n4 = (D * H * W) >> 2;
  int chunks = (int)((n4 + NTHR * 8 - 1) / (NTHR * 8));      // 8 float4 per thread
  if (chunks < 1) chunks = 1;
  if (chunks > 64) chunks = 64;
  hipLaunchKernelGGL(softargmax_bwd_kernel, dim3(chunks, (unsigned)BJ), dim3(NTHR), 0, (hipStream_t)stream, logits,
                     stats, gcoords, (int)D, (int)H, (int)W, centred, ncoord, dlogits);
  PL_CHECK_LAUNCH("softargmax_bwd");
  return PL_OK;
}

extern "C" int pl_softargmax3d_nhwc_fwd(const float* logits, int64_t B, int64_t J, int64_t H, int64_t W,
                                        float* coords, float* stats, void* stream) {
  if (!logits || !coords || !stats) PL_FAIL(PL_EINVAL, "pl_softargmax3d_nhwc_fwd: null pointer");
  if (B <= 0 || J <= 0 || H <= 0 || W <= 0 || B * J > 0x7fffffff || H * W > (1 << 24))
    PL_FAIL(PL_ESHAPE, "pl_softargmax3d_nhwc_fwd: bad dims");
  hipLaunchKernelGGL(softargmax_nhwc_fwd_kernel, dim3((unsigned)(B * J)), dim3(NTHR), 0, (hipStream_t)stream, logits,
                     (int)J, (int)H, (int)W, coords, stats);
  PL_CHECK_LAUNCH("softargmax_nhwc_fwd");
  return PL_OK;
}

// {S, 1/S} for the fp16 planes of dlogits: S the power of two that maps the bound 2 max_(b,j) sum_c |g_c| >= max |dlogit|
// into (2^13, 2^14] (softmax weights <= 1, index offsets < the map size); bound 0 / inf / nan -> 1.  One workgroup.
__global__ __launch_bounds__(NTHR) void softargmax_dl_scale_kernel(const float* __restrict__ g, int64_t rows, int ncoord,
                                                                   float* __restrict__ out) {
  __shared__ float sm[2 * (NTHR / 64)];
  float m = 0.f, bad = 0.f;                           // bad: a row sum that is inf / nan (fmaxf would drop a nan)
  for (int64_t r = threadIdx.x; r < rows; r += NTHR) {
    float a = 0.f;
    for (int c = 0; c < ncoord; ++c) a += fabsf(g[r * ncoord + c]);
    if (!(a < 3.0e38f)) bad = 1.f;
    m = fmaxf(m, a);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o)); bad = fmaxf(bad, __shfl_xor(bad, o)); }
  if ((threadIdx.x & 63) == 0) { sm[2 * (threadIdx.x >> 6)] = m; sm[2 * (threadIdx.x >> 6) + 1] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < NTHR / 64; ++w) { m = fmaxf(m, sm[2 * w]); bad = fmaxf(bad, sm[2 * w + 1]); }
    if (bad > 0.f) m = 0.f;
    const float bound = 2.0f * m;
    float S = 1.0f, Si = 1.0f;
    if (bound > 0.f && bound < 3.0e38f) {
      int e = 0;
      (void)frexpf(bound, &e);
      e = min(max(14 - e, -100), 100);
      S = ldexpf(1.0f, e); Si = ldexpf(1.0f, -e);
    }
    out[0] = S; out[1] = Si;
  }
}

extern "C" int pl_softargmax_dl_scale(const float* gcoords, int64_t rows, int ncoord, float* scale2, void* stream) {
  if (!gcoords || !scale2) PL_FAIL(PL_EINVAL, "pl_softargmax_dl_scale: null pointer");
  if (rows <= 0 || ncoord <= 0 || ncoord > 3) PL_FAIL(PL_ESHAPE, "pl_softargmax_dl_scale: bad dims");
  hipLaunchKernelGGL(softargmax_dl_scale_kernel, dim3(1), dim3(NTHR), 0, (hipStream_t)stream, gcoords, rows, ncoord, scale2);
  PL_CHECK_LAUNCH("softargmax_dl_scale");
  return PL_OK;
}

extern "C" int pl_softargmax3d_nhwc_bwd(const float* logits, const float* stats, const float* gcoords, int64_t B,
                                        int64_t J, int64_t H, int64_t W, float* dlogits, void* stream) {
  if (!dlogits) PL_FAIL(PL_EINVAL, "pl_softargmax3d_nhwc_bwd: null pointer");
  return pl_softargmax3d_nhwc_bwd_ex(logits, stats, gcoords, B, J, H, W, dlogits, nullptr, 0, nullptr, stream);
}

// + dl_planes (optional): dlogits also / only (dlogits == NULL) as operand planes for the final convolution's gradient
// GEMMs.  PL_F16X3: the planes hold dl_scale[0] * dlogits -- dl_scale = {S, 1/S} on the device, S a power of two the
// CALLER derives from the bound |dlogit| <= 2 max_(b,j) sum_c |g_c| (softmax weights <= 1, index offsets < the map size).
extern "C" int pl_softargmax3d_nhwc_bwd_ex(const float* logits, const float* stats, const float* gcoords, int64_t B,
                                           int64_t J, int64_t H, int64_t W, float* dlogits, void* dl_planes, int planes_mode,
                                           const float* dl_scale, void* stream) {
  if (!logits || !stats || !gcoords || (!dlogits && !dl_planes)) PL_FAIL(PL_EINVAL, "pl_softargmax3d_nhwc_bwd: null pointer");
  if ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15)
    PL_FAIL(PL_EINVAL, "pl_softargmax3d_nhwc_bwd: tensors not 16-byte aligned");
  if (B <= 0 || J <= 0 || H <= 0 || W <= 0 || B * J > 0x7fffffff || H * W > (1 << 24))
    PL_FAIL(PL_ESHAPE, "pl_softargmax3d_nhwc_bwd: bad dims");
  const int64_t n4 = B * H * W * J * 16;
  if (n4 > (int64_t)INT32_MAX * NTHR) PL_FAIL(PL_ESHAPE, "pl_softargmax3d_nhwc_bwd: too large");
  if (dl_planes && planes_mode == PL_F16X3 && !dl_scale) PL_FAIL(PL_EINVAL, "pl_softargmax3d_nhwc_bwd_ex: fp16 planes need dl_scale");
  PlaneOut po;
  PL_TRY(plane_out_of(planes_mode, dl_planes, n4 * 4, 1.0f, planes_mode == PL_F16X3 ? dl_scale : nullptr, &po,
                      "pl_softargmax3d_nhwc_bwd_ex"));
  hipLaunchKernelGGL(softargmax_nhwc_bwd_kernel, dim3((unsigned)((n4 + NTHR - 1) / NTHR)), dim3(NTHR), 0,
                     (hipStream_t)stream, logits, stats, gcoords, (int)J, (int)H, (int)W, n4, dlogits, po);
  PL_CHECK_LAUNCH("softargmax_nhwc_bwd");
  return PL_OK;
}
