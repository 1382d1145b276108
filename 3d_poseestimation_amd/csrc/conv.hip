// Convolution path of the phase4 / phase5 models, NHWC throughout: forward, input- and
// weight-gradient pieces.  The GEMM bodies are gemm_f32.hip's planes pipeline; this file holds the geometry, the
// split-K policy and the non-GEMM kernels.
//
//   pl_conv2d_nhwc_fwd       nn.Conv2d (+ folded eval BatchNorm2d, ReLU, residual, bias): a 1x1 is the plain GEMM, a
//                            KxK with Cin % 32 == 0 the same GEMM with a gathering A loader (ragged tiles clamped
//                            and masked); few-tile problems are split over K (slabs + reduce_slabs_epi_kernel);
//                            the 7x7 Cin = 3 stem is a direct fp32 kernel; anything else: im2col + generic GEMM
//   pl_conv2d_nhwc_wgrad     dW = dy^T x_gathered over the pixels, split-K by a cost model; the stem has its own
//                            LDS-staged fp32-MFMA kernel.  (dgrad is composed from the forward kernels, conv.py)
//   pl_maxpool3x3s2_nhwc*    nn.MaxPool2d(3, 2, 1), its backward from x, and the training pair with a tap index
//                            phase4_joined/Resnet.py:119
//   pl_deconv4x4s2_nhwc_fwd  nn.ConvTranspose2d(k=4, s=2, p=1, bias=False) (+ folded BN, ReLU)
//                            phase4_joined/Model.py:47-69: four 2x2-tap convolutions, one per output parity, in one
//                            grouped launch over the INPUT resolution, then one interleave pass
//   pl_nhwc_to_nchw          layout pass for callers that want [B][J*D][H][W] (Model_2D's head)
//   pl_colsum, pl_upsample2x_zero_nhwc   bias gradient; placement of a 1x1 stride-2 convolution's input gradient
#include "pl_internal.h"
#include "plane_store.h"

namespace pl {
namespace {

constexpr int NTHR = 256;

// col[m][k], m = (b, oh, ow), k = (kh*KW + kw)*Cin + ci, zero outside the image and for k >= K (row padding)
__global__ __launch_bounds__(NTHR) void im2col_nhwc_kernel(const float* __restrict__ x, int H, int W, int Cin,
                                                           int Ho, int Wo, int KH, int KW, int stride, int pad,
                                                           int64_t M, int K, int Kp, float* __restrict__ col) {
  const int64_t t = (int64_t)blockIdx.x * NTHR + threadIdx.x;
  if (t >= M * Kp) return;
  const int64_t m = t / Kp;
  const int k = (int)(t - m * Kp);
  float v = 0.f;
  if (k < K) {
    const int ci = k % Cin, tap = k / Cin, kw = tap % KW, kh = tap / KW;
    const int ow = (int)(m % Wo);
    const int64_t r = m / Wo;
    const int oh = (int)(r % Ho);
    const int64_t b = r / Ho;
    const int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
    if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = x[((b * H + ih) * W + iw) * Cin + ci];
  }
  col[t] = v;
}

// weights [Cout][K] -> [Cout][Kp] zero padded (once per call of the fallback path; tiny)
__global__ void pad_rows_kernel(const float* __restrict__ w, int rows, int K, int Kp, float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * Kp) return;
  const int r = t / Kp, k = t - r * Kp;
  out[t] = k < K ? w[(size_t)r * K + k] : 0.f;
}

__global__ __launch_bounds__(NTHR) void maxpool3x3s2_nhwc_kernel(const float* __restrict__ x, int H, int W, int C,
                                                                 int Ho, int Wo, int64_t n4,
                                                                 float* __restrict__ y) {
  const int64_t t = (int64_t)blockIdx.x * NTHR + threadIdx.x;     // one float4 of channels per thread
  if (t >= n4) return;
  const int c4 = C >> 2;
  const int c = (int)(t % c4) * 4;
  int64_t r = t / c4;
  const int ow = (int)(r % Wo); r /= Wo;
  const int oh = (int)(r % Ho);
  const int64_t b = r / Ho;
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
      if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
        const float4 v = *reinterpret_cast<const float4*>(x + ((b * H + ih) * W + iw) * C + c);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
  *reinterpret_cast<float4*>(y + t * 4) = m;
}

// backward of the 3x3 / stride 2 / pad 1 max-pool: one thread per INPUT float4; an input pixel lies in at most
// 2 x 2 windows, and it receives a window's gradient where it is that window's FIRST maximum in (kh, kw) order
// (torch's tie rule) -- recomputed from x and y, no index tensor, no atomics.
__global__ __launch_bounds__(NTHR) void maxpool3x3s2_nhwc_bwd_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ dy, int H, int W, int C,
                                                                     int Ho, int Wo, int64_t n4,
                                                                     float* __restrict__ dx) {
  const int64_t t = (int64_t)blockIdx.x * NTHR + threadIdx.x;
  if (t >= n4) return;
  const int c4 = C >> 2;
  const int c = (int)(t % c4) * 4;
  int64_t r = t / c4;
  const int iw = (int)(r % W); r /= W;
  const int ih = (int)(r % H);
  const int64_t b = r / H;
  const float4 xv = *reinterpret_cast<const float4*>(x + t * 4);
  const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  for (int oh = max(0, ih / 2); oh <= min(Ho - 1, (ih + 1) / 2); ++oh)
    for (int ow = max(0, iw / 2); ow <= min(Wo - 1, (iw + 1) / 2); ++ow) {
      const int kh0 = ih - (oh * 2 - 1), kw0 = iw - (ow * 2 - 1);            // this pixel's tap in that window
      if (kh0 < 0 || kh0 > 2 || kw0 < 0 || kw0 > 2) continue;
      const float4 gv = *reinterpret_cast<const float4*>(dy + ((b * Ho + oh) * Wo + ow) * C + c);
      const float gs[4] = {gv.x, gv.y, gv.z, gv.w};
      bool first[4] = {true, true, true, true};                               // no earlier tap holds a value >= ours
      bool ismax[4] = {true, true, true, true};                               // no later tap holds a value > ours
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int jh = oh * 2 - 1 + kh, jw = ow * 2 - 1 + kw;
          if ((unsigned)jh >= (unsigned)H || (unsigned)jw >= (unsigned)W || (kh == kh0 && kw == kw0)) continue;
          const float4 ov = *reinterpret_cast<const float4*>(x + ((b * H + jh) * W + jw) * C + c);
          const float os[4] = {ov.x, ov.y, ov.z, ov.w};
          const bool earlier = kh * 3 + kw < kh0 * 3 + kw0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (earlier) first[j] = first[j] && !(os[j] >= xs[j]);
            else ismax[j] = ismax[j] && !(os[j] > xs[j]);
          }
        }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (first[j] && ismax[j]) g[j] += gs[j];
    }
  *reinterpret_cast<float4*>(dx + t * 4) = make_float4(g[0], g[1], g[2], g[3]);
}

// Training-mode pair: the forward also records WHICH tap (kh*3 + kw, first maximum in that order = torch's tie
// rule) produced every output element, one byte each; the backward then needs neither x nor the nine-tap
// recomputation: an input pixel looks at the <= 4 windows that contain it and takes dy where the recorded tap is
// its own (2.9 -> 0.5 ms for the stem's map at B = 256: 1 GB of dx written, 0.34 GB of dy + indices read).
__global__ __launch_bounds__(NTHR) void maxpool3x3s2_nhwc_idx_kernel(const float* __restrict__ x, int H, int W, int C,
                                                                     int Ho, int Wo, int64_t n4, float* __restrict__ y,
                                                                     uchar4* __restrict__ idx) {
  const int64_t t = (int64_t)blockIdx.x * NTHR + threadIdx.x;
  if (t >= n4) return;
  const int c4 = C >> 2;
  const int c = (int)(t % c4) * 4;
  int64_t r = t / c4;
  const int ow = (int)(r % Wo); r /= Wo;
  const int oh = (int)(r % Ho);
  const int64_t b = r / Ho;
  float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  unsigned char k[4] = {0, 0, 0, 0};
  bool any = false;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
      if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
        const float4 v = *reinterpret_cast<const float4*>(x + ((b * H + ih) * W + iw) * C + c);
        const float vs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!any || vs[j] > m[j]) { m[j] = vs[j]; k[j] = (unsigned char)(kh * 3 + kw); }
        any = true;
      }
    }
  *reinterpret_cast<float4*>(y + t * 4) = make_float4(m[0], m[1], m[2], m[3]);
  idx[t] = make_uchar4(k[0], k[1], k[2], k[3]);
}

__global__ __launch_bounds__(NTHR) void maxpool3x3s2_nhwc_bwd_idx_kernel(const uchar4* __restrict__ idx,
                                                                         const float* __restrict__ dy, int H, int W,
                                                                         int C, int Ho, int Wo, int64_t n4,
                                                                         float* __restrict__ dx) {
  const int64_t t = (int64_t)blockIdx.x * NTHR + threadIdx.x;
  if (t >= n4) return;
  const int c4 = C >> 2;
  const int cq = (int)(t % c4);
  int64_t r = t / c4;
  const int iw = (int)(r % W); r /= W;
  const int ih = (int)(r % H);
  const int64_t b = r / H;
  // a pixel lies in the windows oh in {ih/2, (ih+1)/2} x ow in {iw/2, (iw+1)/2} (one or two per axis): the (at most four)
  // index words and gradients are requested together, at clamped coordinates, and the window tests applied to the values in
  // the loops' order -- as nested loops with the tests around the loads every window was its own dependent round trip
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  const int ohs[2] = {ih / 2, (ih + 1) / 2}, ows[2] = {iw / 2, (iw + 1) / 2};
  uchar4 kk[4];
  float4 gg[4];
  bool use[4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int oh = ohs[a], ow = ows[c];
      const int kh0 = ih - (oh * 2 - 1), kw0 = iw - (ow * 2 - 1);            // this pixel's tap in that window
      use[a * 2 + c] = oh <= Ho - 1 && ow <= Wo - 1 && (a == 0 || ohs[1] != ohs[0]) && (c == 0 || ows[1] != ows[0]) &&
                       kh0 >= 0 && kh0 <= 2 && kw0 >= 0 && kw0 <= 2;
      const int64_t o = ((b * Ho + min(oh, Ho - 1)) * Wo + min(ow, Wo - 1)) * c4 + cq;
      kk[a * 2 + c] = idx[o];
      gg[a * 2 + c] = *reinterpret_cast<const float4*>(dy + o * 4);
    }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int q = a * 2 + c;
      if (!use[q]) continue;
      const unsigned char mine = (unsigned char)((ih - (ohs[a] * 2 - 1)) * 3 + (iw - (ows[c] * 2 - 1)));
      if (kk[q].x == mine) g[0] += gg[q].x;
      if (kk[q].y == mine) g[1] += gg[q].y;
      if (kk[q].z == mine) g[2] += gg[q].z;
      if (kk[q].w == mine) g[3] += gg[q].w;
    }
  *reinterpret_cast<float4*>(dx + t * 4) = make_float4(g[0], g[1], g[2], g[3]);
}

// part[chunk][c] = sum over the chunk's rows of X[r][c]; grid = (ceil(cols/256), chunks), rows interleaved by 4*chunks
__global__ __launch_bounds__(NTHR) void colsum_wide_kernel(const float* __restrict__ X, int rows, int cols,
                                                           float* __restrict__ part) {
  __shared__ float4 sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const bool active = c < cols;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    // eight rows' loads in flight per wave, added in row order (one load per trip left 1 KB per wave in flight:
    // 2 TB/s on the 4.5 GB dlogits of the phase4 head)
    const int step = gridDim.y * 4;
    int r = blockIdx.y * 4 + wave;
    for (; (int64_t)r + 7 * (int64_t)step < rows; r += 8 * step) {
      float4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4*>(X + (size_t)(r + q * step) * cols + c);
#pragma unroll
      for (int q = 0; q < 8; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
    }
    for (; r < rows; r += step) {
      const float4 v = *reinterpret_cast<const float4*>(X + (size_t)r * cols + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  sm[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && active) {
    float4 t = sm[0][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) { t.x += sm[w][lane].x; t.y += sm[w][lane].y; t.z += sm[w][lane].z; t.w += sm[w][lane].w; }
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.y * cols + c) = t;
  }
}

// the same over a tensor that exists only as operand planes (the conv head's dlogits): x = (h + l / 2048) * inv[0]
// KIND is a template parameter: with the plane format tested per load (load_planes4's runtime `kind`) every load sat in its
// own basic block behind an s_waitcnt vmcnt(0) -- one 512-byte request in flight per wave, 2.1 TB/s on the 4.5 GB dlogits.
template <int KIND>
__device__ __forceinline__ float4 colsum_planes_rows(const unsigned short* __restrict__ h, const unsigned short* __restrict__ l,
                                                     float iv, int rows, int cols, int c, int wave) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const int step = gridDim.y * 4;
  int r = blockIdx.y * 4 + wave;
  for (; (int64_t)r + 7 * (int64_t)step < rows; r += 8 * step) {       // (as colsum_wide_kernel)
    float4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = load_planes4(KIND, h, l, (size_t)(r + q * step) * cols + c, iv);
#pragma unroll
    for (int q = 0; q < 8; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
  }
  for (; r < rows; r += step) {
    const float4 v = load_planes4(KIND, h, l, (size_t)r * cols + c, iv);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  return s;
}

__global__ __launch_bounds__(NTHR) void colsum_wide_planes_kernel(const unsigned short* __restrict__ h, const unsigned short* __restrict__ l,
                                                                  int kind, const float* __restrict__ inv, int rows, int cols,
                                                                  float* __restrict__ part) {
  __shared__ float4 sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const bool active = c < cols;
  const float iv = inv ? inv[0] : 1.0f;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) s = kind == 2 ? colsum_planes_rows<2>(h, l, iv, rows, cols, c, wave) : colsum_planes_rows<1>(h, l, iv, rows, cols, c, wave);
  sm[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && active) {
    float4 t = sm[0][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) { t.x += sm[w][lane].x; t.y += sm[w][lane].y; t.z += sm[w][lane].z; t.w += sm[w][lane].w; }
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.y * cols + c) = t;
  }
}

// y[b][2a][2c][:] = x[b][a][c][:], zero elsewhere: the input gradient of a 1x1 stride-2 convolution is dy W placed
// on the even pixels
__global__ __launch_bounds__(NTHR) void upsample2x_zero_kernel(const float* __restrict__ x, int Hi, int Wi, int C,
                                                               int64_t n4_out, float* __restrict__ y) {
  const int64_t t = (int64_t)blockIdx.x * NTHR + threadIdx.x;
  if (t >= n4_out) return;
  const int c4 = C >> 2;
  const int c = (int)(t % c4) * 4;
  int64_t r = t / c4;
  const int ow = (int)(r % (2 * Wi)); r /= 2 * Wi;
  const int oh = (int)(r % (2 * Hi));
  const int64_t b = r / (2 * Hi);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!(oh & 1) && !(ow & 1)) v = *reinterpret_cast<const float4*>(x + ((b * Hi + (oh >> 1)) * Wi + (ow >> 1)) * C + c);
  *reinterpret_cast<float4*>(y + t * 4) = v;
}

// y[b][2a+ph][2c+pw][:] = part[ph*2+pw][b][a][c][:]
__global__ __launch_bounds__(NTHR) void deconv_interleave_kernel(const float* __restrict__ part, int Hi, int Wi, int C,
                                                                 int64_t n4_per_part, float* __restrict__ y) {
  const int64_t t = (int64_t)blockIdx.x * NTHR + threadIdx.x;
  if (t >= n4_per_part * 4) return;
  const int par = (int)(t / n4_per_part);
  int64_t r = t - par * n4_per_part;
  const int c4 = C >> 2;
  const int c = (int)(r % c4) * 4; r /= c4;
  const int cw = (int)(r % Wi); r /= Wi;
  const int a = (int)(r % Hi);
  const int64_t b = r / Hi;
  const int oh = 2 * a + (par >> 1), ow = 2 * cw + (par & 1);
  const float4 v = *reinterpret_cast<const float4*>(part + t * 4);
  *reinterpret_cast<float4*>(y + ((b * 2 * Hi + oh) * 2 * Wi + ow) * C + c) = v;
}

// out[b][c][p] = in[b][p][c]   (p = pixel); 32x32 tiles through LDS, both sides coalesced (any P, C)
__global__ __launch_bounds__(1024) void nhwc_to_nchw_kernel(const float* __restrict__ in, int P, int C,
                                                            float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int64_t b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (p0 + ty < P && c0 + tx < C) tile[ty][tx] = in[(b * P + p0 + ty) * C + c0 + tx];
  __syncthreads();
  if (c0 + ty < C && p0 + tx < P) out[(b * C + c0 + ty) * P + p0 + tx] = tile[tx][ty];
}

// the same for P % 4 == 0 and C % 4 == 0 (the head: P = 4096, C = 1088): 64x64 tiles, 16-byte loads along C and
// 16-byte stores along P, 256 B per row on both sides (the 32x32 scalar version moved 0.77 TB/s, 3 ms per call at
// B = 256); the 65-float row stride keeps both LDS phases conflict-free
__global__ __launch_bounds__(NTHR) void nhwc_to_nchw_vec_kernel(const float* __restrict__ in, int P, int C,
                                                                float* __restrict__ out) {
  __shared__ float tile[64][65];                   // [p][c]
  const int64_t b = blockIdx.z;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = p0 + ty + 16 * i, c = c0 + tx * 4;
    if (p < P && c < C) {
      const float4 v = *reinterpret_cast<const float4*>(in + (b * P + p) * C + c);
      float* t = &tile[ty + 16 * i][tx * 4];
      t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cl = ty + 16 * i, c = c0 + cl, p = p0 + tx * 4;
    if (c < C && p < P)
      *reinterpret_cast<float4*>(out + (b * C + c) * P + p) =
          make_float4(tile[tx * 4][cl], tile[tx * 4 + 1][cl], tile[tx * 4 + 2][cl], tile[tx * 4 + 3][cl]);
  }
}

// The 7x7 / stride 2 / pad 3 stem on 3 input channels (Resnet.py:112-113): K = 147 is no multiple of anything the
// matrix pipeline likes and the explicit im2col it needed cost more than the GEMM it fed (0.6 + 0.56 ms at
// B = 64).  Direct form on the vector ALU, exact fp32: one thread = one output pixel x all 64 channels, the
// 37.6 KB filter in LDS as [tap][ci][64 co] so a tap's 64 weights are 16 broadcast ds_read_b128; 147 x 64 FMAs
// per pixel (19.7 GFLOP per 64 frames) with BatchNorm + ReLU folded into the store.
// (Tried: two horizontally adjacent output pixels per thread -- a tap's 16 ds_read_b128 feeding both pixels' FMAs, the two
//  windows sharing a 9-column strip of the input row.  128 accumulators + the strip need all 256 VGPRs: one wave per SIMD
//  instead of four, and 530 us against 388 us at B = 64.  Not kept.)
__global__ __launch_bounds__(NTHR) void stem7x7_c3_kernel(const float* __restrict__ x, int H, int W, int Ho, int Wo,
                                                          int64_t npix, const float* __restrict__ w,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int relu,
                                                          float* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) float ws[147 * 64];     // [(kh*7 + kw)*3 + ci][co]
  // The filter is staged ONCE per workgroup and the workgroup then walks pixel tiles (grid = what fits on the chip): with one
  // tile per workgroup, 16,384 workgroups at B = 256 each began with this transposing copy -- and as an element loop it kept one
  // 4-byte load per thread in flight, 37 dependent round trips.  Here: batches of eight loads in flight.
  for (int e0 = threadIdx.x; e0 < 147 * 64; e0 += 8 * NTHR) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = min(e0 + u * NTHR, 147 * 64 - 1);
      t[u] = w[(e & 63) * 147 + (e >> 6)];                        // w is OHWI: [co][kh][kw][ci] = [co][k]
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + u * NTHR < 147 * 64) ws[e0 + u * NTHR] = t[u];
  }
  __syncthreads();
  for (int64_t p = (int64_t)blockIdx.x * NTHR + threadIdx.x; p < npix; p += (int64_t)gridDim.x * NTHR) {
  const int ow = (int)(p % Wo);
  const int64_t r = p / Wo;
  const int oh = (int)(r % Ho);
  const int64_t b = r / Ho;
  float acc[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) acc[c] = 0.f;
  const float* xb = x + b * H * W * 3;
  for (int kh = 0; kh < 7; ++kh) {
    const int ih = oh * 2 - 3 + kh;
    if ((unsigned)ih >= (unsigned)H) continue;
    for (int kw = 0; kw < 7; ++kw) {
      const int iw = ow * 2 - 3 + kw;
      if ((unsigned)iw >= (unsigned)W) continue;
      const float* px = xb + ((size_t)ih * W + iw) * 3;
      const float v[3] = {px[0], px[1], px[2]};
      const float4* wt = reinterpret_cast<const float4*>(ws + (kh * 7 + kw) * 3 * 64);
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float4 f = wt[ci * 16 + q];
          acc[4 * q + 0] = fmaf(v[ci], f.x, acc[4 * q + 0]);
          acc[4 * q + 1] = fmaf(v[ci], f.y, acc[4 * q + 1]);
          acc[4 * q + 2] = fmaf(v[ci], f.z, acc[4 * q + 2]);
          acc[4 * q + 3] = fmaf(v[ci], f.w, acc[4 * q + 3]);
        }
    }
  }
  float4* out = reinterpret_cast<float4*>(y + p * 64);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = acc[4 * q + j];
      if (scale) t = fmaf(t, scale[4 * q + j], shift[4 * q + j]);
      if (relu) t = fmaxf(t, 0.f);
      o[j] = t;
    }
    out[q] = make_float4(o[0], o[1], o[2], o[3]);
  }
  }                                                               // (pixel tiles of this workgroup)
}

// Weight gradient of the stem: dw[co][k] = sum over pixels of dy[p][co] * x_gathered[p][k], k = (kh*7 + kw)*3 + ci
// -- a 64 x 147 GEMM over B*Ho*Wo pixels whose B operand is a 7x7x3 window per pixel.  As a gathered-B launch of
// the planes kernel it ran 1.65 ms at B = 32 (8-byte scattered loads); a first LDS-staged VALU kernel, 0.83 ms
// (four waves queueing on broadcast LDS reads).  Here a workgroup walks whole output rows: the row's dy [Wo][64]
// and the 7 zero-padded input rows under it are staged in LDS once, and each wave runs exact-fp32 MFMAs
// (32x32x2: A = dy^T, 2 co-tiles; B = the windows, 5 k-tiles of which 147 columns are real) over its quarter of
// the row's pixel pairs -- 7 LDS reads per 10 MFMAs.  Two workgroups per CU overlap staging with MFMA work.
// Waves are combined through LDS at the end; one partial slab per workgroup, fixed-order reduce afterwards.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int STEM_NT = 5;               // ceil(147 / 32)
constexpr int STEM_LDW = STEM_NT * 32;   // combine buffer row
__global__ __launch_bounds__(NTHR, 2) void stem7x7_c3_wgrad_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ dy, int B, int H, int W,
                                                                   int Ho, int Wo, float* __restrict__ part) {
  extern __shared__ float smem[];
  const int PW = 2 * Wo + 5;                       // padded input row: iw = -3 .. 2*Wo + 1
  float* dys = smem;                               // [Wo][64]
  float* xs = smem + (size_t)Wo * 64;              // [7][PW][3]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  int boff[STEM_NT];
#pragma unroll
  for (int nt = 0; nt < STEM_NT; ++nt) {
    const int k = min(nt * 32 + j, 146);           // columns >= 147 repeat the last one and are never stored
    const int kh = k / 21;
    boff[nt] = kh * PW * 3 + (k - kh * 21);        // (kw, ci) is contiguous in a padded row
  }
  f32x16 acc[2][STEM_NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < STEM_NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  const int rows = B * Ho;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int b = r / Ho, oh = r - b * Ho;
    __syncthreads();
    const float4* src = reinterpret_cast<const float4*>(dy + (size_t)r * Wo * 64);
    // both copies in batches of eight loads in flight per thread (clamped addresses, the bounds applied to the value): as
    // element loops they were ~30 dependent memory round trips per output row -- most of this kernel's time
    const int n16 = Wo * 16;
    for (int e0 = threadIdx.x; e0 < n16; e0 += 8 * NTHR) {
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = src[min(e0 + u * NTHR, n16 - 1)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (e0 + u * NTHR < n16) reinterpret_cast<float4*>(dys)[e0 + u * NTHR] = t[u];
    }
    const int nx = 7 * PW * 3;
    for (int e0 = threadIdx.x; e0 < nx; e0 += 8 * NTHR) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = min(e0 + u * NTHR, nx - 1);
        const int kh = e / (PW * 3), q = e - kh * PW * 3;
        const int px = q / 3, iw = px - 3, ci = q - px * 3, ih = oh * 2 - 3 + kh;
        const bool in = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        const float v = x[(((size_t)b * H + min(max(ih, 0), H - 1)) * W + min(max(iw, 0), W - 1)) * 3 + ci];
        t[u] = in ? v : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (e0 + u * NTHR < nx) xs[e0 + u * NTHR] = t[u];
    }
    __syncthreads();
    for (int ow = wave * 2; ow < Wo; ow += 8) {    // lane half h takes pixel ow + h of the pair
      const bool valid = ow + h < Wo;
      const int p = min(ow + h, Wo - 1);
      const float a0 = valid ? dys[p * 64 + j] : 0.f, a1 = valid ? dys[p * 64 + 32 + j] : 0.f;
      float bv[STEM_NT];
#pragma unroll
      for (int nt = 0; nt < STEM_NT; ++nt) bv[nt] = xs[boff[nt] + p * 6];
#pragma unroll
      for (int nt = 0; nt < STEM_NT; ++nt) {
        acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[nt], acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[nt], acc[1][nt], 0, 0, 0);
      }
    }
  }
  for (int w = 0; w < NTHR / 64; ++w) {            // fixed-order combine of the four waves: [64][STEM_LDW] in LDS
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < STEM_NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = mt * 32 + (r >> 2) * 8 + h * 4 + (r & 3);
            float* d = smem + co * STEM_LDW + nt * 32 + j;
            *d = w == 0 ? acc[mt][nt][r] : *d + acc[mt][nt][r];
          }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 147; e += NTHR) {
    const int co = e / 147, k = e - co * 147;
    part[(size_t)blockIdx.x * 64 * 147 + e] = smem[co * STEM_LDW + k];
  }
}
constexpr int STEM_WGRAD_WGS = 512;
// the staged row pair (dy [Wo][64] + 7 padded input rows) must fit 64 KB of dynamic LDS: 424 Wo + 420 bytes
constexpr int STEM_WGRAD_MAX_WO = 152;     // wider maps take the generic (im2col) path

bool implicit_ok(int64_t M, int64_t Cin, int64_t Cout, int64_t K) {
  (void)Cout; (void)M;                   // any width, any pixel count: ragged last tiles are clamped and masked
  return Cin % 32 == 0 && K % 32 == 0;
}

// y = epilogue(sum of split-K slabs): the epilogue of gemm_epilogue (bias, scale/shift, ReLU, residual, ReLU) applied
// after the fixed-order slab sum; one float4 per thread, N % 4 == 0
__global__ __launch_bounds__(NTHR) void reduce_slabs_epi_kernel(const float* __restrict__ slabs, int nslab, int64_t n4,
                                                                int N, const float* __restrict__ bias,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int relu,
                                                                const float* __restrict__ resid,
                                                                float* __restrict__ y) {
  const int64_t t = (int64_t)blockIdx.x * NTHR + threadIdx.x;
  if (t >= n4) return;
  float4 v = reinterpret_cast<const float4*>(slabs)[t];
  for (int sl = 1; sl < nslab; ++sl) {
    const float4 q = reinterpret_cast<const float4*>(slabs)[(size_t)sl * n4 + t];
    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
  }
  const int c = (int)((t * 4) % N);
  float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (bias) o[j] += bias[c + j];
    if (scale) o[j] = fmaf(o[j], scale[c + j], shift[c + j]);
    if (relu == 1) o[j] = fmaxf(o[j], 0.f);
  }
  if (resid) {
    const float4 q = reinterpret_cast<const float4*>(resid)[t];
    o[0] += q.x; o[1] += q.y; o[2] += q.z; o[3] += q.w;
  }
  if (relu == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
  }
  reinterpret_cast<float4*>(y)[t] = make_float4(o[0], o[1], o[2], o[3]);
}

// Split-K of a FORWARD / input-gradient convolution GEMM (M = pixels, N = Cout, K = taps x Cin): the deep layers
// have few output tiles (layer4 at B = 32: 2048 x 512 = 64 tiles of 128 x 128 for 256 CUs, 144 K tiles each), so K
// is cut into s slices whose plain sums land in slabs, and reduce_slabs_epi_kernel sums them and applies the
// epilogue.  Cost model as wgrad_splits, plus the slab traffic ((s + 1) M N floats at ~4 TB/s) in units of one
// K-tile step (~1.2 us); s divides the K tiles, slices >= 8 tiles, s <= 8.  1 = no split.
int conv_fwd_splits(int64_t M, int64_t N, int64_t K, int arith) {
  if (arith != PL_BF16X6 || (N & 3) || K % 32) return 1;
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128), T = K / 32;
  if (tiles >= 192) return 1;
  int best = 1;
  double best_cost = (double)((tiles + 255) / 256) * (T + 6);
  for (int s = 2; s <= 8; ++s) {
    if (T % s || T / s < 8) continue;
    const double traffic = (double)(s + 1) * M * N * 4.0 / 4.0e6 / 1.2;
    const double cost = (double)((tiles * s + 255) / 256) * (T / s + 6) + traffic;
    if (cost < best_cost * 0.9) { best_cost = cost; best = s; }
  }
  return best;
}

int conv_core(const float* x, int64_t B, int64_t H, int64_t W, int64_t Cin, const float* w, int64_t Cout, int KH,
              int KW, int stride, int pad_h, int pad_w, int64_t Ho, int64_t Wo, const float* scale,
              const float* shift, const float* bias, int relu, const float* resid, float* y, void* scratch,
              size_t scratch_bytes, hipStream_t s, int arith = PL_BF16X6) {
  const int64_t M = B * Ho * Wo, K = (int64_t)KH * KW * Cin;
  if (M > INT32_MAX || K > INT32_MAX) PL_FAIL(PL_ESHAPE, "conv: problem too large");
  GemmArgs g = {};
  g.A = x; g.B = w; g.C = y; g.M = (int)M; g.N = (int)Cout; g.K = (int)K;
  g.lda = (int)K; g.ldb = (int)K; g.ldc = (int)Cout; g.split_k = 1;
  g.bias = bias; g.col_scale = scale; g.col_shift = shift; g.relu = relu; g.resid = resid;
  g.arith = arith;
  if (Cin == 3 && KH == 7 && KW == 7 && stride == 2 && pad_h == 3 && pad_w == 3 && Cout == 64 && !bias && !resid &&
      relu != 2) {                                           // the stem: direct fp32 kernel, no im2col
    const int64_t tiles = (M + NTHR - 1) / NTHR;                // 256 CUs x 4 workgroups of 37.6 KB LDS each
    hipLaunchKernelGGL(stem7x7_c3_kernel, dim3((unsigned)(tiles < 1024 ? tiles : 1024)), dim3(NTHR), 0, s, x, (int)H, (int)W,
                       (int)Ho, (int)Wo, M, w, scale, shift, relu, y);
    PL_CHECK_LAUNCH("stem7x7_c3");
    return PL_OK;
  }
  const int gemm_arith = arith == PL_BF16 ? 7 : arith;     // PL_BF16 on the planes pipeline (gemm_f32.hip)
  const bool one = KH == 1 && KW == 1 && stride == 1 && pad_h == 0 && pad_w == 0;
  const bool implicit = !one && implicit_ok(M, Cin, Cout, K);
  // split-K (whole 128 x 128 tiles only: the split kernels' plain store is the tile store)
  const int splits = (one || implicit) && M % 128 == 0 && Cout % 128 == 0 ? conv_fwd_splits(M, Cout, K, arith) : 1;
  if (splits > 1) {
    const size_t need = (size_t)splits * M * Cout * sizeof(float);
    if (!scratch || scratch_bytes < need)
      PL_FAIL(PL_EWORKSPACE, "conv: this shape is split %d ways over K and needs %zu scratch bytes (got %zu)", splits,
              need, scratch_bytes);
    g.C = static_cast<float*>(scratch); g.split_k = splits;
    g.bias = nullptr; g.col_scale = nullptr; g.col_shift = nullptr; g.relu = 0; g.resid = nullptr;
  }
  if (one) {
    g.arith = gemm_arith;
    PL_TRY(launch_gemm_f32(kNT, g, s));
  } else if (implicit) {
    g.conv_cin = (int)Cin; g.conv_h = (int)H; g.conv_w = (int)W; g.conv_ho = (int)Ho; g.conv_wo = (int)Wo;
    g.conv_kw = KW; g.conv_stride = stride; g.conv_pad_h = pad_h; g.conv_pad_w = pad_w;
    PL_TRY(launch_conv_nhwc(g, s));
  }
  if (splits > 1) {
    const int64_t n4 = M * Cout / 4;
    hipLaunchKernelGGL(reduce_slabs_epi_kernel, dim3((unsigned)((n4 + NTHR - 1) / NTHR)), dim3(NTHR), 0, s,
                       static_cast<const float*>(scratch), splits, n4, (int)Cout, bias, scale, shift, relu, resid, y);
    PL_CHECK_LAUNCH("reduce_slabs_epi");
  }
  if (one || implicit) return PL_OK;
  // fallback: explicit im2col (rows zero-padded to a multiple of 32 floats = whole K tiles) + the generic GEMM
  if (pad_h != pad_w) PL_FAIL(PL_ESHAPE, "conv: asymmetric padding only on the implicit path");
  const int64_t Kp = (K + 31) / 32 * 32;
  const size_t need = ((size_t)M * Kp + (size_t)Cout * Kp) * sizeof(float);
  if (!scratch || scratch_bytes < need)
    PL_FAIL(PL_EWORKSPACE, "conv: this shape takes the im2col path and needs %zu scratch bytes (got %zu)", need,
            scratch_bytes);
  float* col = static_cast<float*>(scratch);
  float* wp = col + (size_t)M * Kp;
  const int64_t tot = M * Kp;
  if (tot > (int64_t)INT32_MAX * NTHR) PL_FAIL(PL_ESHAPE, "conv: im2col too large");
  hipLaunchKernelGGL(im2col_nhwc_kernel, dim3((unsigned)((tot + NTHR - 1) / NTHR)), dim3(NTHR), 0, s, x, (int)H,
                     (int)W, (int)Cin, (int)Ho, (int)Wo, KH, KW, stride, pad_h, M, (int)K, (int)Kp, col);
  PL_CHECK_LAUNCH("im2col_nhwc");
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((Cout * Kp + 255) / 256)), dim3(256), 0, s, w, (int)Cout,
                     (int)K, (int)Kp, wp);
  PL_CHECK_LAUNCH("pad_rows");
  g.A = col; g.B = wp; g.K = (int)Kp; g.lda = (int)Kp; g.ldb = (int)Kp;
  g.arith = gemm_arith;
  return launch_gemm_f32(kNT, g, s);
}

}  // namespace
}  // namespace pl

using namespace pl;

// The 7x7 / Cin 3 stem has a scratch-free kernel of its own, but only for the epilogue it folds (BatchNorm scale / shift,
// ReLU before the add): with a bias, a residual or relu == 2 the call takes the im2col path.  The plain query cannot
// know the epilogue, so it answers for that worst case; pl_conv2d_nhwc_scratch_bytes_ex takes the epilogue flags
// and answers exactly (0 for the stem as Model_3D calls it).
static size_t conv_scratch(int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int KH, int KW, int stride, int pad,
                           bool stem_kernel_ok) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return 0;
  const int64_t Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 0;
  const int64_t M = B * Ho * Wo, K = (int64_t)KH * KW * Cin;
  if (stem_kernel_ok && Cin == 3 && KH == 7 && KW == 7 && stride == 2 && pad == 3 && Cout == 64) return 0;   // stem7x7_c3_kernel
  if ((KH == 1 && KW == 1 && stride == 1 && pad == 0) || implicit_ok(M, Cin, Cout, K)) {
    const int splits = M % 128 == 0 && Cout % 128 == 0 ? conv_fwd_splits(M, Cout, K, PL_BF16X6) : 1;   // split-K slabs
    return splits > 1 ? (size_t)splits * M * Cout * sizeof(float) : 0;
  }
  const int64_t Kp = (K + 31) / 32 * 32;
  return ((size_t)M * Kp + (size_t)Cout * Kp) * sizeof(float);
}

extern "C" size_t pl_conv2d_nhwc_scratch_bytes(int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int KH,
                                               int KW, int stride, int pad) {
  return conv_scratch(B, H, W, Cin, Cout, KH, KW, stride, pad, false);
}

extern "C" size_t pl_conv2d_nhwc_scratch_bytes_ex(int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int KH,
                                                  int KW, int stride, int pad, int has_bias, int has_resid, int relu) {
  return conv_scratch(B, H, W, Cin, Cout, KH, KW, stride, pad, !has_bias && !has_resid && relu != 2);
}

extern "C" int pl_conv2d_nhwc_fwd(const float* x, int64_t B, int64_t H, int64_t W, int64_t Cin, const float* w,
                                  int64_t Cout, int KH, int KW, int stride, int pad, const float* scale,
                                  const float* shift, const float* bias, int relu, const float* resid, float* y,
                                  int arith, void* scratch, size_t scratch_bytes, void* stream) {
  if (!x || !w || !y) PL_FAIL(PL_EINVAL, "pl_conv2d_nhwc_fwd: null pointer");
  if (arith != PL_BF16X6 && arith != PL_BF16) PL_FAIL(PL_EDTYPE, "pl_conv2d_nhwc_fwd: arith %d (PL_BF16X6 or PL_BF16)", arith);
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
    PL_FAIL(PL_ESHAPE, "pl_conv2d_nhwc_fwd: bad geometry");
  if ((scale != nullptr) != (shift != nullptr)) PL_FAIL(PL_EINVAL, "pl_conv2d_nhwc_fwd: scale without shift");
  if (relu < 0 || relu > 2) PL_FAIL(PL_EINVAL, "pl_conv2d_nhwc_fwd: relu=%d", relu);
  const int64_t Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  if (Ho <= 0 || Wo <= 0) PL_FAIL(PL_ESHAPE, "pl_conv2d_nhwc_fwd: empty output");
  return conv_core(x, B, H, W, Cin, w, Cout, KH, KW, stride, pad, pad, Ho, Wo, scale, shift, bias, relu, resid, y,
                   scratch, scratch_bytes, (hipStream_t)stream, arith);
}

extern "C" int pl_maxpool3x3s2_nhwc(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* y,
                                    void* stream) {
  if (!x || !y) PL_FAIL(PL_EINVAL, "pl_maxpool3x3s2_nhwc: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) PL_FAIL(PL_ESHAPE, "pl_maxpool3x3s2_nhwc: C %% 4 == 0 needed");
  const int64_t Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t n4 = B * Ho * Wo * (C >> 2);
  if (n4 > (int64_t)INT32_MAX * NTHR) PL_FAIL(PL_ESHAPE, "pl_maxpool3x3s2_nhwc: too large");
  hipLaunchKernelGGL(maxpool3x3s2_nhwc_kernel, dim3((unsigned)((n4 + NTHR - 1) / NTHR)), dim3(NTHR), 0,
                     (hipStream_t)stream, x, (int)H, (int)W, (int)C, (int)Ho, (int)Wo, n4, y);
  PL_CHECK_LAUNCH("maxpool3x3s2_nhwc");
  return PL_OK;
}

// w_sub: [4 parities][Cout][2][2][Cin], parity = (oh&1)*2 + (ow&1); tap (th, tw) of parity (ph, pw) is
// ConvTranspose2d weight[ci][co][kh][kw] with kh = ph ? 2 - 2*th : 3 - 2*th, kw likewise (conv.py deconv_subkernels)
extern "C" size_t pl_deconv4x4s2_nhwc_scratch_bytes(int64_t B, int64_t Hi, int64_t Wi, int64_t Cout) {
  return B > 0 && Hi > 0 && Wi > 0 && Cout > 0 ? (size_t)4 * B * Hi * Wi * Cout * sizeof(float) : 0;
}

extern "C" int pl_deconv4x4s2_nhwc_fwd(const float* x, int64_t B, int64_t Hi, int64_t Wi, int64_t Cin,
                                       const float* w_sub, int64_t Cout, const float* scale, const float* shift,
                                       int relu, float* y, int arith, void* scratch, size_t scratch_bytes,
                                       void* stream) {
  if (!x || !w_sub || !y || !scratch) PL_FAIL(PL_EINVAL, "pl_deconv4x4s2_nhwc_fwd: null pointer");
  if (arith != PL_BF16X6 && arith != PL_BF16) PL_FAIL(PL_EDTYPE, "pl_deconv4x4s2_nhwc_fwd: arith %d", arith);
  if (B <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0 || (Cout & 3)) PL_FAIL(PL_ESHAPE, "pl_deconv4x4s2_nhwc_fwd: bad geometry");
  if ((scale != nullptr) != (shift != nullptr)) PL_FAIL(PL_EINVAL, "pl_deconv4x4s2_nhwc_fwd: scale without shift");
  const size_t part = (size_t)B * Hi * Wi * Cout;
  if (scratch_bytes < 4 * part * sizeof(float)) PL_FAIL(PL_EWORKSPACE, "pl_deconv4x4s2_nhwc_fwd: scratch too small");
  if (!implicit_ok(B * Hi * Wi, Cin, Cout, 4 * Cin))
    PL_FAIL(PL_ESHAPE, "pl_deconv4x4s2_nhwc_fwd: needs Cin %% 32 == 0");
  hipStream_t s = (hipStream_t)stream;
  float* tmp = static_cast<float*>(scratch);
  GemmArgs g4[4];
  for (int par = 0; par < 4; ++par) {
    const int ph = par >> 1, pw = par & 1;
    GemmArgs g = {};
    g.A = x; g.B = w_sub + (size_t)par * Cout * 4 * Cin; g.C = tmp + par * part;
    g.M = (int)(B * Hi * Wi); g.N = (int)Cout; g.K = (int)(4 * Cin);
    g.lda = g.K; g.ldb = g.K; g.ldc = (int)Cout; g.split_k = 1;
    g.col_scale = scale; g.col_shift = shift; g.relu = relu; g.arith = arith;
    g.conv_cin = (int)Cin; g.conv_h = (int)Hi; g.conv_w = (int)Wi; g.conv_ho = (int)Hi; g.conv_wo = (int)Wi;
    g.conv_kw = 2; g.conv_stride = 1;
    // even output rows read input rows a-1, a (pad 1); odd ones a, a+1 (pad 0, the last tap runs off the edge)
    g.conv_pad_h = ph ? 0 : 1; g.conv_pad_w = pw ? 0 : 1;
    g4[par] = g;
  }
  PL_TRY(launch_conv_nhwc_group4(g4, s));
  const int64_t n4 = (int64_t)(part >> 2);
  hipLaunchKernelGGL(deconv_interleave_kernel, dim3((unsigned)((4 * n4 + NTHR - 1) / NTHR)), dim3(NTHR), 0, s, tmp,
                     (int)Hi, (int)Wi, (int)Cout, n4, y);
  PL_CHECK_LAUNCH("deconv_interleave");
  return PL_OK;
}

extern "C" int pl_nhwc_to_nchw(const float* in, int64_t B, int64_t P, int64_t C, float* out, void* stream) {
  if (!in || !out || in == out) PL_FAIL(PL_EINVAL, "pl_nhwc_to_nchw: null or aliased pointers");
  if (B <= 0 || P <= 0 || C <= 0 || B > 65535) PL_FAIL(PL_ESHAPE, "pl_nhwc_to_nchw: bad shape");
  if ((P & 3) == 0 && (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    dim3 grid((unsigned)((P + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)B);
    hipLaunchKernelGGL(nhwc_to_nchw_vec_kernel, grid, dim3(NTHR), 0, (hipStream_t)stream, in, (int)P, (int)C, out);
    PL_CHECK_LAUNCH("nhwc_to_nchw_vec");
    return PL_OK;
  }
  dim3 grid((unsigned)((P + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(1024), 0, (hipStream_t)stream, in, (int)P, (int)C, out);
  PL_CHECK_LAUNCH("nhwc_to_nchw");
  return PL_OK;
}

// ---- weight gradient -----------------------------------------------------------------------
// Split-K factor of a weight gradient: the pixel dimension is long (B*Ho*Wo) and the output small, so the slices
// fill the chip.  Picks the s minimising rounds x slice length, rounds = ceil(tiles*s / 256 CUs) (one workgroup
// per CU: 150 KB of LDS), slice length in 32-pixel tiles plus ~6 tiles' worth of prologue / epilogue -- any s,
// not only powers of two (18 tiles x 16 slices = 288 workgroups ran two rounds, 14 slices run one).  Slices are
// at least 8 tiles long and none is empty.  A fixed function of the shape, so results are reproducible.
static int wgrad_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128), T = K / 32;
  int best = 1;
  int64_t best_cost = (tiles + 255) / 256 * (T + 6);
  for (int s = 2; s <= 256; ++s) {
    const int64_t per = (T + s - 1) / s;
    if (per < 8) break;
    if ((int64_t)(s - 1) * per >= T) continue;                 // would leave the last slice empty
    const int64_t cost = (tiles * s + 255) / 256 * (per + 6);
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}

static bool wgrad_implicit_ok(int64_t Cin, int64_t Cout, int64_t N, int64_t K, int64_t Wo) {
  (void)N;                               // ragged Cout / KH*KW*Cin tiles are clamped and masked
  return (Cout & 1) == 0 && (Cin & 1) == 0 && Wo % 8 == 0 && K % 32 == 0;
}

// scratch: [split-K slabs: splits x Cout x N] then, on the fallback path, [im2col: pixels x N]
extern "C" size_t pl_conv2d_nhwc_wgrad_scratch_bytes(int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                                     int KH, int KW, int stride, int pad) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return 0;
  const int64_t Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 0;
  const int64_t K = B * Ho * Wo, N = (int64_t)KH * KW * Cin;
  if (Cin == 3 && KH == 7 && KW == 7 && stride == 2 && pad == 3 && Cout == 64 && Wo <= STEM_WGRAD_MAX_WO)
    return (size_t)STEM_WGRAD_WGS * 64 * 147 * sizeof(float);            // stem7x7_c3_wgrad_kernel partials
  const int s = wgrad_splits(Cout, N, K);
  size_t bytes = s > 1 ? (size_t)s * Cout * N * sizeof(float) : 0;
  if (!wgrad_implicit_ok(Cin, Cout, N, K, Wo)) bytes += (size_t)K * N * sizeof(float);
  return bytes;
}

// dw [Cout][KH][KW][Cin] = sum_{b,oh,ow} dy[b][oh][ow][co] * x[b][oh*s - p + kh][ow*s - p + kw][ci]
extern "C" int pl_conv2d_nhwc_wgrad(const float* x, int64_t B, int64_t H, int64_t W, int64_t Cin, const float* dy,
                                    int64_t Cout, int KH, int KW, int stride, int pad, float* dw, int arith,
                                    void* scratch, size_t scratch_bytes, void* stream) {
  if (arith != PL_BF16X6 && arith != PL_BF16) PL_FAIL(PL_EDTYPE, "pl_conv2d_nhwc_wgrad: arith %d (PL_BF16X6 or PL_BF16)", arith);
  if (!x || !dy || !dw) PL_FAIL(PL_EINVAL, "pl_conv2d_nhwc_wgrad: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
    PL_FAIL(PL_ESHAPE, "pl_conv2d_nhwc_wgrad: bad geometry");
  const int64_t Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  const int64_t K = B * Ho * Wo, N = (int64_t)KH * KW * Cin;
  if (Ho <= 0 || Wo <= 0 || K > INT32_MAX || N > INT32_MAX) PL_FAIL(PL_ESHAPE, "pl_conv2d_nhwc_wgrad: bad output size");
  if (Cin == 3 && KH == 7 && KW == 7 && stride == 2 && pad == 3 && Cout == 64 && Wo <= STEM_WGRAD_MAX_WO) {      // the stem
    const int nwg = STEM_WGRAD_WGS;
    const size_t need = (size_t)nwg * 64 * 147 * sizeof(float);
    if (!scratch || scratch_bytes < need) PL_FAIL(PL_EWORKSPACE, "pl_conv2d_nhwc_wgrad: needs %zu scratch bytes (got %zu)", need, scratch_bytes);
    const size_t lds = std::max(((size_t)Wo * 64 + 7 * (2 * (size_t)Wo + 5) * 3), (size_t)64 * STEM_LDW) * sizeof(float);
    hipLaunchKernelGGL(stem7x7_c3_wgrad_kernel, dim3(nwg), dim3(NTHR), lds, (hipStream_t)stream, x, dy, (int)B, (int)H,
                       (int)W, (int)Ho, (int)Wo, static_cast<float*>(scratch));
    PL_CHECK_LAUNCH("stem7x7_c3_wgrad");
    return launch_reduce_slabs(static_cast<const float*>(scratch), nwg, 64 * 147, dw, (hipStream_t)stream);
  }
  const bool implicit = wgrad_implicit_ok(Cin, Cout, N, K, Wo);
  int splits = wgrad_splits(Cout, N, K);
  // a 1x1 stride-1 convolution with whole tiles gathers nothing: dW = dy^T x is the plain TN GEMM (same thread
  // mapping, but no pixel decode and no border selects in the staging path) -- the lifter's dW kernel.  It wants equal K slices: the largest divisor of the K tiles <= splits.
  const bool plain_tn = KH == 1 && KW == 1 && stride == 1 && pad == 0 && Cout % 128 == 0 && Cin % 128 == 0 &&
                        K % 32 == 0 && arith == PL_BF16X6 && !getenv("POSELIFT_WGRAD_GATHER");
  if (plain_tn)
    while (splits > 1 && (K / 32) % splits) --splits;
  const size_t slab_bytes = splits > 1 ? (size_t)splits * Cout * N * sizeof(float) : 0;
  const size_t need = slab_bytes + ((implicit || plain_tn) ? 0 : (size_t)K * N * sizeof(float));
  if (need && (!scratch || scratch_bytes < need))
    PL_FAIL(PL_EWORKSPACE, "pl_conv2d_nhwc_wgrad: needs %zu scratch bytes (got %zu)", need, scratch_bytes);
  hipStream_t s = (hipStream_t)stream;
  float* slabs = static_cast<float*>(scratch);
  GemmArgs g = {};
  g.A = dy; g.C = splits > 1 ? slabs : dw;
  g.M = (int)Cout; g.N = (int)N; g.K = (int)K; g.lda = (int)Cout; g.ldb = (int)N; g.ldc = (int)N;
  g.split_k = splits; g.arith = implicit ? arith : PL_BF16X6;   // the im2col fallback (odd shapes) stays fp32-grade
  if (plain_tn) {
    g.B = x;
    PL_TRY(launch_gemm_f32(kTN, g, s));
  } else if (implicit) {
    g.B = x;
    g.conv_cin = (int)Cin; g.conv_h = (int)H; g.conv_w = (int)W; g.conv_ho = (int)Ho; g.conv_wo = (int)Wo;
    g.conv_kw = KW; g.conv_stride = stride; g.conv_pad_h = pad; g.conv_pad_w = pad;
    PL_TRY(launch_conv_wgrad(g, s));
  } else {
    // shapes without whole tiles (the stem, the 64-wide layer1 convolutions): explicit im2col + the generic TN GEMM
    float* col = reinterpret_cast<float*>(static_cast<char*>(scratch) + slab_bytes);
    const int64_t tot = K * N;
    if (tot > (int64_t)INT32_MAX * NTHR) PL_FAIL(PL_ESHAPE, "pl_conv2d_nhwc_wgrad: im2col too large");
    hipLaunchKernelGGL(im2col_nhwc_kernel, dim3((unsigned)((tot + NTHR - 1) / NTHR)), dim3(NTHR), 0, s, x, (int)H,
                       (int)W, (int)Cin, (int)Ho, (int)Wo, KH, KW, stride, pad, K, (int)N, (int)N, col);
    PL_CHECK_LAUNCH("im2col_nhwc");
    g.B = col;
    PL_TRY(launch_gemm_f32(kTN, g, s));
  }
  if (splits > 1) return launch_reduce_slabs(slabs, splits, Cout * N, dw, s);
  return PL_OK;
}

extern "C" int pl_maxpool3x3s2_nhwc_bwd(const float* x, const float* dy, int64_t B, int64_t H, int64_t W, int64_t C,
                                        float* dx, void* stream) {
  if (!x || !dy || !dx) PL_FAIL(PL_EINVAL, "pl_maxpool3x3s2_nhwc_bwd: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) PL_FAIL(PL_ESHAPE, "pl_maxpool3x3s2_nhwc_bwd: C %% 4 == 0 needed");
  const int64_t Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t n4 = B * H * W * (C >> 2);
  if (n4 > (int64_t)INT32_MAX * NTHR) PL_FAIL(PL_ESHAPE, "pl_maxpool3x3s2_nhwc_bwd: too large");
  hipLaunchKernelGGL(maxpool3x3s2_nhwc_bwd_kernel, dim3((unsigned)((n4 + NTHR - 1) / NTHR)), dim3(NTHR), 0,
                     (hipStream_t)stream, x, dy, (int)H, (int)W, (int)C, (int)Ho, (int)Wo, n4, dx);
  PL_CHECK_LAUNCH("maxpool3x3s2_nhwc_bwd");
  return PL_OK;
}

// out[c] = sum_r X[r][c]: the bias gradient of a convolution (fixed-order two-stage sum).  cols % 4 == 0 takes a
// strip-parallel pass (256 columns x a row chunk per workgroup, float4 per lane); the generic kernel walks all
// columns in one workgroup per row chunk (3.5 ms for the head's 131072 x 1088 gradient).
static int colsum_wide_chunks(int64_t rows) {
  int64_t rc = (rows + 255) / 256;
  if (rc > 128) rc = 128;
  return rc < 1 ? 1 : (int)rc;
}

extern "C" size_t pl_colsum_scratch_bytes(int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  const int rc = colsum_chunks((int)rows) > colsum_wide_chunks(rows) ? colsum_chunks((int)rows) : colsum_wide_chunks(rows);
  return (size_t)rc * cols * sizeof(float);
}

extern "C" int pl_colsum(const float* X, int64_t rows, int64_t cols, float* out, void* scratch, void* stream) {
  if (!X || !out || !scratch) PL_FAIL(PL_EINVAL, "pl_colsum: null pointer");
  if (rows <= 0 || rows > INT32_MAX || cols <= 0 || cols > INT32_MAX) PL_FAIL(PL_ESHAPE, "pl_colsum: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if ((cols & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
    const int rc = colsum_wide_chunks(rows);
    hipLaunchKernelGGL(colsum_wide_kernel, dim3((unsigned)((cols + 255) / 256), rc), dim3(NTHR), 0, s, X, (int)rows,
                       (int)cols, static_cast<float*>(scratch));
    PL_CHECK_LAUNCH("colsum_wide");
    return launch_reduce_slabs(static_cast<const float*>(scratch), rc, cols, out, s);
  }
  PL_TRY(launch_colsum_partial(X, (int)rows, (int)cols, static_cast<float*>(scratch), s));
  return launch_reduce_slabs(static_cast<const float*>(scratch), colsum_chunks((int)rows), cols, out, s);
}

// column sums of a [rows][cols] tensor given as operand planes (scratch: pl_colsum_scratch_bytes); cols % 4 == 0
extern "C" int pl_colsum_planes(const void* planes, int planes_mode, int64_t rows, int64_t cols, const float* inv_scale,
                                float* out, void* scratch, void* stream) {
  if (!planes || !out || !scratch) PL_FAIL(PL_EINVAL, "pl_colsum_planes: null pointer");
  if (rows <= 0 || rows > INT32_MAX || cols <= 0 || cols > INT32_MAX || (cols & 3)) PL_FAIL(PL_ESHAPE, "pl_colsum_planes: rows=%lld cols=%lld", (long long)rows, (long long)cols);
  PlaneOut po;
  PL_TRY(plane_out_of(planes_mode, const_cast<void*>(planes), rows * cols, 1.0f, nullptr, &po, "pl_colsum_planes"));
  hipStream_t s = (hipStream_t)stream;
  const int rc = colsum_wide_chunks(rows);
  hipLaunchKernelGGL(colsum_wide_planes_kernel, dim3((unsigned)((cols + 255) / 256), rc), dim3(NTHR), 0, s, po.h, po.l, po.kind,
                     planes_mode == PL_F16X3 ? inv_scale : nullptr, (int)rows, (int)cols, static_cast<float*>(scratch));
  PL_CHECK_LAUNCH("colsum_wide_planes");
  return launch_reduce_slabs(static_cast<const float*>(scratch), rc, cols, out, s);
}

extern "C" int pl_upsample2x_zero_nhwc(const float* x, int64_t B, int64_t Hi, int64_t Wi, int64_t C, float* y,
                                       void* stream) {
  if (!x || !y) PL_FAIL(PL_EINVAL, "pl_upsample2x_zero_nhwc: null pointer");
  if (B <= 0 || Hi <= 0 || Wi <= 0 || C <= 0 || (C & 3)) PL_FAIL(PL_ESHAPE, "pl_upsample2x_zero_nhwc: C %% 4 == 0 needed");
  const int64_t n4 = B * 4 * Hi * Wi * (C >> 2);
  if (n4 > (int64_t)INT32_MAX * NTHR) PL_FAIL(PL_ESHAPE, "pl_upsample2x_zero_nhwc: too large");
  hipLaunchKernelGGL(upsample2x_zero_kernel, dim3((unsigned)((n4 + NTHR - 1) / NTHR)), dim3(NTHR), 0,
                     (hipStream_t)stream, x, (int)Hi, (int)Wi, (int)C, n4, y);
  PL_CHECK_LAUNCH("upsample2x_zero");
  return PL_OK;
}

extern "C" int pl_maxpool3x3s2_nhwc_idx(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* y,
                                        unsigned char* idx, void* stream) {
  if (!x || !y || !idx) PL_FAIL(PL_EINVAL, "pl_maxpool3x3s2_nhwc_idx: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) PL_FAIL(PL_ESHAPE, "pl_maxpool3x3s2_nhwc_idx: C %% 4 == 0 needed");
  if (reinterpret_cast<uintptr_t>(idx) & 3) PL_FAIL(PL_EINVAL, "pl_maxpool3x3s2_nhwc_idx: idx not 4-byte aligned");
  const int64_t Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t n4 = B * Ho * Wo * (C >> 2);
  if (n4 > (int64_t)INT32_MAX * NTHR) PL_FAIL(PL_ESHAPE, "pl_maxpool3x3s2_nhwc_idx: too large");
  hipLaunchKernelGGL(maxpool3x3s2_nhwc_idx_kernel, dim3((unsigned)((n4 + NTHR - 1) / NTHR)), dim3(NTHR), 0,
                     (hipStream_t)stream, x, (int)H, (int)W, (int)C, (int)Ho, (int)Wo, n4, y,
                     reinterpret_cast<uchar4*>(idx));
  PL_CHECK_LAUNCH("maxpool3x3s2_nhwc_idx");
  return PL_OK;
}

extern "C" int pl_maxpool3x3s2_nhwc_bwd_idx(const unsigned char* idx, const float* dy, int64_t B, int64_t H, int64_t W,
                                            int64_t C, float* dx, void* stream) {
  if (!idx || !dy || !dx) PL_FAIL(PL_EINVAL, "pl_maxpool3x3s2_nhwc_bwd_idx: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) PL_FAIL(PL_ESHAPE, "pl_maxpool3x3s2_nhwc_bwd_idx: C %% 4 == 0 needed");
  if (reinterpret_cast<uintptr_t>(idx) & 3) PL_FAIL(PL_EINVAL, "pl_maxpool3x3s2_nhwc_bwd_idx: idx not 4-byte aligned");
  const int64_t Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t n4 = B * H * W * (C >> 2);
  if (n4 > (int64_t)INT32_MAX * NTHR) PL_FAIL(PL_ESHAPE, "pl_maxpool3x3s2_nhwc_bwd_idx: too large");
  hipLaunchKernelGGL(maxpool3x3s2_nhwc_bwd_idx_kernel, dim3((unsigned)((n4 + NTHR - 1) / NTHR)), dim3(NTHR), 0,
                     (hipStream_t)stream, reinterpret_cast<const uchar4*>(idx), dy, (int)H, (int)W, (int)C, (int)Ho,
                     (int)Wo, n4, dx);
  PL_CHECK_LAUNCH("maxpool3x3s2_nhwc_bwd_idx");
  return PL_OK;
}
