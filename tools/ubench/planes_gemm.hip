// Microbenchmark / correctness harness for the "planes in HBM" GEMM main loop (round 2).
//
// Operands are delivered PRE-SPLIT as bf16 planes ([NPL][rows][cols], row-major) by the producing kernels, the
// GEMM stages them global -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no split VALU, no ds_write)
// and the inner loop is ds_read + MFMA only.  k-strided operands are read with ds_read_b64_tr_b16.
//
//   build:  hipcc --offload-arch=gfx950 -O3 tools/ubench/planes_gemm.hip -o build/planes_gemm
//   run:    build/planes_gemm            (checks all three layouts against a naive fp64 kernel, then times them)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../3d_poseestimation_amd/csrc/gemm_planes16.h"

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)

using namespace plp;

template <bool A_KS, bool B_KS, int MODE, int NLW, int ABL, int NST>
__global__ __launch_bounds__(256 + 64 * NLW) void planes_gemm_kernel(PlanesArgs p, float out_scale) {
  constexpr int NPL = ModeCfg<MODE>::NPL;
  __shared__ __attribute__((aligned(16))) char lds[PlanesCfg<32, NPL, NST>::LDS];
  f32x16 acc[ModeCfg<MODE>::NACC][2][2];
  int m0, n0, slice;
  if (!planes_mainloop<A_KS, B_KS, 32, MODE, NLW, ABL, NST>(p, blockIdx.x, gridDim.x, lds, acc, m0, n0, slice)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  float* C = p.C + (size_t)slice * p.M * p.ldc;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col = n0 + wn * 64 + b * 32 + i;
        float v = acc[0][a][b][r];
        if constexpr (MODE == kF16x3) v = fmaf(acc[1][a][b][r], 1.0f / kF16LoScale, v) * out_scale;
        C[(size_t)row * p.ldc + col] = v;
      }
}

// the 16x16x32 main loop (gemm_planes16.h), plain register stores
template <bool A_KS, bool B_KS, int MODE>
__global__ __launch_bounds__(512) void planes_gemm16_kernel(PlanesArgs p, float out_scale) {
  constexpr int NPL = ModeCfg<MODE>::NPL;
  __shared__ __attribute__((aligned(16))) char lds[PlanesCfg<32, NPL, 3>::LDS];
  f32x4v acc[ModeCfg<MODE>::NACC][4][4];
  int m0, n0, slice;
  if (!planes_mainloop16<A_KS, B_KS, MODE>(p, blockIdx.x, gridDim.x, lds, acc, m0, n0, slice)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float* C = p.C + (size_t)slice * p.M * p.ldc;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + a * 16 + 4 * (lane >> 4) + r;
        const int col = n0 + wn * 64 + b * 16 + (lane & 15);
        float v = acc[0][a][b][r];
        if constexpr (MODE == kF16x3) v = fmaf(acc[1][a][b][r], 1.0f / kF16LoScale, v) * out_scale;
        C[(size_t)row * p.ldc + col] = v;
      }
}

// planes of x (n elements): bf16 modes [npl][n] bf16; f16x3 [2][n] fp16 = {h, l} of scale * x
__global__ void split_kernel(const float* x, unsigned short* planes, size_t n, int mode, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float f = x[i];
  if (mode == kF16x3) {
    const float v = f * scale;
    const _Float16 hh = (_Float16)v;
    const _Float16 ll = (_Float16)((v - (float)hh) * kF16LoScale);
    planes[i] = __builtin_bit_cast(unsigned short, hh);
    planes[n + i] = __builtin_bit_cast(unsigned short, ll);
    return;
  }
  const __bf16 x0 = (__bf16)f;
  planes[i] = __builtin_bit_cast(unsigned short, x0);
  if (mode == kBf16x6) {
    const float r1 = f - (float)x0;
    const __bf16 x1 = (__bf16)r1;
    const float r2 = r1 - (float)x1;
    planes[n + i] = __builtin_bit_cast(unsigned short, x1);
    planes[2 * n + i] = __builtin_bit_cast(unsigned short, (__bf16)r2);
  }
}

// C[m][n] = sum_k A(m,k) B(n,k) in fp64 on the ORIGINAL fp32 values (bf16 mode: on the rounded ones)
__global__ void ref_kernel(const float* A, const float* B, int mode, bool a_ks, bool b_ks, int M, int N, int K, double* C) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  double s = 0;
  for (int k = 0; k < K; ++k) {
    const size_t ia = a_ks ? (size_t)k * M + m : (size_t)m * K + k;
    const size_t ib = b_ks ? (size_t)k * N + n : (size_t)n * K + k;
    float a = A[ia], b = B[ib];
    if (mode == kBf16) { a = (float)(__bf16)a; b = (float)(__bf16)b; }
    s += (double)a * (double)b;
  }
  C[(size_t)m * N + n] = s;
}

template <bool A_KS, bool B_KS, int MODE, int NLW = 4, int ABL = 0, int NST = 3, bool S16 = false>
static void run(const char* name, int M, int N, int K, int splits, bool check, float amag = 1.f, float bmag = 0.03f) {
  if (ABL) check = false;
  constexpr int NPL = ModeCfg<MODE>::NPL;
  const int nthr = 256 + 64 * NLW;
  const size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
  std::vector<float> ha(na), hb(nb);
  srand(1234);
  for (auto& v : ha) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * amag;
  for (auto& v : hb) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * bmag;
  // f16x3: power-of-two scales that put the largest magnitude near 2^12
  auto pow2 = [](float mag) { int e; frexpf(mag, &e); return ldexpf(1.f, 12 - e); };
  const float sa = MODE == kF16x3 ? pow2(amag) : 1.f, sb = MODE == kF16x3 ? pow2(bmag) : 1.f;
  float *da, *db, *dc;
  unsigned short *pa, *pb;
  double* dref;
  CK(hipMalloc(&da, na * 4)); CK(hipMalloc(&db, nb * 4)); CK(hipMalloc(&dc, nc * 4 * splits));
  CK(hipMalloc(&pa, na * 2 * NPL)); CK(hipMalloc(&pb, nb * 2 * NPL)); CK(hipMalloc(&dref, nc * 8));
  CK(hipMemcpy(da, ha.data(), na * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), nb * 4, hipMemcpyHostToDevice));
  split_kernel<<<(na + 255) / 256, 256>>>(da, pa, na, MODE, sa);
  split_kernel<<<(nb + 255) / 256, 256>>>(db, pb, nb, MODE, sb);
  PlanesArgs p = {};
  p.A = (const __bf16*)pa; p.B = (const __bf16*)pb; p.C = dc; p.M = M; p.N = N; p.K = K; p.ldc = N; p.split_k = splits;
  p.lda = A_KS ? M : K; p.ldb = B_KS ? N : K; p.a_plane = na; p.b_plane = nb;
  const float out_scale = 1.f / (sa * sb);
  const int grid = (M / 128) * (N / 128) * splits;
  CK(hipMemset(dc, 0xff, nc * 4 * splits));
  auto launch = [&]() {
    if constexpr (S16) planes_gemm16_kernel<A_KS, B_KS, MODE><<<grid, 512>>>(p, out_scale);
    else planes_gemm_kernel<A_KS, B_KS, MODE, NLW, ABL, NST><<<grid, nthr>>>(p, out_scale);
  };
  launch();
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  const char* mname = MODE == kBf16 ? "bf16  " : MODE == kBf16x6 ? "bf16x6" : "f16x3 ";
  if (check) {
    ref_kernel<<<dim3((N + 255) / 256, M), 256>>>(da, db, MODE, A_KS, B_KS, M, N, K, dref);
    CK(hipDeviceSynchronize());
    std::vector<float> hc(nc * splits);
    std::vector<double> hr(nc);
    CK(hipMemcpy(hc.data(), dc, nc * 4 * splits, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), dref, nc * 8, hipMemcpyDeviceToHost));
    double maxref = 0, maxerr = 0, sq = 0, sqr = 0; size_t bad = 0, worst = 0;
    for (size_t i = 0; i < nc; ++i) {
      double v = 0;
      for (int s = 0; s < splits; ++s) v += hc[(size_t)s * nc + i];
      maxref = fmax(maxref, fabs(hr[i]));
      const double e = fabs(v - hr[i]);
      sq += e * e; sqr += hr[i] * hr[i];
      if (!(e <= maxerr)) { maxerr = e; worst = i; }
      if (!(e == e)) ++bad;
    }
    printf("%-2s %s %s lw%d st%d %dx%dx%d s%d mag %g/%g: max|ref|=%.4g max err=%.3g (rel %.3g) rms rel %.3g nan=%zu %s\n", name, mname,
           S16 ? "16x16x32" : "32x32x16", NLW, NST, M, N, K, splits, amag, bmag, maxref, maxerr, maxerr / maxref, sqrt(sq / sqr), bad,
           (bad == 0 && maxerr / maxref < 3e-6) ? "OK" : "FAIL");
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 5; ++it) launch();
  const int iters = 50;
  CK(hipEventRecord(e0));
  for (int it = 0; it < iters; ++it) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  printf("%-2s %s %s lw%d st%d abl%d %dx%dx%d s%d grid=%d: %.2f us  %.1f TF algorithmic (%.1f TF issued)\n", name, mname,
         S16 ? "16x16x32" : "32x32x16", NLW, NST,
         ABL, M, N, K, splits, grid, us, 2.0 * M * N * K / us * 1e-6, 2.0 * M * N * K / us * 1e-6 * ModeCfg<MODE>::NPROD);
  CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dc)); CK(hipFree(pa)); CK(hipFree(pb)); CK(hipFree(dref));
}

int main(int argc, char** argv) {
  const bool check = argc < 2 || atoi(argv[1]) != 0;
  if (argc > 2 && atoi(argv[2]) == 1) {
    // MFMA shape question only (timing builds): 32x32x16 vs the same FLOPs as 16x16x32, with and without the DMA
    for (int rep = 0; rep < 2; ++rep) {
      run<false, false, kF16x3, 4, 0, 3>("NT", 4096, 1024, 1024, 1, false);
      run<false, false, kF16x3, 4, 4, 3>("NT", 4096, 1024, 1024, 1, false);
      run<false, false, kF16x3, 4, 2, 3>("NT", 4096, 1024, 1024, 1, false);
      run<false, false, kF16x3, 4, 5, 3>("NT", 4096, 1024, 1024, 1, false);
    }
    return 0;
  }
  if (argc > 2 && atoi(argv[2]) == 2) {
    // the 16x16x32 main loop: every layout against fp64 (small first), then the lifter's shapes beside the 32x32x16 loop
    run<false, false, kF16x3, 4, 0, 3, true>("NT", 128, 128, 32, 1, check);
    run<false, false, kF16x3, 4, 0, 3, true>("NT", 128, 128, 64, 1, check);
    run<false, false, kF16x3, 4, 0, 3, true>("NT", 128, 128, 96, 1, check);
    run<false, false, kF16x3, 4, 0, 3, true>("NT", 256, 256, 128, 1, check);
    run<false, true, kF16x3, 4, 0, 3, true>("NN", 256, 256, 128, 1, check);
    run<true, true, kF16x3, 4, 0, 3, true>("TN", 256, 256, 256, 2, check);
    run<false, false, kBf16, 4, 0, 3, true>("NT", 256, 256, 128, 1, check);
    run<false, true, kBf16, 4, 0, 3, true>("NN", 256, 256, 128, 1, check);
    run<true, true, kBf16, 4, 0, 3, true>("TN", 256, 256, 256, 2, check);
    for (int rep = 0; rep < 2; ++rep) {
      run<false, false, kF16x3, 4, 0, 3>("NT", 4096, 1024, 1024, 1, false);
      run<false, false, kF16x3, 4, 0, 3, true>("NT", 4096, 1024, 1024, 1, check && rep == 0);
      run<false, true, kF16x3, 4, 0, 3>("NN", 4096, 1024, 1024, 1, false);
      run<false, true, kF16x3, 4, 0, 3, true>("NN", 4096, 1024, 1024, 1, check && rep == 0);
      run<true, true, kF16x3, 4, 0, 3>("TN", 1024, 1024, 4096, 4, false);
      run<true, true, kF16x3, 4, 0, 3, true>("TN", 1024, 1024, 4096, 4, check && rep == 0);
      run<false, false, kBf16, 4, 0, 3>("NT", 4096, 1024, 1024, 1, false);
      run<false, false, kBf16, 4, 0, 3, true>("NT", 4096, 1024, 1024, 1, check && rep == 0);
    }
    return 0;
  }
  // small shapes first (a wrong kernel should fail here, quickly)
  run<false, false, kF16x3>("NT", 256, 256, 128, 1, check);
  run<false, true, kF16x3>("NN", 256, 256, 128, 1, check);
  run<true, true, kF16x3>("TN", 256, 256, 256, 2, check);
  run<false, false, kF16x3, 4, 0, 4>("NT", 256, 256, 256, 1, check);
  run<false, false, kF16x3, 4, 0, 4>("NT", 128, 128, 32, 1, check);
  run<false, false, kF16x3, 4, 0, 4>("NT", 128, 128, 64, 1, check);
  run<false, false, kF16x3, 4, 0, 4>("NT", 128, 128, 96, 1, check);
  run<false, false, kF16x3, 4, 0, 4>("NT", 128, 128, 128, 1, check);
  run<false, false, kF16x3, 0, 0, 3>("NT", 256, 256, 128, 1, check);
  run<false, false, kBf16x6>("NT", 256, 256, 128, 1, check);
  run<false, false, kBf16>("NT", 256, 256, 128, 1, check);
  // tiny gradients x ordinary weights: the per-tensor scale keeps h in fp16's normal range
  run<false, true, kF16x3>("NN", 256, 256, 128, 1, check, 3e-7f, 0.03f);
  run<true, true, kF16x3>("TN", 256, 256, 256, 2, check, 3e-7f, 5.f);
  // the lifter's shapes at B = 4096
  run<false, false, kF16x3, 4, 0, 3>("NT", 4096, 1024, 1024, 1, check);
  run<false, false, kF16x3, 4, 0, 4>("NT", 4096, 1024, 1024, 1, false);
  run<false, false, kF16x3, 0, 0, 3>("NT", 4096, 1024, 1024, 1, false);
  run<false, true, kF16x3, 4, 0, 3>("NN", 4096, 1024, 1024, 1, check);
  run<false, true, kF16x3, 4, 0, 4>("NN", 4096, 1024, 1024, 1, false);
  run<true, true, kF16x3, 4, 0, 3>("TN", 1024, 1024, 4096, 4, check);
  run<true, true, kF16x3, 4, 0, 4>("TN", 1024, 1024, 4096, 4, false);
  run<false, false, kBf16, 4, 0, 3>("NT", 4096, 1024, 1024, 1, check);
  run<false, false, kBf16, 4, 0, 4>("NT", 4096, 1024, 1024, 1, false);
  run<false, true, kBf16, 4, 0, 4>("NN", 4096, 1024, 1024, 1, check);
  run<true, true, kBf16, 4, 0, 4>("TN", 1024, 1024, 4096, 4, check);
  run<false, false, kBf16x6, 4, 0, 3>("NT", 4096, 1024, 1024, 1, false);
  // where the time goes (timing-only builds: wrong results by construction)
  run<false, false, kF16x3, 4, 1, 4>("NT", 4096, 1024, 1024, 1, false);
  run<false, false, kF16x3, 4, 2, 4>("NT", 4096, 1024, 1024, 1, false);
  run<false, false, kF16x3, 4, 3, 4>("NT", 4096, 1024, 1024, 1, false);
  run<true, true, kF16x3, 4, 1, 4>("TN", 1024, 1024, 4096, 4, false);
  run<true, true, kF16x3, 4, 2, 4>("TN", 1024, 1024, 4096, 4, false);
  return 0;
}
