// Micro-benchmark 2: VALU ops in the shadow of v_mfma_f32_32x32x16_bf16, ALL INDEPENDENT within one
// shadow (16 distinct registers), versus dependent chains.  hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));

// operands: %0..%3 accumulators, %4 = 16 scratch regs (vector), %5 fa, %6 fb, %7 const
#define MFMA(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", %5, %6, %" #acc "\n"
#define S(i) "v_sub_f32 %4[" #i "], %4[" #i "], %7\n"

template <int K, int CHAIN>
__global__ __launch_bounds__(512) void bench(float* out, int iters) {
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  float s[16];
  for (int i = 0; i < 16; ++i) s[i] = (float)(threadIdx.x + i);
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (short)(0x3f80 + threadIdx.x); fb[i] = (short)0x3f80; }
  float c0 = 1.0f;
  for (int it = 0; it < iters; ++it) {
#define OPI(i) "v_sub_f32 %" #i ", %" #i ", %22\n"
    // operand map: 0-3 acc, 4-19 scratch s[0..15], 20 fa, 21 fb, 22 c0
#define M(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", %20, %21, %" #acc "\n"
#define OUTS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), \
             "+v"(s[5]), "+v"(s[6]), "+v"(s[7]), "+v"(s[8]), "+v"(s[9]), "+v"(s[10]), "+v"(s[11]), "+v"(s[12]),  \
             "+v"(s[13]), "+v"(s[14]), "+v"(s[15])
#define INS "v"(fa), "v"(fb), "v"(c0)
    if constexpr (CHAIN == 0) {   // independent: each register touched once per shadow
      if constexpr (K == 4) asm volatile(M(0) OPI(4) OPI(5) OPI(6) OPI(7) M(1) OPI(8) OPI(9) OPI(10) OPI(11)
                                         M(2) OPI(12) OPI(13) OPI(14) OPI(15) M(3) OPI(16) OPI(17) OPI(18) OPI(19) : OUTS : INS);
      if constexpr (K == 8) asm volatile(M(0) OPI(4) OPI(5) OPI(6) OPI(7) OPI(8) OPI(9) OPI(10) OPI(11)
                                         M(1) OPI(12) OPI(13) OPI(14) OPI(15) OPI(16) OPI(17) OPI(18) OPI(19)
                                         M(2) OPI(4) OPI(5) OPI(6) OPI(7) OPI(8) OPI(9) OPI(10) OPI(11)
                                         M(3) OPI(12) OPI(13) OPI(14) OPI(15) OPI(16) OPI(17) OPI(18) OPI(19) : OUTS : INS);
      if constexpr (K == 16) asm volatile(
          M(0) OPI(4) OPI(5) OPI(6) OPI(7) OPI(8) OPI(9) OPI(10) OPI(11) OPI(12) OPI(13) OPI(14) OPI(15) OPI(16) OPI(17) OPI(18) OPI(19)
          M(1) OPI(4) OPI(5) OPI(6) OPI(7) OPI(8) OPI(9) OPI(10) OPI(11) OPI(12) OPI(13) OPI(14) OPI(15) OPI(16) OPI(17) OPI(18) OPI(19)
          M(2) OPI(4) OPI(5) OPI(6) OPI(7) OPI(8) OPI(9) OPI(10) OPI(11) OPI(12) OPI(13) OPI(14) OPI(15) OPI(16) OPI(17) OPI(18) OPI(19)
          M(3) OPI(4) OPI(5) OPI(6) OPI(7) OPI(8) OPI(9) OPI(10) OPI(11) OPI(12) OPI(13) OPI(14) OPI(15) OPI(16) OPI(17) OPI(18) OPI(19)
          : OUTS : INS);
    } else if constexpr (CHAIN == 1) {  // one serial chain: every op depends on the previous one
      if constexpr (K == 4) asm volatile(M(0) OPI(4) OPI(4) OPI(4) OPI(4) M(1) OPI(4) OPI(4) OPI(4) OPI(4)
                                         M(2) OPI(4) OPI(4) OPI(4) OPI(4) M(3) OPI(4) OPI(4) OPI(4) OPI(4) : OUTS : INS);
      if constexpr (K == 8) asm volatile(M(0) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4)
                                         M(1) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4)
                                         M(2) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4)
                                         M(3) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4) OPI(4) : OUTS : INS);
    } else {  // CHAIN == 2: two MFMAs back to back, then 2K VALU ops (clustered instead of interleaved)
      if constexpr (K == 4) asm volatile(M(0) M(1) OPI(4) OPI(5) OPI(6) OPI(7) OPI(8) OPI(9) OPI(10) OPI(11)
                                         M(2) M(3) OPI(12) OPI(13) OPI(14) OPI(15) OPI(16) OPI(17) OPI(18) OPI(19) : OUTS : INS);
      if constexpr (K == 8) asm volatile(M(0) M(1) OPI(4) OPI(5) OPI(6) OPI(7) OPI(8) OPI(9) OPI(10) OPI(11) OPI(12) OPI(13) OPI(14) OPI(15) OPI(16) OPI(17) OPI(18) OPI(19)
                                         M(2) M(3) OPI(4) OPI(5) OPI(6) OPI(7) OPI(8) OPI(9) OPI(10) OPI(11) OPI(12) OPI(13) OPI(14) OPI(15) OPI(16) OPI(17) OPI(18) OPI(19) : OUTS : INS);
    }
  }
  float r = a0[0] + a1[1] + a2[2] + a3[3];
  for (int i = 0; i < 16; ++i) r += s[i];
  if (r == 123.456f) out[0] = r;
}

template <int K, int CHAIN>
double run(int waves_per_simd, float* d_out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int threads = 256 * waves_per_simd;
  hipLaunchKernelGGL((bench<K, CHAIN>), dim3(256), dim3(threads), 0, 0, d_out, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((bench<K, CHAIN>), dim3(256), dim3(threads), 0, 0, d_out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / ((double)iters * 4);
}

int main() {
  float* d_out;
  (void)hipMalloc(&d_out, 64);
  printf("ns per [MFMA + K v_sub_f32] per wave\n");
  printf("independent regs   1w: K4 %.1f K8 %.1f K16 %.1f | 2w: K4 %.1f K8 %.1f K16 %.1f\n", run<4, 0>(1, d_out),
         run<8, 0>(1, d_out), run<16, 0>(1, d_out), run<4, 0>(2, d_out), run<8, 0>(2, d_out), run<16, 0>(2, d_out));
  printf("one serial chain   1w: K4 %.1f K8 %.1f | 2w: K4 %.1f K8 %.1f\n", run<4, 1>(1, d_out), run<8, 1>(1, d_out),
         run<4, 1>(2, d_out), run<8, 1>(2, d_out));
  printf("clustered (2 MFMA then 2K valu) 1w: K4 %.1f K8 %.1f | 2w: K4 %.1f K8 %.1f\n", run<4, 2>(1, d_out),
         run<8, 2>(1, d_out), run<4, 2>(2, d_out), run<8, 2>(2, d_out));
  return 0;
}
