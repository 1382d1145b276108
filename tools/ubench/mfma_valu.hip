// Micro-benchmark: how many VALU instructions fit in the shadow of one v_mfma_f32_32x32x16_bf16 on
// gfx950, per VALU opcode, from the same wave (1 wave/SIMD) and from a sibling wave (2 waves/SIMD)?
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_valu.hip -o tools/ubench/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", %8, %9, %" #acc "\n"
// VALU flavours; operands %4..%7 are four scratch VGPRs, %10 a constant VGPR
#define OP_AND(d) "v_and_b32 %" #d ", 0xffff0000, %" #d "\n"
#define OP_SUB(d) "v_sub_f32 %" #d ", %" #d ", %10\n"
#define OP_PERM(d) "v_perm_b32 %" #d ", %" #d ", %10, %11\n"
#define OP_CVT(d) "v_cvt_pk_bf16_f32 %" #d ", %" #d ", %10\n"
#define OP_DOT2(d) "v_dot2_f32_bf16 %" #d ", %10, %11, %" #d "\n"
#define OP_LSHL(d) "v_lshlrev_b32 %" #d ", 16, %" #d "\n"

#define BODY(OP, K)                                                                                    \
  asm volatile(MFMA(0) K(OP) MFMA(1) K(OP) MFMA(2) K(OP) MFMA(3) K(OP)                                 \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)        \
               : "v"(fa), "v"(fb), "v"(c0), "v"(c1));

#define BODYA(OP, K)                                                                                   \
  asm volatile(MFMA(0) K(OP) MFMA(1) K(OP) MFMA(2) K(OP) MFMA(3) K(OP)                                 \
               : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)        \
               : "v"(fa), "v"(fb), "v"(c0), "v"(c1));

#define K0(OP)
#define K2(OP) OP(4) OP(5)
#define K4(OP) OP(4) OP(5) OP(6) OP(7)
#define K6(OP) K4(OP) K2(OP)
#define K8(OP) K4(OP) K4(OP)
#define K12(OP) K8(OP) K4(OP)
#define K16(OP) K8(OP) K8(OP)

template <int OPK, int K, bool AG>
__global__ __launch_bounds__(512) void bench(float* out, int iters, int valu_only_waves) {
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  unsigned s0 = threadIdx.x, s1 = threadIdx.x * 3, s2 = 7, s3 = 11;
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (short)(0x3f80 + threadIdx.x); fb[i] = (short)0x3f80; }
  unsigned c0 = 0x3f800000u, c1 = 0x07060302u;
  const int wave = threadIdx.x >> 6;
  // waves >= 4 (second wave of each SIMD) run VALU only when valu_only_waves is set
  const bool valu_only = valu_only_waves && wave >= 4;
  for (int it = 0; it < iters; ++it) {
    if (!valu_only) {
#define CASE(OPN, OPM)                                  \
      if constexpr (OPK == OPN && !AG) {                \
        if constexpr (K == 0) { BODY(OPM, K0) }         \
        else if constexpr (K == 2) { BODY(OPM, K2) }    \
        else if constexpr (K == 4) { BODY(OPM, K4) }    \
        else if constexpr (K == 6) { BODY(OPM, K6) }    \
        else if constexpr (K == 8) { BODY(OPM, K8) }    \
        else if constexpr (K == 12) { BODY(OPM, K12) }  \
        else { BODY(OPM, K16) }                         \
      }                                                 \
      if constexpr (OPK == OPN && AG) {                 \
        if constexpr (K == 0) { BODYA(OPM, K0) }        \
        else if constexpr (K == 2) { BODYA(OPM, K2) }   \
        else if constexpr (K == 4) { BODYA(OPM, K4) }   \
        else if constexpr (K == 6) { BODYA(OPM, K6) }   \
        else if constexpr (K == 8) { BODYA(OPM, K8) }   \
        else if constexpr (K == 12) { BODYA(OPM, K12) } \
        else { BODYA(OPM, K16) }                        \
      }
      CASE(0, OP_AND) CASE(1, OP_SUB) CASE(2, OP_PERM) CASE(3, OP_CVT) CASE(4, OP_LSHL) CASE(5, OP_DOT2)
    } else {
      // the same number of VALU ops as the MFMA waves would have woven in, no MFMAs
#define VCASE(OPN, OPM)                                                                                 \
      if constexpr (OPK == OPN) {                                                                       \
        asm volatile(K16(OPM) K16(OPM) : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(a0)               \
                     : "v"(c0), "v"(c1), "v"(c0), "v"(c1), "v"(fa), "v"(fb), "v"(c0), "v"(c1));         \
      }
      // operand numbering differs in this asm: remap through a dedicated macro set below
    }
  }
  float r = a0[0] + a1[1] + a2[2] + a3[3] + (float)(s0 ^ s1 ^ s2 ^ s3);
  if (r == 123.456f) out[0] = r;
}

template <int OPK, int K, bool AG>
double run(int waves_per_simd, int iters, float* d_out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 256 * waves_per_simd;
  hipLaunchKernelGGL((bench<OPK, K, AG>), dim3(256), dim3(threads), 0, 0, d_out, 100, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((bench<OPK, K, AG>), dim3(256), dim3(threads), 0, 0, d_out, iters, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / ((double)iters * 4);   // ns per MFMA (+K VALU) per wave
}

template <int OPK, bool AG>
void sweep(const char* name, float* d_out) {
  const int iters = 20000;
  printf("%-18s acc in %s", name, AG ? "AGPR" : "VGPR");
  printf(" 1w/SIMD ns per [MFMA+K valu]: K0 %.1f K2 %.1f K4 %.1f K6 %.1f K8 %.1f K12 %.1f K16 %.1f |", run<OPK, 0, AG>(1, iters, d_out),
         run<OPK, 2, AG>(1, iters, d_out), run<OPK, 4, AG>(1, iters, d_out), run<OPK, 6, AG>(1, iters, d_out),
         run<OPK, 8, AG>(1, iters, d_out), run<OPK, 12, AG>(1, iters, d_out), run<OPK, 16, AG>(1, iters, d_out));
  printf(" 2w/SIMD: K0 %.1f K4 %.1f K8 %.1f K16 %.1f\n", run<OPK, 0, AG>(2, iters, d_out), run<OPK, 4, AG>(2, iters, d_out),
         run<OPK, 8, AG>(2, iters, d_out), run<OPK, 16, AG>(2, iters, d_out));
}

int main() {
  float* d_out;
  hipMalloc(&d_out, 64);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s clock %d kHz, CUs %d\n", p.gcnArchName, p.clockRate, p.multiProcessorCount);
  sweep<0, false>("v_and_b32", d_out);
  sweep<1, false>("v_sub_f32", d_out);
  sweep<2, false>("v_perm_b32", d_out);
  sweep<3, false>("v_cvt_pk_bf16_f32", d_out);
  sweep<5, false>("v_dot2_f32_bf16", d_out);
  hipFree(d_out);
  return 0;
}
