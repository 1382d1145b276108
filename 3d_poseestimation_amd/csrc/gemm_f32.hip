// fp32-in / fp32-accumulate GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces the ATen addmm calls behind nn.Linear forward/backward on the lifter path
// (reference phase1_lifting/baselineModel.py:33,39,90,100 and their autograd).  fp32
// operands are kept because the parity gate (1e-3 mm MPJPE against the reference's
// CPU forward) is 1000x tighter than bf16 round-off.
//
// Structure (MI355X-first, not a warp-tiling port):
//   * 128x128 output tile per 256-thread workgroup = 4 wavefronts in a 2x2 grid, each
//     wavefront owning 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).
//     B=4096 x H=1024 gives exactly 256 tiles: one per CU, one wave per SIMD.
//   * K is walked in 32-wide tiles, double-buffered in LDS; the next tile's global loads
//     are issued into registers before the current tile's MFMAs and written to LDS after
//     them, so HBM/L2 latency hides under 64 MFMAs (4096 cycles) per wave.
//   * operands are staged in one of two LDS images:
//       k-contiguous ("KC", source rows are K-major):  [128][32+4]  read with ds_read_b128
//       k-strided    ("KS", source rows are M/N-major): [32][128+4] read with ds_read_b32
//     The 32x32x2 MFMA takes A[i][k], B[k][j] with i/j = lane&31 and k = lane>>5.  Any
//     bijection of k between the two operands is a valid contraction order, so a lane
//     reads FOUR consecutive k (4h..4h+3) of its row with one b128 and feeds them to four
//     consecutive MFMAs; the KS image serves the same k = 8c + 4h + j with b32 reads.
//     Row strides 36 / 132 floats keep both read patterns bank-conflict free.
//   * epilogue, all in registers: bias, residual-gradient addend, eval-mode BN fold +
//     ReLU + skip, and training-mode BatchNorm partial statistics (column sum and M2 per
//     64-row group, merged later with Chan's formula -- no E[z^2]-E[z]^2 cancellation).
//   * blockIdx -> (K slice, tile) map gives each XCD (private 4 MiB L2) a contiguous band of
//     M-tiles of one K slice.
#include <stdlib.h>
#include <type_traits>
#include <vector>

#include "gemm_epilogue.h"
#include "pl_internal.h"

namespace pl {

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NTHR = 256;
constexpr int KC_LD = BK + 4;             // 36 floats = 9 x 16 B
constexpr int KS_LD = 128 + 4;            // 132 floats = 33 x 16 B
constexpr int OP_FLOATS = 128 * KC_LD;    // 4608 >= 32*132
constexpr int LDS_FLOATS = 4 * OP_FLOATS; // 2 operands x 2 buffers = 73,728 B

// Global -> registers: 4 float4 per thread per operand tile, held in NAMED registers (a
// float4[4] passed by reference ends up in scratch memory under hipcc once scheduling
// barriers are present).  EDGE=false is the hot path: whole tiles, 16-byte aligned rows, no
// guards -- eight back-to-back global_load_dwordx4 per thread and K tile.
struct Stage { float4 v0, v1, v2, v3; };   // NT = 256 threads use all four, NT = 512 the first two

template <bool KS, bool EDGE>
__device__ __forceinline__ float4 load_one(const float* __restrict__ base, int ld, int r0, int R,
                                           int k0, int Kend, bool vec_ok, int idx) {
  if (!KS) {
    const int gr = r0 + (idx >> 3);
    const int gk = k0 + (idx & 7) * 4;
    const float* p = base + (size_t)gr * ld + gk;
    if (!EDGE || (vec_ok && gr < R && gk + 3 < Kend)) return *reinterpret_cast<const float4*>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gr < R) {
      if (gk + 0 < Kend) v.x = p[0];
      if (gk + 1 < Kend) v.y = p[1];
      if (gk + 2 < Kend) v.z = p[2];
      if (gk + 3 < Kend) v.w = p[3];
    }
    return v;
  } else {
    const int gk = k0 + (idx >> 5);
    const int gr = r0 + (idx & 31) * 4;
    const float* p = base + (size_t)gk * ld + gr;
    if (!EDGE || (vec_ok && gk < Kend && gr + 3 < R)) return *reinterpret_cast<const float4*>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gk < Kend) {
      if (gr + 0 < R) v.x = p[0];
      if (gr + 1 < R) v.y = p[1];
      if (gr + 2 < R) v.z = p[2];
      if (gr + 3 < R) v.w = p[3];
    }
    return v;
  }
}

template <bool KS, bool EDGE, int NT>
__device__ __forceinline__ Stage load_tile(const float* __restrict__ base, int ld, int r0, int R,
                                           int k0, int Kend, bool vec_ok, int tid) {
  Stage s;
  if (!EDGE) {
    // wave-uniform tile origin (scalar registers) + per-thread 32-bit element offsets: the loads
    // become `global_load_dwordx4 v, voff, s[base]` with no per-tile 64-bit address arithmetic
    const float* t = KS ? base + (size_t)k0 * ld + r0 : base + (size_t)r0 * ld + k0;
    const int o0 = KS ? (tid >> 5) * ld + (tid & 31) * 4 : (tid >> 3) * ld + (tid & 7) * 4;
    const int step = KS ? (NT / 32) * ld : (NT / 8) * ld;   // NT threads further on: +NT/32 k rows / +NT/8 rows
    s.v0 = *reinterpret_cast<const float4*>(t + o0);
    s.v1 = *reinterpret_cast<const float4*>(t + o0 + step);
    if (NT == 256) {
      s.v2 = *reinterpret_cast<const float4*>(t + o0 + 2 * step);
      s.v3 = *reinterpret_cast<const float4*>(t + o0 + 3 * step);
    }
    return s;
  }
  s.v0 = load_one<KS, EDGE>(base, ld, r0, R, k0, Kend, vec_ok, tid);
  s.v1 = load_one<KS, EDGE>(base, ld, r0, R, k0, Kend, vec_ok, tid + NT);
  if (NT == 256) {
    s.v2 = load_one<KS, EDGE>(base, ld, r0, R, k0, Kend, vec_ok, tid + 2 * NT);
    s.v3 = load_one<KS, EDGE>(base, ld, r0, R, k0, Kend, vec_ok, tid + 3 * NT);
  }
  return s;
}

template <bool KS>
__device__ __forceinline__ void store_one(float* __restrict__ s, int idx, float4 v) {
  if (!KS)
    *reinterpret_cast<float4*>(s + (idx >> 3) * KC_LD + (idx & 7) * 4) = v;
  else
    *reinterpret_cast<float4*>(s + (idx >> 5) * KS_LD + (idx & 31) * 4) = v;
}

template <bool KS, int NT>
__device__ __forceinline__ void store_tile(float* __restrict__ s, int tid, const Stage& g) {
  store_one<KS>(s, tid, g.v0);
  store_one<KS>(s, tid + NT, g.v1);
  if (NT == 256) {
    store_one<KS>(s, tid + 2 * NT, g.v2);
    store_one<KS>(s, tid + 3 * NT, g.v3);
  }
}

// Fragment fetch for one 8-wide k chunk: f[t][j] feeds MFMA j of 32x32 tile t.
template <bool KS, int NTILE>
__device__ __forceinline__ void read_frag(const float* __restrict__ s, int row0, int c8, int i,
                                          int h, float (&f)[NTILE][4]) {
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    if (!KS) {
      const float4 v =
          *reinterpret_cast<const float4*>(s + (row0 + t * 32 + i) * KC_LD + c8 * 8 + 4 * h);
      f[t][0] = v.x; f[t][1] = v.y; f[t][2] = v.z; f[t][3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) f[t][j] = s[(c8 * 8 + 4 * h + j) * KS_LD + row0 + t * 32 + i];
    }
  }
}

// bf16 arithmetic on the same fp32 LDS images (PL_BF16 mode): a lane of v_mfma_f32_32x32x16_bf16
// holds A[row = lane&31][k = 8h .. 8h+7]; the 8 floats are fetched (2 x b128 from a KC image,
// 8 x b32 from a KS image) and rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on the way to the MFMA.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool KS>
__device__ __forceinline__ bf16x8 read_frag_bf16(const float* __restrict__ s, int row, int s16, int h) {
  float f[8];
  if (!KS) {
    const float4 u = *reinterpret_cast<const float4*>(s + row * KC_LD + s16 * 16 + 8 * h);
    const float4 v = *reinterpret_cast<const float4*>(s + row * KC_LD + s16 * 16 + 8 * h + 4);
    f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w; f[4] = v.x; f[5] = v.y; f[6] = v.z; f[7] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = s[(s16 * 16 + 8 * h + j) * KS_LD + row];
  }
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)f[j];
  return r;
}

// PL_BF16X6: fp32-grade products from bf16 MFMAs.  Each fp32 operand is split into three bf16
// pieces (8 + 8 + 8 = 24 mantissa bits: x = x0 + x1 + x2 exactly), and a*b is accumulated as the six
// products a0b0, a0b1, a1b0, a0b2, a1b1, a2b0 (every bf16 x bf16 product is exact in the fp32
// accumulator; the three dropped terms are below 2^-24 |ab|).  Six MFMAs at 16x the fp32 rate.
template <bool KS>
__device__ __forceinline__ void read_frag_raw(const float* __restrict__ s, int row, int s16, int h,
                                              float (&f)[8]) {
  if (!KS) {
    const float4 u = *reinterpret_cast<const float4*>(s + row * KC_LD + s16 * 16 + 8 * h);
    const float4 v = *reinterpret_cast<const float4*>(s + row * KC_LD + s16 * 16 + 8 * h + 4);
    f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w; f[4] = v.x; f[5] = v.y; f[6] = v.z; f[7] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = s[(s16 * 16 + 8 * h + j) * KS_LD + row];
  }
}

__device__ __forceinline__ void split3(const float (&f)[8], bf16x8 (&p)[3]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 x0 = (__bf16)f[j];
    const float r1 = f[j] - (float)x0;
    const __bf16 x1 = (__bf16)r1;
    const float r2 = r1 - (float)x1;
    p[0][j] = x0; p[1][j] = x1; p[2][j] = (__bf16)r2;
  }
}

// Epilogue for SHORT-K problems (the 1x1 convolutions of the conv path: K = 64..512, two to sixteen K tiles
// per 64 KB of output): there the dword-per-lane stores and residual loads of gemm_epilogue ARE the kernel
// (layer1 conv3 at B = 64: 600 MB moved in 590 us).  Whole tiles only.  bias / scale / shift / first ReLU in
// registers, the wave's 64x64 block through its own 16 KB of LDS, then 16 x (16-byte residual load, add,
// second ReLU, 16-byte store): 4 rows x 256 B per instruction instead of 2 rows x 128 B.
__device__ __forceinline__ void gemm_epilogue_vec(const GemmArgs& p, float* __restrict__ C, f32x16 (&acc)[2][2],
                                                  const int m0, const int n0, const int wm, const int wn,
                                                  const int i, const int h, float* __restrict__ ldsw) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int col = n0 + wn * 64 + b * 32 + i;
    const float bias = p.bias ? p.bias[col] : 0.f;
    const float scale = p.col_scale ? p.col_scale[col] : 1.f, shift = p.col_scale ? p.col_shift[col] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[a][b][r] + bias;
        if (p.col_scale) v = fmaf(v, scale, shift);
        if (p.relu == 1) v = fmaxf(v, 0.f);
        ldsw[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + b * 32 + i] = v;
      }
  }
  const int lane = h * 32 + i;
  const int lr = lane >> 4, lc = (lane & 15) * 4;
  const size_t o0 = (size_t)(m0 + wm * 64 + lr) * p.ldc + n0 + wn * 64 + lc;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    float4 v = *reinterpret_cast<const float4*>(ldsw + (it * 4 + lr) * 64 + lc);
    const size_t o = o0 + (size_t)it * 4 * p.ldc;
    if (p.resid) {
      const float4 q = *reinterpret_cast<const float4*>(p.resid + o);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    if (p.relu == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4*>(C + o) = v;
  }
}

__device__ __forceinline__ bool epilogue_vec_ok(const GemmArgs& p, const float* C) {
  return p.K <= 512 && p.split_k <= 1 && !p.stat_sum && !p.addend && (p.ldc & 3) == 0 &&
         ((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(p.resid)) & 15) == 0;
}

// NW = 4 wavefronts (2x2, 64x64 per wave) is what ships.  NW = 8 (2x4, 64x32 per wave, 512 threads) is
// kept compilable: it was built for the VALU-heavy PL_BF16X6 arithmetic on the theory that two
// waves per SIMD would share the instruction issue port, and measured SLOWER (forward 62 vs 59 us; the
// dual launch 127 vs 106 us because its two halves no longer fit one CU together).
template <bool A_KS, bool B_KS, bool EDGE, int AR = 0, int NW = 4>
__device__ __forceinline__ void gemm_body(const GemmArgs& p, const int block_id, const int nwork,
                                          float* __restrict__ lds) {
  constexpr int NT = 64 * NW;               // threads
  constexpr int NB = NW == 4 ? 2 : 1;       // 32-column MFMA tiles per wave
  constexpr int WCOLS = 32 * NB;            // columns per wave
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = NW == 4 ? wave >> 1 : wave >> 2, wn = NW == 4 ? wave & 1 : wave & 3;

  // ---- tile id: XCD-aware remap (blocks b and b+8 share an XCD) ------------------------
  // The grid is 1-D over (K slice, tile), slice-major.  Workgroups b and b+8 share an XCD, so
  // the remap hands each XCD a contiguous run of work items: a band of M tiles x all N tiles
  // of ONE K slice (its operand panels then sit in that XCD's L2 once; a first version mapped
  // slices to blockIdx.z and every XCD streamed ALL of K: 4.5x the algorithmic bytes).
  const int tiles_n = (p.N + BN - 1) / BN;
  const int splits = p.split_k > 1 ? p.split_k : 1;
  const int ntiles = nwork / splits;
  int w = block_id;
  if ((nwork & 7) == 0) w = (w & 7) * (nwork >> 3) + (w >> 3);
  const int slice = w / ntiles;
  const int t = w - slice * ntiles;
  const int m0 = (t / tiles_n) * BM;
  const int n0 = (t % tiles_n) * BN;

  // ---- K range of this split -------------------------------------------------------------
  int kbeg = 0, kend = p.K;
  float* C = p.C;
  if (p.split_k > 1) {
    const int per = (((p.K + p.split_k - 1) / p.split_k) + BK - 1) / BK * BK;
    kbeg = slice * per;
    kend = min(p.K, kbeg + per);
    C += (size_t)slice * p.M * p.ldc;
  }
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

  const bool a_vec = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
  const bool b_vec = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0);

  f32x16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- main loop -------------------------------------------------------------------------
  // Three-stage operand pipeline + double-buffered fragments:
  //   * tile kt is consumed from LDS buffer kt&1 while tile kt+1 (already in registers) is
  //     written to the other buffer and tile kt+2 is in flight from L2/HBM;
  //   * the fragments of 8-wide k chunk c+1 are fetched from LDS while the 16 MFMAs of chunk
  //     c run (hipcc on its own issues them one MFMA ahead and exposes the ~300-cycle LDS
  //     latency four times per tile: 72 % -> MFMA-bound);
  //   * the one barrier per tile sits between chunk 2 and chunk 3: by then every wave has
  //     fetched its last fragments of this buffer (so the next step may overwrite it) and
  //     has long since written the other buffer (so chunk 3 can prefetch the next tile's
  //     first fragments from it) -- the MFMA pipe never drains at a tile boundary.
  float fa[2][2][4], fb[2][NB][4];
#define PL_FRAGS(set, buf, c8)                                                  \
  do {                                                                          \
    read_frag<A_KS, 2>((buf), wm * 64, (c8), i, h, fa[set]);                    \
    read_frag<B_KS, NB>((buf) + OP_FLOATS, wn * WCOLS, (c8), i, h, fb[set]);    \
  } while (0)
#define PL_MFMAS(set)                                                                           \
  do {                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                               \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                               \
    _Pragma("unroll") for (int b = 0; b < NB; ++b)                                              \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][a][j], fb[set][b][j], acc[a][b], 0, 0, 0); \
  } while (0)

  // Instruction mix of one steady-state step: 64 MFMAs (64 cycles each on the matrix pipe),
  // 8 ds_write_b128, 8 global_load_dwordx4 and the fragment ds_reads.  Left in a lump between
  // MFMA groups they idle the pipe ~700 cycles per tile; sched_group_barrier threads ONE of
  // them into each MFMA shadow instead.
  Stage ra, rb;
  constexpr int NFR = (A_KS ? 8 : 2) + (B_KS ? 4 * NB : NB);   // ds_read instructions per fragment set
#define PL_SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
  auto step = [&](const int kt, auto do_store, auto do_load, auto has_next) {
    const float* cur = lds + (kt & 1) * 2 * OP_FLOATS;
    float* nxt = lds + ((kt + 1) & 1) * 2 * OP_FLOATS;
    if (do_store.value) {
      store_tile<A_KS, NT>(nxt, tid, ra);
      store_tile<B_KS, NT>(nxt + OP_FLOATS, tid, rb);
    }
    if (do_load.value) {
      const int k0 = kbeg + (kt + 2) * BK;
      ra = load_tile<A_KS, EDGE, NT>(p.A, p.lda, m0, p.M, k0, kend, a_vec, tid);
      rb = load_tile<B_KS, EDGE, NT>(p.B, p.ldb, n0, p.N, k0, kend, b_vec, tid);
    }
    PL_FRAGS(1, cur, 1);
    PL_MFMAS(0);
    PL_FRAGS(0, cur, 2);
    PL_MFMAS(1);
    PL_FRAGS(1, cur, 3);
    PL_MFMAS(0);
    // ---- schedule of the region above (one basic block) ----
    PL_SGB(0x100, NFR);                                     // fragments of chunk 1 first
    if (do_store.value) {
#pragma unroll
      for (int q = 0; q < 4 * NB; ++q) { PL_SGB(0x008, 1); PL_SGB(0x200, 1); }
    } else {
      PL_SGB(0x008, 4 * NB);
    }
    if (do_load.value) {
#pragma unroll
      for (int q = 0; q < 4 * NB; ++q) { PL_SGB(0x008, 1); PL_SGB(0x020, 1); }
    } else {
      PL_SGB(0x008, 4 * NB);
    }
    PL_SGB(0x100, NFR); PL_SGB(0x008, 8 * NB);
    PL_SGB(0x100, NFR); PL_SGB(0x008, 8 * NB);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (has_next.value) PL_FRAGS(0, nxt, 0);
    PL_MFMAS(1);
    PL_SGB(0x100, NFR); PL_SGB(0x008, 8 * NB);
    __builtin_amdgcn_sched_barrier(0);
  };
  // PL_BF16 variant of one pipeline step: the 32-wide fp32 tile is two 16-deep bf16 MFMA steps
  // (2 x 4 MFMAs of 32 cycles: this loop is bound by staging bytes from L2, not by the pipe).
  bf16x8 ba[2][2], bb[2][NB];
#define PL_FRAGS_BF(set, buf, s16)                                                   \
  do {                                                                               \
    _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2)                                 \
      ba[set][t2] = read_frag_bf16<A_KS>((buf), wm * 64 + t2 * 32 + i, (s16), h);    \
    _Pragma("unroll") for (int t2 = 0; t2 < NB; ++t2)                                \
      bb[set][t2] = read_frag_bf16<B_KS>((buf) + OP_FLOATS, wn * WCOLS + t2 * 32 + i, (s16), h); \
  } while (0)
#define PL_MFMAS_BF(set)                                                                    \
  do {                                                                                      \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                           \
    _Pragma("unroll") for (int b = 0; b < NB; ++b)                                          \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba[set][a], bb[set][b], acc[a][b], 0, 0, 0); \
  } while (0)
  auto step_bf = [&](const int kt, auto do_store, auto do_load, auto has_next) {
    const float* cur = lds + (kt & 1) * 2 * OP_FLOATS;
    float* nxt = lds + ((kt + 1) & 1) * 2 * OP_FLOATS;
    if (do_store.value) {
      store_tile<A_KS, NT>(nxt, tid, ra);
      store_tile<B_KS, NT>(nxt + OP_FLOATS, tid, rb);
    }
    if (do_load.value) {
      const int k0 = kbeg + (kt + 2) * BK;
      ra = load_tile<A_KS, EDGE, NT>(p.A, p.lda, m0, p.M, k0, kend, a_vec, tid);
      rb = load_tile<B_KS, EDGE, NT>(p.B, p.ldb, n0, p.N, k0, kend, b_vec, tid);
    }
    PL_FRAGS_BF(1, cur, 1);
    PL_MFMAS_BF(0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (has_next.value) PL_FRAGS_BF(0, nxt, 0);
    PL_MFMAS_BF(1);
  };
  // PL_BF16X6 variant: raw fp32 fragments are double-buffered; each is split into three bf16
  // planes right before its 24 MFMAs.  Measured at B=4096: 60 us per forward GEMM against 41 us in
  // PL_BF16 mode (whose loop is bound by staging 32 KB of fp32 operands per K tile from L2, ~10.7
  // TB/s chip-wide) and 77 us on the fp32 MFMA.  The extra 19 us is the wave's VALU ISSUE port:
  // ~450 split instructions per tile at 4 cycles each exceed the 1,536 cycles of matrix work, and a
  // lone wave per SIMD issues one instruction at a time -- weaving the split into the MFMA shadows
  // (sched_group_barrier) produced the intended ISA and changed nothing, and neither did 8-wave
  // workgroups (two waves per SIMD), two operand tiles in flight (a second staging register set; the
  // dual launch then loses its two-blocks-per-CU residency), or weaving the staging ds_writes /
  // global_loads into the MFMA shadows.  Compile-time ablations (-DPL_ABLATE, arith 3 / 4 of
  // pl_gemm_arith; forward GEMM, us): bf16 40, full split + ONE product 44, NO split + six products
  // 51-55, full bf16x6 61: the six MFMAs, not the split, are the larger share, and the two add
  // super-linearly (both want the wave's single issue port).  Untried: splitting once per element
  // at staging (bf16 planes in LDS: half the split work and fragment bytes for k-contiguous operands).
  float xa[2][2][8], xb[2][NB][8];
#define PL_FRAGS_X6(set, buf, s16)                                                       \
  do {                                                                                   \
    _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2)                                     \
      read_frag_raw<A_KS>((buf), wm * 64 + t2 * 32 + i, (s16), h, xa[set][t2]);          \
    _Pragma("unroll") for (int t2 = 0; t2 < NB; ++t2)                                    \
      read_frag_raw<B_KS>((buf) + OP_FLOATS, wn * WCOLS + t2 * 32 + i, (s16), h, xb[set][t2]); \
  } while (0)
#define PL_MF(aa, bb, ia, ib) acc[aa][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[aa][ia], b3[bb][ib], acc[aa][bb], 0, 0, 0)
#define PL_MFMAS_X6(set)                                               \
  do {                                                                 \
    bf16x8 a3[2][3], b3[NB][3];                                        \
    if (AR == 4) { /* ablation build only: no split, six MFMAs on the rounded value */ \
      _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) { _Pragma("unroll") for (int j = 0; j < 8; ++j) a3[t2][0][j] = (__bf16)xa[set][t2][j]; a3[t2][1] = a3[t2][0]; a3[t2][2] = a3[t2][0]; } \
      _Pragma("unroll") for (int t2 = 0; t2 < NB; ++t2) { _Pragma("unroll") for (int j = 0; j < 8; ++j) b3[t2][0][j] = (__bf16)xb[set][t2][j]; b3[t2][1] = b3[t2][0]; b3[t2][2] = b3[t2][0]; } \
    } else {                                                           \
    _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) split3(xa[set][t2], a3[t2]);  \
    _Pragma("unroll") for (int t2 = 0; t2 < NB; ++t2) split3(xb[set][t2], b3[t2]); \
    }                                                                  \
    _Pragma("unroll") for (int aa = 0; aa < 2; ++aa)                   \
    _Pragma("unroll") for (int bb = 0; bb < NB; ++bb) {                \
      if (AR != 3) { PL_MF(aa, bb, 2, 0); PL_MF(aa, bb, 1, 1); PL_MF(aa, bb, 0, 2);   \
      PL_MF(aa, bb, 1, 0); PL_MF(aa, bb, 0, 1); }                      \
      else { asm volatile("" :: "v"(a3[aa][1]), "v"(a3[aa][2]), "v"(b3[bb][1]), "v"(b3[bb][2])); } /* ablation: one product */ \
      PL_MF(aa, bb, 0, 0);                                             \
    }                                                                  \
  } while (0)
  auto step_x6 = [&](const int kt, auto do_store, auto do_load, auto has_next) {
    const float* cur = lds + (kt & 1) * 2 * OP_FLOATS;
    float* nxt = lds + ((kt + 1) & 1) * 2 * OP_FLOATS;
    if (do_store.value) {
      store_tile<A_KS, NT>(nxt, tid, ra);
      store_tile<B_KS, NT>(nxt + OP_FLOATS, tid, rb);
    }
    if (do_load.value) {
      const int k0 = kbeg + (kt + 2) * BK;
      ra = load_tile<A_KS, EDGE, NT>(p.A, p.lda, m0, p.M, k0, kend, a_vec, tid);
      rb = load_tile<B_KS, EDGE, NT>(p.B, p.ldb, n0, p.N, k0, kend, b_vec, tid);
    }
    PL_FRAGS_X6(1, cur, 1);
    PL_MFMAS_X6(0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (has_next.value) PL_FRAGS_X6(0, nxt, 0);
    PL_MFMAS_X6(1);
  };
  using T = std::true_type;
  using F = std::false_type;

  if (nk > 0) {
    ra = load_tile<A_KS, EDGE, NT>(p.A, p.lda, m0, p.M, kbeg, kend, a_vec, tid);
    rb = load_tile<B_KS, EDGE, NT>(p.B, p.ldb, n0, p.N, kbeg, kend, b_vec, tid);
    store_tile<A_KS, NT>(lds, tid, ra);
    store_tile<B_KS, NT>(lds + OP_FLOATS, tid, rb);
    if (nk > 1) {
      ra = load_tile<A_KS, EDGE, NT>(p.A, p.lda, m0, p.M, kbeg + BK, kend, a_vec, tid);
      rb = load_tile<B_KS, EDGE, NT>(p.B, p.ldb, n0, p.N, kbeg + BK, kend, b_vec, tid);
    }
    __syncthreads();
    int kt = 0;
    if (AR >= 2) {
      PL_FRAGS_X6(0, lds, 0);
      for (; kt + 2 < nk; ++kt) step_x6(kt, T{}, T{}, T{});
      if (kt + 1 < nk) { step_x6(kt, T{}, F{}, T{}); ++kt; }
      step_x6(kt, F{}, F{}, F{});
    } else if (AR == 1) {
      PL_FRAGS_BF(0, lds, 0);
      for (; kt + 2 < nk; ++kt) step_bf(kt, T{}, T{}, T{});
      if (kt + 1 < nk) { step_bf(kt, T{}, F{}, T{}); ++kt; }
      step_bf(kt, F{}, F{}, F{});
    } else {
      PL_FRAGS(0, lds, 0);
      for (; kt + 2 < nk; ++kt) step(kt, T{}, T{}, T{});
      if (kt + 1 < nk) { step(kt, T{}, F{}, T{}); ++kt; }
      step(kt, F{}, F{}, F{});
    }
  }
#undef PL_FRAGS_BF
#undef PL_MFMAS_BF
#undef PL_FRAGS_X6
#undef PL_MFMAS_X6
#undef PL_MF
#undef PL_SGB
#undef PL_FRAGS
#undef PL_MFMAS

  gemm_epilogue<EDGE, NB>(p, C, acc, m0, n0, wm, wn, i, h);
}

// =====================================================================================
// PL_BF16X6, "planes" pipeline: the three-way bf16 split is done ONCE per element while the operand
// tile is staged (global -> registers -> split -> LDS), and LDS holds MFMA-ready bf16 planes.
//
// Why (tools/ubench/mfma_valu*.hip, measured on gfx950): a VALU instruction issued in the shadow of
// v_mfma_f32_32x32x16_bf16 is only nearly free up to ~4 per MFMA; beyond that each costs 1.2-1.6 ns
// of matrix-pipe time.  The fragment-time split above spends ~7.3 VALU instructions per MFMA (every
// element of the A tile is split by two wavefronts, every element of B by two).  Splitting at staging
// halves the work (3.7 per MFMA) and leaves the inner loop as ds_read_b128 + MFMA; to keep the VALU
// density UNIFORM over a tile's MFMAs the split of tile kt+2 is spread over the whole of step kt, which
// needs three LDS stages (tile kt consumed, kt+1 complete and readable, kt+2 being written) and two
// staging register sets (tile kt+2 being split, kt+3 in flight).  One barrier per tile, at its end.
//
// LDS image of one operand tile (128 rows x BK k): 3 planes x BK/8 k-octets x [128 rows][8 bf16 = 16 B].
// A lane of the 32x32x16 MFMA wants row = lane&31, k = 8*(lane>>5) .. +7 of a 16-deep step: one
// ds_read_b128 from octet 2*step + (lane>>5), 32 rows contiguous -> conflict-free without padding.  The
// octet stride carries a bank rotation (32 B for four octets, 64 B for two): ds_write_b128 is banked
// modulo 32 dwords over groups of 8 consecutive lanes, and the k-contiguous staging writes of such a
// group (4 octets x 2 rows, or 2 octets x 4 rows) then cover the 32 banks exactly once (PMC with the
// rotation a multiple of 128 B: SQ_LDS_BANK_CONFLICT = a third of SQ_LDS_IDX_ACTIVE).
//   BK = 32: 3 stages x 49,920 B = 149,760 B, one workgroup per CU          (single-GEMM launches)
//   BK = 16: 3 stages x 25,344 B =  76,032 B, two workgroups per CU         (the backward dual launch)
// =====================================================================================
template <int BKX>
struct PCfg {
  static constexpr int OCT = BKX / 8;                               // k-octets per tile
  static constexpr int OCTS = 2048 + (OCT == 4 ? 32 : 64);           // octet stride, bytes (bank rotation)
  static constexpr int PLANE = OCT * OCTS;
  static constexpr int OPP = 3 * PLANE;                             // one operand tile
  static constexpr int STAGE = 2 * OPP;
  static constexpr int LDS = 3 * STAGE;
};

// Global -> registers: BKX/2 floats per thread and operand tile, in the order split8 wants them.
//   k-contiguous source (rows K-major):
//     BK 32: thread = (octet = tid&3, row = tid>>2 [+64]): v0,v1 = the 8 k of row, v2,v3 = of row + 64
//     BK 16: thread = (octet = tid&1, row = tid>>1):       v0,v1
//   k-strided source (rows M/N-major):
//     BK 32: thread = (row pair = tid&63, octet = tid>>6): eight float2; v[j/2] = (r0@k_j, r1@k_j, r0@k_j+1, r1@k_j+1)
//     BK 16: thread = (row = tid&127, octet = tid>>7):     eight floats;  v0 = k0..3, v1 = k4..7
template <bool KS, int BKX>
__device__ __forceinline__ void load_tile_p(Stage& s, const float* __restrict__ base, int ld, int r0, int k0,
                                            int tid, int rmax = 0x7fffffff) {
  if (!KS && BKX == 32) {
    // rmax: last valid row (N_EDGE launches: a ragged last column tile re-reads it; the epilogue drops those columns)
    const float* t0 = base + (size_t)min(r0 + (tid >> 2), rmax) * ld + k0 + (tid & 3) * 8;
    const float* t1 = base + (size_t)min(r0 + (tid >> 2) + 64, rmax) * ld + k0 + (tid & 3) * 8;
    s.v0 = *reinterpret_cast<const float4*>(t0);
    s.v1 = *reinterpret_cast<const float4*>(t0 + 4);
    s.v2 = *reinterpret_cast<const float4*>(t1);
    s.v3 = *reinterpret_cast<const float4*>(t1 + 4);
  } else if (!KS) {
    const float* t = base + (size_t)r0 * ld + k0 + (size_t)(tid >> 1) * ld + (tid & 1) * 8;
    s.v0 = *reinterpret_cast<const float4*>(t);
    s.v1 = *reinterpret_cast<const float4*>(t + 4);
  } else if (BKX == 32) {
    // rmax (ragged wgrad launches): the last valid column; a pair past it re-reads the last pair
    const float* t = base + (size_t)(k0 + (tid >> 6) * 8) * ld + min(r0 + (tid & 63) * 2, rmax - 1);
    float2 q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = *reinterpret_cast<const float2*>(t + (size_t)j * ld);
    s.v0 = make_float4(q[0].x, q[0].y, q[1].x, q[1].y);
    s.v1 = make_float4(q[2].x, q[2].y, q[3].x, q[3].y);
    s.v2 = make_float4(q[4].x, q[4].y, q[5].x, q[5].y);
    s.v3 = make_float4(q[6].x, q[6].y, q[7].x, q[7].y);
  } else {
    const float* t = base + (size_t)(k0 + (tid >> 7) * 8) * ld + r0 + (tid & 127);
    float q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = t[(size_t)j * ld];
    s.v0 = make_float4(q[0], q[1], q[2], q[3]);
    s.v1 = make_float4(q[4], q[5], q[6], q[7]);
  }
}

// Implicit-GEMM convolution: the A "matrix" is the NHWC input gathered on the fly (GemmArgs::conv_*).
// Same thread mapping as the k-contiguous BK 32 loader -- (octet = tid&3, row = tid>>2 [+64]) -- and because
// Cin % 32 == 0 a 32-wide K tile lies inside ONE filter tap: per tile the tap (kh, kw) and channel offset
// are wave-uniform scalars, each thread only checks its two output pixels against the image border and
// loads 2 x 32 B from the (clamped) pixel, zeroed by a select when the tap falls outside.
struct ConvRow { const float* base; int ih0, iw0; };    // pixel (b, 0, 0) and the tap-(0,0) input coordinates

__device__ __forceinline__ ConvRow conv_row(const GemmArgs& p, int m) {
  m = min(m, p.M - 1);                      // ragged last row tile: re-read the last pixel, the guarded epilogue drops it
  const int ow = m % p.conv_wo, t = m / p.conv_wo;
  const int oh = t % p.conv_ho, b = t / p.conv_ho;
  ConvRow r;
  r.base = p.A + (size_t)b * p.conv_h * p.conv_w * p.conv_cin;
  r.ih0 = oh * p.conv_stride - p.conv_pad_h;
  r.iw0 = ow * p.conv_stride - p.conv_pad_w;
  return r;
}

__device__ __forceinline__ void load_tile_conv(Stage& s, const GemmArgs& p, const ConvRow& r0, const ConvRow& r1,
                                               int k0, int tid) {
  const int tap = k0 / p.conv_cin, c0 = k0 - tap * p.conv_cin + (tid & 3) * 8;
  const int kh = tap / p.conv_kw, kw = tap - kh * p.conv_kw;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const int ih = r0.ih0 + kh, iw = r0.iw0 + kw;
    const bool ok = (unsigned)ih < (unsigned)p.conv_h && (unsigned)iw < (unsigned)p.conv_w;
    const float* t = r0.base + ((size_t)(ok ? ih : 0) * p.conv_w + (ok ? iw : 0)) * p.conv_cin + c0;
    const float4 u = *reinterpret_cast<const float4*>(t), v = *reinterpret_cast<const float4*>(t + 4);
    s.v0 = ok ? u : zero; s.v1 = ok ? v : zero;
  }
  {
    const int ih = r1.ih0 + kh, iw = r1.iw0 + kw;
    const bool ok = (unsigned)ih < (unsigned)p.conv_h && (unsigned)iw < (unsigned)p.conv_w;
    const float* t = r1.base + ((size_t)(ok ? ih : 0) * p.conv_w + (ok ? iw : 0)) * p.conv_cin + c0;
    const float4 u = *reinterpret_cast<const float4*>(t), v = *reinterpret_cast<const float4*>(t + 4);
    s.v2 = ok ? u : zero; s.v3 = ok ? v : zero;
  }
}

// Weight gradient dW[co][(kh,kw,ci)] = sum over pixels p of dy[p][co] * x_gathered[p][(kh,kw,ci)]: the TN GEMM with
// K = pixels, A = dy [pixels][Cout] as it lies in memory and B gathered from the NHWC input (GemmArgs::conv_*,
// p.B = x).  k-strided BK 32 loader mapping: thread = (column pair = tid&63, pixel octet = tid>>6).  The column
// pair fixes (tap, ci) for the whole K loop; Wo % 8 == 0 makes the octet's 8 consecutive pixels share (b, oh), so
// one div/mod pair per tile decodes them all; each pixel is one 8-byte load or a zero.
struct WgradCol { int kh, kw, ci; };

__device__ __forceinline__ WgradCol wgrad_col(const GemmArgs& p, int n) {
  WgradCol c;
  const int tap = n / p.conv_cin;
  c.ci = n - tap * p.conv_cin;
  c.kh = tap / p.conv_kw;
  c.kw = tap - c.kh * p.conv_kw;
  return c;
}

__device__ __forceinline__ void load_tile_wgrad(Stage& s, const GemmArgs& p, const WgradCol& c, int k0, int tid) {
  const int pix = k0 + (tid >> 6) * 8;
  const int ow0 = pix % p.conv_wo, q = pix / p.conv_wo;
  const int oh = q % p.conv_ho, b = q / p.conv_ho;
  const int ih = oh * p.conv_stride - p.conv_pad_h + c.kh;
  const bool rok = (unsigned)ih < (unsigned)p.conv_h;
  const float* row = p.B + ((size_t)b * p.conv_h + (rok ? ih : 0)) * p.conv_w * p.conv_cin + c.ci;
  const int iw0 = ow0 * p.conv_stride - p.conv_pad_w + c.kw, iw7 = iw0 + 7 * p.conv_stride;
  float2 v[8];
  // interior octets (the wave's eight pixels and its lanes' taps all inside the image -- most of them): no
  // per-pixel compare / select pairs in the staging path, which competes with the operand split for VALU slots
  const bool inside = rok && iw0 >= 0 && iw7 < p.conv_w;
  if (__builtin_amdgcn_ballot_w64(!inside) == 0) {
    const float* t = row + (size_t)iw0 * p.conv_cin;
    const size_t step = (size_t)p.conv_stride * p.conv_cin;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float2*>(t + j * step);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int iw = iw0 + j * p.conv_stride;
      const bool ok = rok && (unsigned)iw < (unsigned)p.conv_w;
      const float2 t = *reinterpret_cast<const float2*>(row + (size_t)(ok ? iw : 0) * p.conv_cin);
      v[j] = ok ? t : make_float2(0.f, 0.f);
    }
  }
  s.v0 = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
  s.v1 = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
  s.v2 = make_float4(v[4].x, v[4].y, v[5].x, v[5].y);
  s.v3 = make_float4(v[6].x, v[6].y, v[7].x, v[7].y);
}

// x = x0 + x1 + x2 exactly, the same RNE split as split3 (results are bit-identical to the fragment path)
template <int PLANE, int NPL = 3>
__device__ __forceinline__ void split8_store(char* __restrict__ dst, const float (&f)[8]) {
  if (NPL == 1) {                          // PL_BF16: one plane, operands rounded to bf16 (RNE)
    bf16x8 q;
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = (__bf16)f[j];
    *reinterpret_cast<bf16x8*>(dst) = q;
    return;
  }
  // stage-major over the eight values: every instruction's operands were produced a whole stage
  // (>= 4 instructions) earlier.  Written value-major, hipcc emits one serial cvt -> shift -> sub -> cvt
  // chain per value pair and every instruction waits on its predecessor (PMC: a third of the wave's
  // cycles in SQ_WAIT_INST_ANY).
  bf16x8 p0, p1, p2;
  float r1[8], r2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) p0[j] = (__bf16)f[j];
#pragma unroll
  for (int j = 0; j < 8; ++j) r1[j] = f[j] - (float)p0[j];
#pragma unroll
  for (int j = 0; j < 8; ++j) p1[j] = (__bf16)r1[j];
#pragma unroll
  for (int j = 0; j < 8; ++j) r2[j] = r1[j] - (float)p1[j];
#pragma unroll
  for (int j = 0; j < 8; ++j) p2[j] = (__bf16)r2[j];
  *reinterpret_cast<bf16x8*>(dst) = p0;
  *reinterpret_cast<bf16x8*>(dst + PLANE) = p1;
  *reinterpret_cast<bf16x8*>(dst + 2 * PLANE) = p2;
}

// Split + store one half (HALF = 0, 1) of a staged operand tile; BK 16 has a single half (0).
template <bool KS, int BKX, int HALF, int NPL = 3>
__device__ __forceinline__ void store_half_p(char* __restrict__ op, int tid, const Stage& g) {
  using Cf = PCfg<BKX>;
  if (!KS && BKX == 32) {
    char* d = op + (tid & 3) * Cf::OCTS + (tid >> 2) * 16 + HALF * 64 * 16;
    const float4 u = HALF ? g.v2 : g.v0, v = HALF ? g.v3 : g.v1;
    const float a[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    split8_store<Cf::PLANE, NPL>(d, a);
  } else if (!KS) {
    char* d = op + (tid & 1) * Cf::OCTS + (tid >> 1) * 16;
    const float a[8] = {g.v0.x, g.v0.y, g.v0.z, g.v0.w, g.v1.x, g.v1.y, g.v1.z, g.v1.w};
    split8_store<Cf::PLANE, NPL>(d, a);
  } else if (BKX == 32) {
    char* d = op + (tid >> 6) * Cf::OCTS + (tid & 63) * 32 + HALF * 16;
    float a[8];
    if (HALF == 0) { a[0] = g.v0.x; a[1] = g.v0.z; a[2] = g.v1.x; a[3] = g.v1.z; a[4] = g.v2.x; a[5] = g.v2.z; a[6] = g.v3.x; a[7] = g.v3.z; }
    else           { a[0] = g.v0.y; a[1] = g.v0.w; a[2] = g.v1.y; a[3] = g.v1.w; a[4] = g.v2.y; a[5] = g.v2.w; a[6] = g.v3.y; a[7] = g.v3.w; }
    split8_store<Cf::PLANE, NPL>(d, a);
  } else {
    char* d = op + (tid >> 7) * Cf::OCTS + (tid & 127) * 16;
    const float a[8] = {g.v0.x, g.v0.y, g.v0.z, g.v0.w, g.v1.x, g.v1.y, g.v1.z, g.v1.w};
    split8_store<Cf::PLANE, NPL>(d, a);
  }
}

// whole tiles only (M, N % 128 == 0, every K slice % BKX == 0, 16-byte aligned rows); 4 wavefronts 2x2
// Two staging register sets: tile kt+2 being split, tile kt+3 in flight from L2/HBM.
// What bounds this loop now (same-box A/B, tools/ab_env.py, tools/gemm_scan.py; B = 4096 shapes):
//   * steady state 1.33-1.39 us per 32-k tile against 0.79 us of matrix-pipe time; PL_BF16 mode (EIGHT
//     MFMAs per tile) sits at 0.76-0.97 us per tile: moving 32 KB of fp32 operands per tile and CU from L2
//     (8.4 MB per step chip-wide, ~9-10 TB/s) is a floor of its own, and it only partly overlaps the MFMAs;
//   * a third staging register set (two tiles in flight) changed nothing: not latency, bandwidth;
//   * operands delivered PRE-split by the producing kernel (bn_apply writing three bf16 planes, no split
//     VALU left in the GEMM at all) ran SLOWER (57.8 vs 55.9 us per forward GEMM): 6 B instead of 4 B per
//     element through the same L2 path.  Removed again.
//   PMC (SQ_*): MFMA busy 46 % of wave cycles, WAIT_ANY 22 %, issue time of the ~5.6 non-MFMA instructions
//   per MFMA gap not hidden (the guide's limit is <= 5 per 32x32x16 gap, hand-placed).
// N_EDGE (k-contiguous B, BK 32 only): N need not be a multiple of 128 -- the last column tile clamps its B
// rows and the guarded epilogue drops the columns >= N (the 64-wide layer1 convolutions, the 1088-wide head).
// With the convolution gather (A_CONV) M may be ragged too: conv_row clamps the pixel index.
// B_WGRAD + N_EDGE: ragged Cout (M) and KH*KW*Cin (N): both loaders clamp, the guarded epilogue drops the rest.
// NPL = 3: PL_BF16X6 (three planes, six products).  NPL = 1: PL_BF16 on the same pipeline (plane 0 only, one product
// per tile pair, operands rounded to bf16 while staged): the conv path's throughput mode.
template <bool A_KS, bool B_KS, int BKX, bool A_CONV = false, bool N_EDGE = false, bool B_WGRAD = false, int NPL = 3>
__device__ __forceinline__ void gemm_body_planes(const GemmArgs& p, const int block_id, const int nwork,
                                                 char* __restrict__ lds) {
  using Cf = PCfg<BKX>;
  constexpr int KSTEPS = BKX / 16;          // 16-deep MFMA steps per tile
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  static_assert(!N_EDGE || ((!B_KS || B_WGRAD) && BKX == 32), "ragged N only with the k-contiguous or gathered BK 32 B loader");
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nlast = (N_EDGE && !B_WGRAD) ? p.N - 1 : 0x7fffffff;
  const int splits = p.split_k > 1 ? p.split_k : 1;
  const int ntiles = nwork / splits;
  int w = block_id;
  if ((nwork & 7) == 0) w = (w & 7) * (nwork >> 3) + (w >> 3);
  const int slice = w / ntiles;
  const int t = w - slice * ntiles;
  const int m0 = (t / tiles_n) * BM;
  const int n0 = (t % tiles_n) * BN;
  int kbeg = 0, kend = p.K;
  float* C = p.C;
  if (p.split_k > 1) {
    const int per = ((p.K / BKX + p.split_k - 1) / p.split_k) * BKX;     // whole tiles; the last slice may be shorter
    kbeg = min(slice * per, p.K);
    kend = min(kbeg + per, p.K);
    C += (size_t)slice * p.M * p.ldc;
  }
  const int nk = (kend - kbeg) / BKX;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment sets: BK 32: set s = 16-deep step s of the tile in flight; BK 16: set = tile parity
  bf16x8 fa[2][2][NPL], fb[2][2][NPL];
  const int arow = (wm * 64 + i) * 16, brow = (wn * 64 + i) * 16;
#define PL_FRAGS_P(set, buf, s16)                                                                        \
  do {                                                                                                   \
    const char* qa = (buf) + (2 * (s16) + h) * Cf::OCTS + arow;                                          \
    const char* qb = (buf) + Cf::OPP + (2 * (s16) + h) * Cf::OCTS + brow;                                \
    _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2)                                                     \
    _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                                                 \
      fa[set][t2][pl] = *reinterpret_cast<const bf16x8*>(qa + t2 * 32 * 16 + pl * Cf::PLANE);            \
      fb[set][t2][pl] = *reinterpret_cast<const bf16x8*>(qb + t2 * 32 * 16 + pl * Cf::PLANE);            \
    }                                                                                                    \
  } while (0)
#define PL_MFP(set, aa, bb, ia, ib) \
  acc[aa][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][aa][ia], fb[set][bb][ib], acc[aa][bb], 0, 0, 0)
#define PL_MF6(set, aa, bb)                                                                      \
  do {                                                                                           \
    if constexpr (NPL == 3) {                                                                    \
      PL_MFP(set, aa, bb, 2, 0); PL_MFP(set, aa, bb, 1, 1); PL_MFP(set, aa, bb, 0, 2);           \
      PL_MFP(set, aa, bb, 1, 0); PL_MFP(set, aa, bb, 0, 1);                                      \
    }                                                                                            \
    PL_MFP(set, aa, bb, 0, 0);                                                                   \
  } while (0)
#define PL_MFMAS_ROW(set, aa) do { PL_MF6(set, aa, 0); PL_MF6(set, aa, 1); } while (0)
#define PL_MFMAS_P(set) do { PL_MFMAS_ROW(set, 0); PL_MFMAS_ROW(set, 1); } while (0)

  char* cur = lds;                    // tile kt: being consumed
  char* nx1 = lds + Cf::STAGE;        // tile kt+1: complete, readable
  char* nx2 = lds + 2 * Cf::STAGE;    // tile kt+2: being written during step kt
  static_assert(!A_CONV || (!A_KS && BKX == 32), "the convolution gather is a k-contiguous BK 32 loader");
  static_assert(!B_WGRAD || (A_KS && B_KS && BKX == 32 && !A_CONV), "wgrad is the TN BK 32 loop");
  using SA = Stage;
  SA ra0, ra1;                        // staging register sets: tile t lives in set t % 2
  Stage rb0, rb1;
  ConvRow cr0 = {}, cr1 = {};
  if (A_CONV) { cr0 = conv_row(p, m0 + (tid >> 2)); cr1 = conv_row(p, m0 + (tid >> 2) + 64); }
  auto load_a = [&](SA& d, const int k0) {
    if constexpr (A_CONV) load_tile_conv(d, p, cr0, cr1, k0, tid);
    else load_tile_p<A_KS, BKX>(d, p.A, p.lda, m0, k0, tid, (B_WGRAD && N_EDGE) ? p.M - 1 : 0x7fffffff);
  };
  WgradCol wcol = {};
  if (B_WGRAD) wcol = wgrad_col(p, N_EDGE ? min(n0 + (tid & 63) * 2, p.N - 2) : n0 + (tid & 63) * 2);
  auto load_b = [&](Stage& d, const int k0) {
    if constexpr (B_WGRAD) load_tile_wgrad(d, p, wcol, k0, tid);
    else load_tile_p<B_KS, BKX>(d, p.B, p.ldb, n0, k0, tid, nlast);
  };
  auto store_a = [&](char* op, auto half, const SA& g) {
    store_half_p<A_KS, BKX, decltype(half)::value, NPL>(op, tid, g);
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;

  // One pipeline step.  P = parity of kt (selects the fragment set for BK 16); sa/sb: raw tile kt+2 to
  // split now; la/lb: receive tile kt+3.  STEADY: every flag known true at compile time.
  auto step = [&](const int kt, auto par, auto steady, const SA& sa, const Stage& sb, SA& la, Stage& lb) {
    constexpr int P = decltype(par)::value;
    constexpr bool STEADY = decltype(steady)::value;
    const bool do_load = STEADY || kt + 3 < nk;
    const bool do_split = STEADY || kt + 2 < nk;
    const bool has_next = STEADY || kt + 1 < nk;
    if (do_load) {
      const int k0 = kbeg + (kt + 3) * BKX;
      load_a(la, k0);
      load_b(lb, k0);
    }
    if (KSTEPS == 2) {
      PL_FRAGS_P(1, cur, 1);
      if (do_split) { store_a(nx2, H0{}, sa); store_half_p<B_KS, BKX, 0, NPL>(nx2 + Cf::OPP, tid, sb); }
      PL_MFMAS_P(0);
      if (has_next) PL_FRAGS_P(0, nx1, 0);
      if (do_split) { store_a(nx2, H1{}, sa); store_half_p<B_KS, BKX, 1, NPL>(nx2 + Cf::OPP, tid, sb); }
      PL_MFMAS_P(1);
    } else {
      if (has_next) PL_FRAGS_P(1 - P, nx1, 0);
      if (do_split) store_a(nx2, H0{}, sa);
      PL_MFMAS_ROW(P, 0);
      if (do_split) store_half_p<B_KS, BKX, 0, NPL>(nx2 + Cf::OPP, tid, sb);
      PL_MFMAS_ROW(P, 1);
    }
    if (STEADY) {
      // Schedule of the step (one basic block): ONE MFMA, then at most four of the split's VALU
      // instructions in its shadow (more than ~4 per MFMA stop being free, tools/ubench), the global
      // loads up front, the fragment reads and the plane writes threaded between.
      constexpr int NG = (A_KS ? 8 : BKX / 8) + (B_KS ? 8 : BKX / 8);      // global loads per step
#pragma unroll
      for (int q = 0; q < (NPL == 3 ? 24 : 4) * KSTEPS; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        if (q < NG) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    char* o = cur; cur = nx1; nx1 = nx2; nx2 = o;
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using T = std::true_type;
  using F = std::false_type;
  if (nk > 0) {
    // prologue: tiles 0 and 1 into stages 0 and 1, tile 2 into register set 0
    load_a(ra0, kbeg);
    load_b(rb0, kbeg);
    if (nk > 1) {
      load_a(ra1, kbeg + BKX);
      load_b(rb1, kbeg + BKX);
    }
    store_a(cur, H0{}, ra0); store_half_p<B_KS, BKX, 0, NPL>(cur + Cf::OPP, tid, rb0);
    if (KSTEPS == 2) { store_a(cur, H1{}, ra0); store_half_p<B_KS, BKX, 1, NPL>(cur + Cf::OPP, tid, rb0); }
    if (nk > 2) {
      load_a(ra0, kbeg + 2 * BKX);
      load_b(rb0, kbeg + 2 * BKX);
    }
    if (nk > 1) {
      store_a(nx1, H0{}, ra1); store_half_p<B_KS, BKX, 0, NPL>(nx1 + Cf::OPP, tid, rb1);
      if (KSTEPS == 2) { store_a(nx1, H1{}, ra1); store_half_p<B_KS, BKX, 1, NPL>(nx1 + Cf::OPP, tid, rb1); }
    }
    __syncthreads();
    PL_FRAGS_P(0, cur, 0);
    int kt = 0;
    // step kt splits tile kt+2 (set kt % 2) and loads tile kt+3 into the other set
    for (; kt + 4 < nk; kt += 2) {
      step(kt, P0{}, T{}, ra0, rb0, ra1, rb1);
      step(kt + 1, P1{}, T{}, ra1, rb1, ra0, rb0);
    }
    for (; kt < nk; kt += 2) {
      step(kt, P0{}, F{}, ra0, rb0, ra1, rb1);
      if (kt + 1 < nk) step(kt + 1, P1{}, F{}, ra1, rb1, ra0, rb0);
    }
  }
#undef PL_FRAGS_P
#undef PL_MFP
#undef PL_MF6
#undef PL_MFMAS_ROW
#undef PL_MFMAS_P
  // wave-uniform: whole output tiles only (a ragged problem's interior tiles qualify, its last row / column tile not)
  if ((!N_EDGE || (n0 + BN <= p.N && m0 + BM <= p.M)) && epilogue_vec_ok(p, C)) {
    __syncthreads();                            // every wave is done reading operand tiles from LDS
    gemm_epilogue_vec(p, C, acc, m0, n0, wm, wn, i, h, reinterpret_cast<float*>(lds) + wave * 64 * 64);
    return;
  }
  gemm_epilogue<N_EDGE, 2>(p, C, acc, m0, n0, wm, wn, i, h);
}

template <bool A_KS, bool B_KS>
__global__ __launch_bounds__(256) void gemm_x6_planes_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[PCfg<32>::LDS];
  gemm_body_planes<A_KS, B_KS, 32>(p, blockIdx.x, gridDim.x, lds);
}

template <bool N_EDGE>
__global__ __launch_bounds__(256) void conv_x6_planes_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[PCfg<32>::LDS];
  gemm_body_planes<false, false, 32, true, N_EDGE>(p, blockIdx.x, gridDim.x, lds);
}

template <bool MN_EDGE, int NPL = 3>      // NPL = 1: PL_BF16 (operands rounded to bf16 while staged, one product)
__global__ __launch_bounds__(256) void conv_wgrad_x6_planes_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[PCfg<32>::LDS];
  gemm_body_planes<true, true, 32, false, MN_EDGE, true, NPL>(p, blockIdx.x, gridDim.x, lds);
}

// PL_BF16 on the planes pipeline (conv path throughput mode): the same bodies with NPL = 1
template <bool N_EDGE>
__global__ __launch_bounds__(256) void conv_bf16_planes_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[PCfg<32>::LDS];
  gemm_body_planes<false, false, 32, true, N_EDGE, false, 1>(p, blockIdx.x, gridDim.x, lds);
}

template <bool N_EDGE>
__global__ __launch_bounds__(256) void gemm_bf16_planes_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[PCfg<32>::LDS];
  gemm_body_planes<false, false, 32, false, N_EDGE, false, 1>(p, blockIdx.x, gridDim.x, lds);
}

struct GemmArgs4 { GemmArgs g[4]; };

template <bool N_EDGE, int NPL>
__global__ __launch_bounds__(256) void conv_x6_planes_group4_kernel(GemmArgs4 P, int per) {
  __shared__ __attribute__((aligned(16))) char lds[PCfg<32>::LDS];
  const int g = blockIdx.x / per;
  gemm_body_planes<false, false, 32, true, N_EDGE, false, NPL>(P.g[g], blockIdx.x - g * per, per, lds);
}

__global__ __launch_bounds__(256) void gemm_x6_planes_nedge_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[PCfg<32>::LDS];
  gemm_body_planes<false, false, 32, false, true>(p, blockIdx.x, gridDim.x, lds);
}

// backward pair in one launch (see gemm_f32_dual_kernel): BK 16, two workgroups per CU
__global__ __launch_bounds__(256, 2) void gemm_x6_planes_dual_kernel(GemmArgs p0, GemmArgs p1, int n0) {
  __shared__ __attribute__((aligned(16))) char lds[PCfg<16>::LDS];
  if ((int)blockIdx.x < n0)
    gemm_body_planes<false, true, 16>(p0, blockIdx.x, n0, lds);
  else
    gemm_body_planes<true, true, 16>(p1, blockIdx.x - n0, gridDim.x - n0, lds);
}

template <bool A_KS, bool B_KS, bool EDGE, int AR = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gemm_f32_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];   // one static array (73,728 B)
  gemm_body<A_KS, B_KS, EDGE, AR, NW>(p, blockIdx.x, gridDim.x, lds);
}

// Two independent whole-tile GEMMs in ONE launch: workgroups [0, n0) run the NN problem
// (da = dz W), the rest the TN problem (dW = dz^T a, split-K slabs).  Both read the same dz;
// with 256 + 256 workgroups every CU hosts one of each (2 x 73.7 KB LDS, 2 waves per SIMD), so
// one GEMM's prologue / epilogue-store / barrier bubbles are filled by the other's MFMAs, and
// a launch boundary plus its dirty-L2 write-back disappears.
template <int AR, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_f32_dual_kernel(GemmArgs p0, GemmArgs p1, int n0) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  if ((int)blockIdx.x < n0)
    gemm_body<false, true, false, AR, NW>(p0, blockIdx.x, n0, lds);
  else
    gemm_body<true, true, false, AR, NW>(p1, blockIdx.x - n0, gridDim.x - n0, lds);
}

}  // namespace

int gemm_stat_groups(int M) { return 2 * ((M + BM - 1) / BM); }

// ---- measurement hook: HIP events around every GEMM launch, on the launch stream ------------
namespace {
struct ProfRec { hipEvent_t e0, e1; double flops; };
int g_prof_on = 0;                    // 0 = off, n = every n-th GEMM launch gets a pair of events
std::vector<ProfRec> g_prof_pool;     // events are created once and reused
size_t g_prof_used = 0, g_prof_seen = 0;
}  // namespace

// on = sampling period: 1 times every GEMM launch; n > 1 every n-th one (an event pair costs ~2.5 us of stream
// time per launch -- 5 % of the lifter step when every launch carries one)
int prof_enable(int on) {
  g_prof_on = on > 0 ? on : 0;
  g_prof_used = 0;
  g_prof_seen = 0;
  return PL_OK;
}

int prof_read(double min_flops, double max_flops, double* ms_total, int64_t* launches, double* flops_total) {
  double ms = 0, fl = 0;
  int64_t n = 0;
  for (size_t i = 0; i < g_prof_used; ++i) {
    const ProfRec& r = g_prof_pool[i];
    if (r.flops < min_flops || r.flops > max_flops) continue;
    if (hipEventSynchronize(r.e1) != hipSuccess) PL_FAIL(PL_EHIP, "prof_read: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) PL_FAIL(PL_EHIP, "prof_read: elapsed time failed");
    ms += t; fl += r.flops; ++n;
  }
  *ms_total = ms; *launches = n; *flops_total = fl;
  return PL_OK;
}

static ProfRec* prof_begin(const GemmArgs& a, hipStream_t s) {
  if (!g_prof_on) return nullptr;
  if (g_prof_seen++ % (size_t)g_prof_on) return nullptr;
  if (g_prof_used == g_prof_pool.size()) {
    ProfRec r;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return nullptr;
    g_prof_pool.push_back(r);
  }
  ProfRec* r = &g_prof_pool[g_prof_used++];
  r->flops = 2.0 * a.M * a.N * a.K;
  (void)hipEventRecord(r->e0, s);
  return r;
}

// the same hook for launches made elsewhere (gemm_planes.hip)
void* prof_begin_flops(double flops, hipStream_t s) {
  GemmArgs a = {};
  a.M = 1; a.N = 1; a.K = 1;
  ProfRec* r = prof_begin(a, s);
  if (r) r->flops = flops;
  return r;
}
void prof_end(void* rec, hipStream_t s) {
  if (rec) (void)hipEventRecord(static_cast<ProfRec*>(rec)->e1, s);
}

// hot path: whole 128x128x32 tiles in every split, 16-byte aligned rows
static bool whole_tiles(const GemmArgs& a) {
  const int splits = a.split_k > 1 ? a.split_k : 1;
  const int kper = splits > 1 ? (((a.K + splits - 1) / splits) + BK - 1) / BK * BK : a.K;
  return (a.M % BM == 0) && (a.N % BN == 0) && (a.K % BK == 0) && (kper * splits == a.K || splits == 1) &&
         (a.lda % 4 == 0) && (a.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0) &&
         ((reinterpret_cast<uintptr_t>(a.B) & 15) == 0);
}

// the planes pipeline additionally wants even row strides for its float2 loads and whole K slices
static bool planes_ok(const GemmArgs& a) {
  const int splits = a.split_k > 1 ? a.split_k : 1;
  return a.K % (BK * splits) == 0 && (a.lda % 4 == 0) && (a.ldb % 4 == 0);
}

// PL_BF16X6 main loop of the single-GEMM launches: planes pipeline unless POSELIFT_X6_FRAG=1
static bool x6_planes_default() {
  static const bool on = [] {
    const char* e = getenv("POSELIFT_X6_FRAG");
    return !(e && e[0] == '1');
  }();
  return on;
}

static int grid_of(const GemmArgs& a) {
  return ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * (a.split_k > 1 ? a.split_k : 1);
}

// da = dz W (NN) and dW = dz^T a (TN) in one launch; falls back to two launches off the hot path
int launch_gemm_f32_pair(const GemmArgs& nn, const GemmArgs& tn, hipStream_t s) {
  const int g0 = grid_of(nn), g1 = grid_of(tn);
  if (!whole_tiles(nn) || !whole_tiles(tn) || (g0 & 7) || (g1 & 7)) {
    PL_TRY(launch_gemm_f32(kNN, nn, s));
    return launch_gemm_f32(kTN, tn, s);
  }
  GemmArgs both = nn;                 // profiling record: one launch, the work of two
  ProfRec* prof = prof_begin(both, s);
  if (prof) prof->flops += 2.0 * tn.M * tn.N * tn.K;
  if (nn.arith != tn.arith) PL_FAIL(PL_EINVAL, "gemm pair: mixed arithmetic");
  if (nn.arith == 2 && x6_planes_default() && planes_ok(nn) && planes_ok(tn))
    hipLaunchKernelGGL(gemm_x6_planes_dual_kernel, dim3(g0 + g1), dim3(NTHR), 0, s, nn, tn, g0);
  else if (nn.arith == 2) hipLaunchKernelGGL((gemm_f32_dual_kernel<2, 4>), dim3(g0 + g1), dim3(NTHR), 0, s, nn, tn, g0);
  else if (nn.arith == 1) hipLaunchKernelGGL((gemm_f32_dual_kernel<1, 4>), dim3(g0 + g1), dim3(NTHR), 0, s, nn, tn, g0);
  else hipLaunchKernelGGL((gemm_f32_dual_kernel<0, 4>), dim3(g0 + g1), dim3(NTHR), 0, s, nn, tn, g0);
  if (prof) (void)hipEventRecord(prof->e1, s);
  PL_CHECK_LAUNCH("gemm_f32_dual");
  return PL_OK;
}

int launch_conv_nhwc(const GemmArgs& a, hipStream_t s) {
  if (!a.A || !a.B || !a.C) PL_FAIL(PL_EINVAL, "conv: null operand");
  const int splits = a.split_k > 1 ? a.split_k : 1;         // > 1: a.C = slabs [splits][M][ldc], plain sums
  if (a.conv_cin <= 0 || a.conv_cin % 32 || a.M < 1 || a.N < 1 || a.K % BK || a.K % a.conv_cin ||
      (splits > 1 && (a.arith == 1 || (a.K / BK) % splits)))
    PL_FAIL(PL_ESHAPE, "conv: needs Cin %% 32 == 0 and whole K slices (M=%d N=%d K=%d Cin=%d splits=%d)", a.M, a.N,
            a.K, a.conv_cin, splits);
  if ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B)) & 15)
    PL_FAIL(PL_EINVAL, "conv: operands not 16-byte aligned");
  ProfRec* prof = prof_begin(a, s);
  const dim3 grid(((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * splits);
  const bool edge = a.N % BN || a.M % BM;
  if (a.arith == 1) {
    if (edge) hipLaunchKernelGGL(conv_bf16_planes_kernel<true>, grid, dim3(NTHR), 0, s, a);
    else hipLaunchKernelGGL(conv_bf16_planes_kernel<false>, grid, dim3(NTHR), 0, s, a);
  } else if (edge) hipLaunchKernelGGL(conv_x6_planes_kernel<true>, grid, dim3(NTHR), 0, s, a);
  else hipLaunchKernelGGL(conv_x6_planes_kernel<false>, grid, dim3(NTHR), 0, s, a);
  if (prof) (void)hipEventRecord(prof->e1, s);
  PL_CHECK_LAUNCH("conv_x6_planes");
  return PL_OK;
}

// dW = dy^T x_gathered: a.A = dy [pixels][Cout] (lda = Cout), a.B = x (NHWC), a.C = slabs when split_k > 1,
// a.M = Cout, a.N = KH*KW*Cin, a.K = B*Ho*Wo; needs Cout and Cin even, Wo % 8 == 0 and pixels % 32 == 0; the K slices
// are whole 32-pixel tiles, the last one possibly shorter.
int launch_conv_wgrad(const GemmArgs& a, hipStream_t s) {
  if (!a.A || !a.B || !a.C) PL_FAIL(PL_EINVAL, "conv wgrad: null operand");
  const int splits = a.split_k > 1 ? a.split_k : 1;
  if (a.conv_cin <= 0 || (a.conv_cin & 1) || a.M < 2 || (a.M & 1) || a.N < 2 || a.N % a.conv_cin || a.conv_wo % 8 ||
      a.K % BK || a.K != (a.K / (a.conv_ho * a.conv_wo)) * a.conv_ho * a.conv_wo || (a.lda & 1))
    PL_FAIL(PL_ESHAPE, "conv wgrad: needs even Cout and Cin, Wo %% 8 == 0, pixels %% 32 == 0 "
                       "(M=%d N=%d K=%d Cin=%d Wo=%d splits=%d)", a.M, a.N, a.K, a.conv_cin, a.conv_wo, splits);
  if ((reinterpret_cast<uintptr_t>(a.A) & 7) || (reinterpret_cast<uintptr_t>(a.B) & 7))
    PL_FAIL(PL_EINVAL, "conv wgrad: operands misaligned");
  ProfRec* prof = prof_begin(a, s);
  const dim3 grid(((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * splits);
  const bool edge = a.M % BM || a.N % BN, bf = a.arith == 1;
  if (edge && bf) hipLaunchKernelGGL((conv_wgrad_x6_planes_kernel<true, 1>), grid, dim3(NTHR), 0, s, a);
  else if (edge) hipLaunchKernelGGL((conv_wgrad_x6_planes_kernel<true, 3>), grid, dim3(NTHR), 0, s, a);
  else if (bf) hipLaunchKernelGGL((conv_wgrad_x6_planes_kernel<false, 1>), grid, dim3(NTHR), 0, s, a);
  else hipLaunchKernelGGL((conv_wgrad_x6_planes_kernel<false, 3>), grid, dim3(NTHR), 0, s, a);
  if (prof) (void)hipEventRecord(prof->e1, s);
  PL_CHECK_LAUNCH("conv_wgrad_x6_planes");
  return PL_OK;
}

int launch_conv_nhwc_group4(const GemmArgs* a, hipStream_t s) {
  GemmArgs4 P;
  for (int g = 0; g < 4; ++g) {
    const GemmArgs& q = a[g];
    if (!q.A || !q.B || !q.C) PL_FAIL(PL_EINVAL, "conv group: null operand");
    if (q.M != a[0].M || q.N != a[0].N || q.K != a[0].K || q.conv_cin != a[0].conv_cin)
      PL_FAIL(PL_ESHAPE, "conv group: the four problems must have one shape");
    if (q.conv_cin <= 0 || q.conv_cin % 32 || q.M < 1 || q.N < 1 || q.K % BK || q.K % q.conv_cin || q.split_k > 1)
      PL_FAIL(PL_ESHAPE, "conv group: needs Cin %% 32 == 0");
    if ((reinterpret_cast<uintptr_t>(q.A) | reinterpret_cast<uintptr_t>(q.B)) & 15)
      PL_FAIL(PL_EINVAL, "conv group: operands not 16-byte aligned");
    P.g[g] = q;
  }
  const int per = ((a[0].M + BM - 1) / BM) * ((a[0].N + BN - 1) / BN);
  GemmArgs all = a[0];
  ProfRec* prof = prof_begin(all, s);
  if (prof) prof->flops *= 4.0;
  const bool edge = a[0].N % BN || a[0].M % BM, bf = a[0].arith == 1;
  if (edge && bf) hipLaunchKernelGGL((conv_x6_planes_group4_kernel<true, 1>), dim3(4 * per), dim3(NTHR), 0, s, P, per);
  else if (edge) hipLaunchKernelGGL((conv_x6_planes_group4_kernel<true, 3>), dim3(4 * per), dim3(NTHR), 0, s, P, per);
  else if (bf) hipLaunchKernelGGL((conv_x6_planes_group4_kernel<false, 1>), dim3(4 * per), dim3(NTHR), 0, s, P, per);
  else hipLaunchKernelGGL((conv_x6_planes_group4_kernel<false, 3>), dim3(4 * per), dim3(NTHR), 0, s, P, per);
  if (prof) (void)hipEventRecord(prof->e1, s);
  PL_CHECK_LAUNCH("conv_x6_planes_group4");
  return PL_OK;
}

int launch_gemm_f32(GemmLayout layout, const GemmArgs& a, hipStream_t s) {
  if (!a.A || !a.B || !a.C) PL_FAIL(PL_EINVAL, "gemm_f32: null operand");
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) PL_FAIL(PL_ESHAPE, "gemm_f32: bad shape %dx%dx%d", a.M, a.N, a.K);
  if ((a.stat_sum != nullptr) != (a.stat_m2 != nullptr)) PL_FAIL(PL_EINVAL, "gemm_f32: stats need both buffers");
  if (a.col_scale && !a.col_shift) PL_FAIL(PL_EINVAL, "gemm_f32: scale without shift");
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  const int splits = a.split_k > 1 ? a.split_k : 1;
  dim3 grid(tiles * splits), block(NTHR);
  const size_t lds_bytes = 0;   // LDS is static
  // small batch: a handful of tiles for 256 CUs -- the contraction split over the chip instead (gemm_thin.hip)
  // (only off the tile grid: whole-tile batches keep one arithmetic whatever their size -- SyncBN shards of 128 rows are
  //  bitwise the single-process batch, tests/test_gpu_dp.py)
  if (a.thin_scratch && !whole_tiles(a) && a.M <= thin_gemm_max_m() && thin_gemm_ok(layout, a))
    return launch_gemm_thin(layout, a, a.thin_scratch, a.thin_scratch_floats, s);
  ProfRec* prof = prof_begin(a, s);
  const bool whole = whole_tiles(a);
#ifdef PL_ABLATE   /* timing-only variants of the bf16x6 loop (wrong results by construction) */
#define PL_ABLATE_LAUNCH(AKS, BKS)                                                                        \
    else if (whole && a.arith == 3) hipLaunchKernelGGL((gemm_f32_kernel<AKS, BKS, false, 3>), grid, block, lds_bytes, s, a); \
    else if (whole && a.arith == 4) hipLaunchKernelGGL((gemm_f32_kernel<AKS, BKS, false, 4>), grid, block, lds_bytes, s, a);
#else
#define PL_ABLATE_LAUNCH(AKS, BKS)
#endif
#define PL_GEMM_LAUNCH(AKS, BKS)                                                                          \
  do {                                                                                                    \
    if (whole && planes_ok(a) && (a.arith == 5 || (a.arith == 2 && x6_planes_default())))                 \
      hipLaunchKernelGGL((gemm_x6_planes_kernel<AKS, BKS>), grid, block, lds_bytes, s, a);                 \
    else if (whole && (a.arith == 2 || a.arith == 5 || a.arith == 6)) hipLaunchKernelGGL((gemm_f32_kernel<AKS, BKS, false, 2>), grid, block, lds_bytes, s, a); \
    PL_ABLATE_LAUNCH(AKS, BKS)                                                                            \
    else if (whole && a.arith == 1) hipLaunchKernelGGL((gemm_f32_kernel<AKS, BKS, false, 1>), grid, block, lds_bytes, s, a); \
    else if (whole) hipLaunchKernelGGL((gemm_f32_kernel<AKS, BKS, false>), grid, block, lds_bytes, s, a);      \
    else hipLaunchKernelGGL((gemm_f32_kernel<AKS, BKS, true>), grid, block, lds_bytes, s, a);             \
  } while (0)
  // conv path, PL_BF16 on the planes pipeline (GemmArgs::arith 7, NT only)
  if (layout == kNT && a.arith == 7 && a.split_k <= 1 && a.M % BM == 0 && a.K % BK == 0 && a.lda % 4 == 0 &&
      a.ldb % 4 == 0 && ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B)) & 15) == 0) {
    if (a.N % BN) hipLaunchKernelGGL(gemm_bf16_planes_kernel<true>, grid, block, lds_bytes, s, a);
    else hipLaunchKernelGGL(gemm_bf16_planes_kernel<false>, grid, block, lds_bytes, s, a);
    if (prof) (void)hipEventRecord(prof->e1, s);
    PL_CHECK_LAUNCH("gemm_bf16_planes");
    return PL_OK;
  }
  // NT, PL_BF16X6, whole M and K tiles but a ragged N: the planes kernel with clamped B rows + guarded stores
  const bool nt_ragged_n = layout == kNT && a.arith == 2 && x6_planes_default() && !whole && a.split_k <= 1 &&
                           a.M % BM == 0 && a.K % BK == 0 && a.lda % 4 == 0 && a.ldb % 4 == 0 &&
                           ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B)) & 15) == 0;
  if (nt_ragged_n) {
    hipLaunchKernelGGL(gemm_x6_planes_nedge_kernel, grid, block, lds_bytes, s, a);
    if (prof) (void)hipEventRecord(prof->e1, s);
    PL_CHECK_LAUNCH("gemm_x6_planes_nedge");
    return PL_OK;
  }
  switch (layout) {
    case kNT: PL_GEMM_LAUNCH(false, false); break;
    case kNN: PL_GEMM_LAUNCH(false, true); break;
    case kTN: PL_GEMM_LAUNCH(true, true); break;
    default: PL_FAIL(PL_EINVAL, "gemm_f32: bad layout %d", (int)layout);
  }
#undef PL_GEMM_LAUNCH
  if (prof) (void)hipEventRecord(prof->e1, s);
  PL_CHECK_LAUNCH("gemm_f32");
  return PL_OK;
}

}  // namespace pl
