// "Planes in HBM" GEMM main loop (gfx950): operands arrive PRE-SPLIT as bf16 planes, written once by the kernel
// that produces the tensor (bn_apply, bn_bwd_dz, adamw, ...), never split inside the GEMM.
//
//   x (fp32) = x0 + x1 + x2 exactly (three RNE bf16 pieces, 8+8+8 mantissa bits); a*b is accumulated as the six
//   products a0b0, a0b1, a1b0, a0b2, a1b1, a2b0 on v_mfma_f32_32x32x16_bf16 (PL_BF16X6, NPL = 3), or as the
//   single product of the rounded operands (PL_BF16, NPL = 1: a plane IS the bf16 tensor).
//
// A plane tensor is [NPL][rows][cols] bf16, row-major, plane stride given in elements.  What this buys over
// the round-1 loop (fp32 operands through registers, split on the vector ALU, ds_write):
//   * global -> LDS by global_load_lds_dwordx4 (LDS-DMA): no VGPR round trip, no ds_write, no split VALU --
//     the inner loop is ds_read + MFMA, ~0.3 VALU instructions per MFMA instead of 4.8;
//   * three LDS stages with the DMA of tile kt+2 in flight across the one barrier per tile (counted vmcnt,
//     raw s_barrier: __syncthreads() would drain the DMA queue);
//   * k-contiguous operands: image [128 rows][BK bf16], 16-byte chunks XOR-swizzled on the SOURCE address so
//     that the fragment ds_read_b128s are bank-conflict free (the DMA writes LDS linearly: lane -> base + 16*lane);
//   * k-strided operands (dW = dz^T a, dX = dz W): image [BK k-rows][128 cols] and the hardware transpose read
//     ds_read_b64_tr_b16 -- no transposed copy in HBM or LDS.
// Tile 128x128, 4 wavefronts (2x2, 64x64 each = 2x2 MFMA tiles), BK 32 (one workgroup per CU: 3 x 48 KB) or
// BK 16 (two per CU: 3 x 24 KB, the backward dual launch).  The k order inside a tile is the natural one
// (k = 16 s + 8 h + j for lane half h, element j), the six products are issued in the round-1 order: results
// are bit-identical to round 1's gemm_body_planes.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

namespace plp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// Arithmetic modes of the planes GEMM (what the 16-bit planes hold and which products are accumulated):
//   kBf16    1 bf16 plane : a*b on the rounded operands (PL_BF16; bf16 STORAGE of activations and weights)
//   kBf16x6  3 bf16 planes: x = x0 + x1 + x2 exactly; a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0, one accumulator
//   kF16x3   2 fp16 planes: S x = h + l / 2048 with h = fp16(S x), l = fp16((S x - h) * 2048), S a per-tensor power
//            of two that keeps h in fp16's normal range (22-23 significant bits); a*b = h_a h_b + (h_a l_b +
//            l_a h_b) / 2048 on TWO accumulators (main, low), combined and un-scaled in the epilogue.  The dropped
//            l_a l_b term is < 2^-22 |ab|.  Three MFMAs per product term instead of six: on a chip whose matrix
//            clock under load makes six-product bf16 MFMA-bound (measured: 40 us for 51.5 GFLOP issued), this
//            halves the bound.  Error against fp64 on the golden eval forward: 6.4e-5 mm (exact products 6.0e-5,
//            bf16x6 5.9e-5) -- indistinguishable; the fp32 accumulation of either is what sets the floor.
enum PlanesMode { kBf16 = 0, kBf16x6 = 1, kF16x3 = 2 };
template <int MODE> struct ModeCfg;
template <> struct ModeCfg<kBf16>   { static constexpr int NPL = 1, NACC = 1, NPROD = 1; };
template <> struct ModeCfg<kBf16x6> { static constexpr int NPL = 3, NACC = 1, NPROD = 6; };
template <> struct ModeCfg<kF16x3>  { static constexpr int NPL = 2, NACC = 2, NPROD = 3; };
constexpr float kF16LoScale = 2048.0f;      // l is stored times 2^11

struct PlanesArgs {
  const __bf16* A;   // planes of A: k-contiguous [M][K] (lda = K) or k-strided [K][M] (lda = M)
  const __bf16* B;   // planes of B: k-contiguous [N][K] (ldb = K) or k-strided [K][N] (ldb = N)
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  int split_k;       // > 1: slice z covers K/split_k and writes C + z*M*ldc
  size_t a_plane, b_plane;   // elements between two planes
  // Implicit-GEMM convolution (16x16x32 loop; cv_cin == 0: none).  The gathered operand is the NHWC tensor x
  // [B][cv_h][cv_w][cv_cin] (planes), every 32-k tile inside ONE filter tap (cv_cin % 32 == 0), padding pixels read as
  // zeros (an out-of-range LDS-DMA lane writes zeros: tools/ubench/dma_oob_probe.hip):
  //   forward / data gradient (NT, gathered A): A = x, row m = output pixel (b, oh, ow), k = (kh*cv_kw + kw)*cv_cin + ci;
  //   weight gradient (TN, gathered B):          B = x, k = output pixel, column n = (kh*cv_kw + kw)*cv_cin + ci.
  // cv_stride: the stride along h, cv_stride_w along w (the stem's pixel-pair view has 2 and 1).  cv_cin == 8 (forward only:
  // cv_kw % 4 == 0): a 32-k tile is four neighbouring taps of one kernel row, 8 channels each.
  int cv_cin, cv_h, cv_w, cv_ho, cv_wo, cv_kw, cv_stride, cv_stride_w, cv_pad_h, cv_pad_w;
  int abl;           // timing-only ablations (POSELIFT_ABL, wrong results by construction): 2 = no operand DMA, 4 = no MFMAs
};

template <int BKX, int NPL, int NST = 3>
struct PlanesCfg {
  static constexpr int OPP = 128 * BKX * 2;     // one plane of one operand tile, bytes
  static constexpr int STAGE = 2 * NPL * OPP;   // A planes then B planes
  static constexpr int LDS = NST * STAGE;
  static constexpr int NJ = BKX / 16;           // DMA instructions per wave, plane and operand tile (1 KiB each)
  static constexpr int NDMA = 2 * NPL * NJ;     // per wave and K tile
};

// 16 bytes per lane, global -> LDS, no VGPR destination: LDS address = lp (wave-uniform) + 16 * lane
#define PLP_BLDS16(rsrc, lp, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(lp), 16, (voff), (soff), 0, 0)

// byte offset (from the tile origin of a plane) of the 16 bytes lane `lane` of DMA instruction `rb` fetches
template <bool KS, int BKX>
__device__ __forceinline__ uint32_t glds_lane_off(int rb, int lane, int ld) {
  if (!KS) {
    if (BKX == 32) {   // 16 rows x 64 B per instruction; chunk ^= (row >> 2) & 3
      const int row = rb * 16 + (lane >> 2);
      const int ch = (lane & 3) ^ ((lane >> 4) & 3);
      return (uint32_t)(row * ld * 2 + ch * 16);
    }
    const int row = rb * 32 + (lane >> 1);   // 32 rows x 32 B; chunk ^= (row >> 3) & 1
    const int ch = (lane & 1) ^ ((lane >> 4) & 1);
    return (uint32_t)(row * ld * 2 + ch * 16);
  }
  // 4 k-rows x 256 B per instruction; chunk ^= ((krow & 3) << 2) | ((krow >> 2) & 3)
  const int krow = rb * 4 + (lane >> 4);
  const int ch = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | (rb & 3));
  return (uint32_t)(krow * ld * 2 + ch * 16);
}

// per-lane LDS byte offsets (inside one plane of one operand tile) of the fragment reads
template <bool KS, int BKX>
struct FragAddr {
  uint32_t b[KS ? 4 : BKX / 16];
  // wq: the wave's 64-row (A) / 64-column (B) block of the tile
  __device__ __forceinline__ void init(int wq, int lane) {
    const int i = lane & 31, h = lane >> 5;
    if (!KS) {
      if (BKX == 32) {
#pragma unroll
        for (int s = 0; s < 2; ++s) b[s] = (uint32_t)((wq * 64 + i) * 64 + (((2 * s + h) ^ ((i >> 2) & 3)) << 4));
      } else {
        b[0] = (uint32_t)((wq * 64 + i) * 32 + ((h ^ ((i >> 3) & 1)) << 4));
      }
    } else {
      // ds_read_b64_tr_b16: lane 4q+pp of a 16-lane group supplies row q, columns 4pp..4pp+3 of a 4 x 16 block and
      // receives column (lane & 15) of its four rows.  Group g: columns 16 (g & 1) .. +15 of the 32-column MFMA
      // tile, k rows 8 (g >> 1) + 4 u .. +3 for read u.
      const int g1 = (lane >> 4) & 1, li = lane & 15, q = li >> 2, pp = li & 3;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int row = 8 * h + 4 * u + q;
          const int ch = wq * 8 + t * 4 + 2 * g1 + (pp >> 1);
          const int x = (q << 2) | ((2 * h + u) & 3);
          b[t * 2 + u] = (uint32_t)(256 * row + 16 * (ch ^ x) + 8 * (pp & 1));
        }
    }
  }
};

// fragment of 32x32 tile t (0, 1) of the wave's block, 16-deep k step s, from plane image `op` (LDS)
template <bool KS, int BKX>
__device__ __forceinline__ s16x8 read_frag_planes(const char* op, const FragAddr<KS, BKX>& fa, int t, int s) {
  if (!KS) {
    if (BKX == 32) return *reinterpret_cast<const s16x8*>(op + fa.b[s] + t * 2048);
    return *reinterpret_cast<const s16x8*>(op + fa.b[0] + t * 1024);
  }
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(op + fa.b[t * 2 + 0] + s * 4096));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(op + fa.b[t * 2 + 1] + s * 4096));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int MODE>
__device__ __forceinline__ f32x16 mfma16(const s16x8 a, const s16x8 b, const f32x16 c) {
  if constexpr (MODE == kF16x3)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// timing-only (ABL 4 / 5): the FLOPs of one 32x32x16 MFMA issued as two 16x16x32 ones on quarters of the accumulator --
// garbage arithmetic, the question is only which shape the chip clocks higher under this loop's load (guide: DVFS item 7)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma16_as_two_16x16x32(const s16x8 a, const s16x8 b, f32x16 c) {
  f32x4 lo = __builtin_shufflevector(c, c, 0, 1, 2, 3), hi = __builtin_shufflevector(c, c, 4, 5, 6, 7);
  lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), lo, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), hi, 0, 0, 0);
  c[0] = lo[0]; c[1] = lo[1]; c[2] = lo[2]; c[3] = lo[3];
  c[4] = hi[0]; c[5] = hi[1]; c[6] = hi[2]; c[7] = hi[3];
  return c;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Issue order of one half step (a single basic block): MFMA, 2 fragment reads, MFMA, 2 reads, ... then MFMA, 1 DMA,
// MFMA, 1 DMA, ...: the matrix pipe never waits for a burst of memory instructions to issue.
template <int NDS, int NVM, int NMF>
__device__ __forceinline__ void sched_half() {
  constexpr int QV0 = (NDS + 1) / 2;   // first MFMA slot that carries a DMA issue
#pragma unroll
  for (int q = 0; q < NMF; ++q) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (2 * q < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    if (q >= QV0 && q < QV0 + NVM) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
  }
  // whatever did not fit under the MFMAs (NPL = 1: four MFMAs per half)
  if (2 * NMF < NDS) __builtin_amdgcn_sched_group_barrier(0x100, NDS - 2 * NMF, 0);
  if (QV0 + NVM > NMF && NVM > 0) __builtin_amdgcn_sched_group_barrier(0x010, NVM, 0);
}

// Whole tiles only: M, N % 128 == 0, every K slice a multiple of BKX, lda / ldb % 8 == 0, planes 16-byte aligned.
// acc[a][b][r]: row = m0 + wm*64 + a*32 + (r&3) + 8*(r>>2) + 4*h, col = n0 + wn*64 + b*32 + i  (wm = wave>>1, wn = wave&1)
//
// NLW = 0: 256 threads, every wave issues its share of the DMA between its MFMAs.
// NLW = 4: 512 threads; waves 0-3 compute (ds_read + MFMA only), waves 4-7 do nothing but issue the DMA and wait
//          for it.  An LDS-DMA instruction costs its issuing wave 60-185 cycles (MI355X guide, cycle constants);
//          a lone wave per SIMD issues in order, so with NLW = 0 every DMA issue is matrix-pipe idle time (measured:
//          1.26 us per 32-k tile against 0.81 us of MFMA time).  Waves w and w+4 share a SIMD: the loader's stall
//          costs the computing wave nothing.  acc is meaningful in waves 0-3 only (`return false` for loaders).
// ABL (timing-only builds of the microbenchmark): 1 no MFMAs, 2 no DMA, 3 DMA only, 4 the MFMAs as 16x16x32, 5 = 2 + 4.
template <bool A_KS, bool B_KS, int BKX, int MODE, int NLW = 0, int ABL = 0, int NST = 3>
__device__ __forceinline__ bool planes_mainloop(const PlanesArgs& p, const int block_id, const int nwork,
                                                char* __restrict__ lds, f32x16 (&acc)[ModeCfg<MODE>::NACC][2][2],
                                                int& m0, int& n0, int& slice) {
  constexpr int NPL = ModeCfg<MODE>::NPL, NACC = ModeCfg<MODE>::NACC;
  using Cf = PlanesCfg<BKX, NPL, NST>;
  static_assert(NST == 3 || NST == 4, "three or four LDS stages");
  constexpr int AHEAD = NST - 1;            // tiles the DMA runs ahead of the MFMAs
  constexpr int KSTEPS = BKX / 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = NLW > 0 && wave >= 4;
  const int lw = wave & 3;                     // this wave's share of the DMA instructions
  const int wm = (wave & 3) >> 1, wn = wave & 1;

  const int tiles_n = p.N / 128;
  const int splits = p.split_k > 1 ? p.split_k : 1;
  const int ntiles = nwork / splits;
  int w = block_id;
  if ((nwork & 7) == 0) w = (w & 7) * (nwork >> 3) + (w >> 3);   // XCD-aware: blocks b and b+8 share an L2
  slice = w / ntiles;
  const int t = w - slice * ntiles;
  m0 = (t / tiles_n) * 128;
  n0 = (t % tiles_n) * 128;
  int kbeg = 0, kend = p.K;
  if (splits > 1) {
    const int per = ((p.K / BKX + splits - 1) / splits) * BKX;
    kbeg = min(slice * per, p.K);
    kend = min(kbeg + per, p.K);
  }
  const int nk = (kend - kbeg) / BKX;

#pragma unroll
  for (int c = 0; c < NACC; ++c)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][a][b][r] = 0.f;
  if (nk <= 0) return !loader;

  // ---- DMA source addressing: buffer descriptor at the tile origin (SGPRs) + per-lane 32-bit byte offset (a
  // loop-invariant VGPR) + scalar offset (K tile, plane): `buffer_load_dwordx4 v, s[0:3], s offen lds`, i.e. NO
  // vector-ALU address arithmetic in the loop (the flat global_load_lds form cost two 64-bit adds per load).
  const __bf16* ta = A_KS ? p.A + (size_t)kbeg * p.lda + m0 : p.A + (size_t)m0 * p.lda + kbeg;
  const __bf16* tb = B_KS ? p.B + (size_t)kbeg * p.ldb + n0 : p.B + (size_t)n0 * p.ldb + kbeg;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(ta), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(tb), 0, 0x7fffffff, 0x00020000);
  const int ga_step = A_KS ? BKX * p.lda * 2 : BKX * 2;
  const int gb_step = B_KS ? BKX * p.ldb * 2 : BKX * 2;
  const int apl = (int)(p.a_plane * 2), bpl = (int)(p.b_plane * 2);
  int oa[Cf::NJ], ob[Cf::NJ];
#pragma unroll
  for (int j = 0; j < Cf::NJ; ++j) {
    oa[j] = (int)glds_lane_off<A_KS, BKX>(lw + 4 * j, lane, p.lda);
    ob[j] = (int)glds_lane_off<B_KS, BKX>(lw + 4 * j, lane, p.ldb);
  }
  auto issue = [&](const int kt, const int stage_off) {
    if (ABL == 2 || ABL == 5) return;
    const int sa = kt * ga_step, sb = kt * gb_step;
    char* d = lds + stage_off + lw * 1024;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int j = 0; j < Cf::NJ; ++j) {
        PLP_BLDS16(ra, d + pl * Cf::OPP + j * 4096, oa[j], sa + pl * apl);
        PLP_BLDS16(rb, d + (NPL + pl) * Cf::OPP + j * 4096, ob[j], sb + pl * bpl);
      }
  };
  auto barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // stage of tile t: (t % NST) * STAGE, kept as rotating offsets (no division in the loop)
  int st[NST];
#pragma unroll
  for (int q = 0; q < NST; ++q) st[q] = q * Cf::STAGE;       // st[q] = stage of tile kt + q
  auto rotate = [&]() {
    const int o = st[0];
#pragma unroll
    for (int q = 0; q + 1 < NST; ++q) st[q] = st[q + 1];
    st[NST - 1] = o;
  };

  if (NLW > 0 && loader) {
    // ---- loader waves: tile kt+AHEAD is issued while the computing waves work on tile kt; the barrier of step kt
    // tells them tile kt+1 has landed and tells us the stage of tile kt is free again
#pragma unroll
    for (int q = 0; q < AHEAD; ++q)
      if (q < nk) issue(q, st[q]);
    // tile 0 must have landed: at most the AHEAD-1 younger tiles may still be in flight
    if (nk >= AHEAD) wait_vmcnt<(AHEAD - 1) * Cf::NDMA>(); else wait_vmcnt<0>();
    barrier();
    for (int kt = 0; kt < nk; ++kt) {
      // tile kt+1 must have landed before the barrier of step kt
      if (kt + AHEAD < nk) { issue(kt + AHEAD, st[AHEAD]); wait_vmcnt<(AHEAD - 1) * Cf::NDMA>(); }
      else if (AHEAD == 3 && kt + 2 < nk) { wait_vmcnt<Cf::NDMA>(); }
      else { wait_vmcnt<0>(); }
      barrier();
      rotate();
    }
    return false;
  }

  FragAddr<A_KS, BKX> fra;
  FragAddr<B_KS, BKX> frb;
  fra.init(wm, lane);
  frb.init(wn, lane);

  s16x8 fa[2][2][NPL], fb[2][2][NPL];   // [set][32x32 tile][plane]
#define PLP_FRAGS(set, stage_off, s)                                                                     \
  do {                                                                                                   \
    if (ABL == 3) break;                                                                                 \
    const char* qa_ = lds + (stage_off);                                                                 \
    const char* qb_ = qa_ + NPL * Cf::OPP;                                                               \
    _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2)                                                     \
    _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                                                 \
      fa[set][t2][pl] = read_frag_planes<A_KS, BKX>(qa_ + pl * Cf::OPP, fra, t2, (s));                   \
      fb[set][t2][pl] = read_frag_planes<B_KS, BKX>(qb_ + pl * Cf::OPP, frb, t2, (s));                   \
    }                                                                                                    \
  } while (0)
#define PLP_MF(c, set, aa, bb, ia, ib)                                                                   \
  do {                                                                                                   \
    if (ABL == 0 || ABL == 2)                                                                            \
      acc[c][aa][bb] = mfma16<MODE>(fa[set][aa][ia], fb[set][bb][ib], acc[c][aa][bb]);                    \
    else if (ABL == 1) asm volatile("" ::"v"(fa[set][aa][ia]), "v"(fb[set][bb][ib]));                    \
    else if (ABL == 4 || ABL == 5) acc[c][aa][bb] = mfma16_as_two_16x16x32(fa[set][aa][ia], fb[set][bb][ib], acc[c][aa][bb]); \
  } while (0)
  // products of one 32x32 output tile and one 16-deep k step (bf16x6: smallest terms first, the round-1 order)
#define PLP_MFS(set, aa, bb)                                                                     \
  do {                                                                                           \
    if constexpr (MODE == kBf16x6) {                                                             \
      PLP_MF(0, set, aa, bb, 2, 0); PLP_MF(0, set, aa, bb, 1, 1); PLP_MF(0, set, aa, bb, 0, 2);  \
      PLP_MF(0, set, aa, bb, 1, 0); PLP_MF(0, set, aa, bb, 0, 1);                                \
    }                                                                                            \
    if constexpr (MODE == kF16x3) {                                                              \
      PLP_MF(1, set, aa, bb, 0, 1); PLP_MF(1, set, aa, bb, 1, 0);                                \
    }                                                                                            \
    PLP_MF(0, set, aa, bb, 0, 0);                                                                \
  } while (0)
#define PLP_ROW(set, aa) do { PLP_MFS(set, aa, 0); PLP_MFS(set, aa, 1); } while (0)

  constexpr int NDS = 2 * NPL * ((A_KS ? 2 : 1) + (B_KS ? 2 : 1));   // ds_reads per fragment set
  constexpr int NMF = ModeCfg<MODE>::NPROD * (KSTEPS == 2 ? 4 : 2);  // MFMAs per half step
  constexpr int NVM = NLW > 0 ? 0 : Cf::NDMA;                        // DMA issues of a computing wave per tile
  static_assert(NLW > 0 || NST == 3, "self-issuing waves run two tiles ahead");
  // One tile.  BK 32: first half = k step 0 (set 0), second half = k step 1 (set 1).  BK 16: the halves are the two
  // 32-row MFMA rows of fragment set P (tile parity).  In the middle: wait for tile kt+1's DMA (issued a whole
  // tile ago), barrier, fetch the next tile's first fragments.
  auto step = [&](const int kt, auto par, auto steady) {
    constexpr int P = decltype(par)::value;
    constexpr bool STEADY = decltype(steady)::value;      // no branches inside the steady-state step
    const bool do_issue = STEADY || kt + 2 < nk, has_next = STEADY || kt + 1 < nk;
    // (program order: the fragment reads BEFORE the DMA issue -- the compiler keeps LDS reads behind an earlier
    //  LDS-DMA, and the reads are wanted first)
    if (KSTEPS == 2) PLP_FRAGS(1, st[0], 1);
    if (NLW == 0 && do_issue) issue(kt + 2, st[2]);
    if (KSTEPS == 2) {
      PLP_ROW(0, 0); PLP_ROW(0, 1);
    } else {
      PLP_ROW(P, 0);
    }
    // steady state, first half: the fragment reads two per MFMA shadow, then one DMA issue per MFMA shadow
    if (STEADY && ABL == 0) sched_half<KSTEPS == 2 ? NDS : 0, NVM, NMF>();
    __builtin_amdgcn_sched_barrier(0);
    if (NLW == 0 && ABL != 2 && ABL != 5) {
      if (do_issue) wait_vmcnt<Cf::NDMA>(); else wait_vmcnt<0>();   // tile kt+1 (issued a whole tile ago) has landed
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this tile's fragment reads have left LDS (WAR on the stage)
    barrier();
    if (KSTEPS == 2) {
      if (has_next) PLP_FRAGS(0, st[1], 0);
      PLP_ROW(1, 0); PLP_ROW(1, 1);
    } else {
      if (has_next) PLP_FRAGS(1 - P, st[1], 0);
      PLP_ROW(P, 1);
    }
    if (STEADY && ABL == 0) sched_half<NDS, 0, NMF>();
    __builtin_amdgcn_sched_barrier(0);
    rotate();
  };

  if (NLW == 0) {
    issue(0, st[0]);
    if (nk > 1) issue(1, st[1]);
    if (ABL != 2 && ABL != 5) { if (nk > 1) wait_vmcnt<Cf::NDMA>(); else wait_vmcnt<0>(); }
  }
  barrier();
  PLP_FRAGS(0, st[0], 0);
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  int kt = 0;
  for (; kt + 3 < nk; kt += 2) {
    step(kt, P0{}, std::true_type{});
    step(kt + 1, P1{}, std::true_type{});
  }
  for (; kt < nk; kt += 2) {
    step(kt, P0{}, std::false_type{});
    if (kt + 1 < nk) step(kt + 1, P1{}, std::false_type{});
  }
#undef PLP_FRAGS
#undef PLP_MF
#undef PLP_MFS
#undef PLP_ROW
  return true;
}

}  // namespace plp
