// The planes GEMM main loop on v_mfma_f32_16x16x32 (gemm_planes.h has the 32x32x16 form and everything shared).
//
// Why a second shape: under this loop's load the chip holds a higher matrix clock on the 16x16x32 shape (the guide's
// DVFS note; measured here with a timing-only build that issues each 32x32x16 MFMA as two 16x16x32 ones on the same
// registers: 30.0 -> 27.4 us per 4096x1024x1024 f16x3 GEMM, MFMA-only 23.9 -> 22.5 us).  Same tile (128x128, BK 32,
// three LDS stages, four computing + four loader wavefronts), same DMA, same planes; what changes is the fragment
// geometry, the accumulator layout (and with it the epilogue, which is LDS-staged throughout: gemm_planes.hip).
//
//   v_mfma_f32_16x16x32_{f16,bf16}: lane l holds A[row l&15][k = 8 (l>>4) + j], B[k = 8 (l>>4) + j][col l&15], j < 8;
//   D: col l&15, row 4 (l>>4) + reg, reg < 4.  A wave's 64x64 block = 4x4 such tiles: acc[4][4] of 4 floats.
//   One BK-32 tile is ONE k step: per plane 4 A fragments + 4 B fragments (one ds_read_b128, or two transposing
//   ds_read_b64_tr_b16, each) feed 16 MFMAs per product.
// Registers (256 per wave with eight waves per workgroup): 2 x 64 accumulators (f16x3) leave room for one A fragment
// set and two B sets, so the prefetch is staggered: A[2,3] of tile kt is read under the MFMAs of A[0,1], and after the
// mid-step barrier A[0,1] and B of tile kt+1 are read under the MFMAs of A[2,3].
// (Tried: the loader waves, idle after their last DMA issue, touching every line the dX epilogue will load -- skip
//  gradient, saved z, bitmap -- two K tiles ahead, as the youngest loads of their queue: 0.6175 / 0.6156 ms per step
//  against 0.6104 / 0.6197 without, same box.  The epilogue is not waiting on HBM latency; not kept.  The other end of the
//  main loop -- every computing wave touching its block's lines BEFORE the loop, by LDS-DMA into a scratch KiB (no register,
//  no wait), so that the burst would find them in the memory-side cache: 0.6395 / 0.6349 against 0.6269 / 0.6221: worse.)
#pragma once
#include "gemm_planes.h"

namespace plp {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// k-contiguous image [128 rows][64 B]: the DMA writes it linearly, the 16-byte chunk a lane FETCHES is XORed with
// f(row) = (-(row >> 2)) & 3 -- for the 16x16x32 read pattern (16 consecutive rows x the 4 chunks per wave instruction)
// every 16-lane group of ds_read_b128 then covers the 16 slots of the 256-byte bank row exactly once.
__device__ __forceinline__ int kc16_swz(int row) { return (0 - (row >> 2)) & 3; }

// `left`: rows (k-contiguous image) / columns (k-strided image) of the matrix from the tile origin on.  A tile that hangs
// over the edge of its matrix re-reads the last valid row / 8-column chunk in the overhang (the LDS slot a lane fills is
// fixed by its lane id; only the SOURCE is clamped): the MFMAs compute values the guarded epilogue never stores.
template <bool KS>
__device__ __forceinline__ uint32_t glds_lane_off16(int rb, int lane, int ld, int left = 0x7fffffff) {
  if (!KS) {   // 16 rows x 64 B per instruction
    const int row = rb * 16 + (lane >> 2);
    const int ch = (lane & 3) ^ kc16_swz(row);
    return (uint32_t)(min(row, left - 1) * ld * 2 + ch * 16);
  }
  // k-strided image: as the 32x32x16 loop (the guide's layout (b)): 4 k-rows x 256 B per instruction
  const int krow = rb * 4 + (lane >> 4);
  const int ch = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | (rb & 3));
  return (uint32_t)(krow * ld * 2 + min(ch, (left >> 3) - 1) * 16);
}

// per-lane LDS byte offsets of the fragment reads inside one plane of one operand tile
template <bool KS>
struct FragAddr16 {
  uint32_t b[KS ? 8 : 1];
  __device__ __forceinline__ void init(int wq, int lane) {
    const int r = lane & 15, q = lane >> 4;
    if (!KS) {
      // row wq*64 + t*16 + r (t adds 1024 bytes: (16 rows >> 2) & 3 == 0, the swizzle does not depend on t), chunk q
      const int row = wq * 64 + r;
      b[0] = (uint32_t)(row * 64 + ((q ^ kc16_swz(row)) << 4));
    } else {
      // group g = lane >> 4 reads k rows 8g + 4u .. +3 (u = 0, 1) of the tile's 16 columns c0 .. c0+15; lane 4qq+pp of the
      // group supplies row qq, columns 4pp .. 4pp+3 and receives column (lane & 15)
      const int qq = r >> 2, pp = r & 3;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int row = 8 * q + 4 * u + qq;
          const int ch = wq * 8 + t * 2 + (pp >> 1);
          const int x = ((row & 3) << 2) | ((row >> 2) & 3);
          b[t * 2 + u] = (uint32_t)(256 * row + 16 * (ch ^ x) + 8 * (pp & 1));
        }
    }
  }
};

template <bool KS>
__device__ __forceinline__ s16x8 read_frag16(const char* op, const FragAddr16<KS>& fa, int t) {
  if (!KS) return *reinterpret_cast<const s16x8*>(op + fa.b[0] + t * 1024);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(op + fa.b[t * 2 + 0]));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(op + fa.b[t * 2 + 1]));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int MODE>
__device__ __forceinline__ f32x4v mfma16x16(const s16x8 a, const s16x8 b, const f32x4v c) {
  if constexpr (MODE == kF16x3)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// acc[c][rt][ct][reg]: row = m0 + wm*64 + rt*16 + 4*(lane>>4) + reg, col = n0 + wn*64 + ct*16 + (lane&15)
// Whole tiles only (as planes_mainloop); 512 threads; the loader waves never call `epi`.
// EDGE: M and N need not be multiples of 128 (M: any for a k-contiguous A, % 8 for a k-strided one; N % 8): see
// glds_lane_off16.  K stays a whole number of 32-k tiles per slice.
// CONV = 1 (with !A_KS, !B_KS): A gathered as an implicit-GEMM convolution input; CONV = 2 (A_KS, B_KS): B gathered as the
// weight gradient's input (PlanesArgs::cv_*).  Only the loader waves change: per-lane source offsets from the pixel
// decomposition, the tap's validity per lane, out-of-range (zero-filling) offsets for padding pixels.
//
// PERSIST (round 3): the workgroup walks a contiguous run of work items (output tile x K slice) instead of one, and the
// operand stream does not stop at an item boundary: the loader waves issue the k-tiles of item i+1 into the three-stage
// ring while the computing waves are still in item i's last steps and in its epilogue (which stages through LDS of its own,
// outside the ring).  What a one-item workgroup pays in series -- first-tile latency from HBM, the MFMAs, the 64 KB store --
// overlaps here; the short-K convolution layers (K = 64 .. 256 at 1M pixels: two to eight k-tiles per item) are the ones
// that were spending most of their time in exactly those two ends.  One barrier per k-tile for every wave, as before;
// `epi(acc, m0, n0, slice)` is called by the computing waves once per item.
constexpr int kDmaOutOfRange = 0x7ffffff0;     // >= num_records of the gathered operand's descriptor
// NST: stages of the operand ring (3: a k-tile being read, one landed, one in flight; the persistent form of the short-K
// layers runs deeper -- every barrier waits for the NEXT k-tile, so the ring depth is what HBM latency is hidden behind).
template <bool A_KS, bool B_KS, int MODE, bool EDGE, int CONV, bool PERSIST, int NST, bool T64, class Epi>
__device__ __forceinline__ void planes_run16(const PlanesArgs& p, const int block_id, const int nblocks, const int nwork,
                                             char* __restrict__ lds, Epi&& epi) {
  static_assert(MODE == kF16x3 || MODE == kBf16, "two fp16 planes or one bf16 plane");
  constexpr int NPL = ModeCfg<MODE>::NPL, NACC = ModeCfg<MODE>::NACC;
  // T64 (NT only): a 256-row x 64-column tile -- four computing waves stacked along M, each still a 64 x 64 block (so the
  // epilogue is unchanged) -- for the 64-channel layers, where half of every 128-wide tile was padding: half the MFMAs,
  // no clamped B rows.  A image [256 rows][64 B], B image [64 rows][64 B] per plane and stage.
  static_assert(!T64 || (!A_KS && !B_KS), "the 256 x 64 tile is an NT tile");
  constexpr int TM = T64 ? 256 : 128, TNW = T64 ? 64 : 128;
  constexpr int OPPA = (A_KS ? 128 : TM) * 64, OPPB = (B_KS ? 128 : TNW) * 64;      // bytes per plane, operand and stage
  constexpr int STAGE = NPL * (OPPA + OPPB);
  constexpr int JA = OPPA / 4096, JB = OPPB / 4096;                                  // DMA instructions per loader wave and plane
  constexpr int NDMA = NPL * (JA + JB);
  static_assert(NST >= 3 && (NST - 2) * NDMA <= 63, "ring depth: vmcnt counts at most 63 requests");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int lw = wave & 3;
  const int wm = T64 ? (wave & 3) : (wave & 3) >> 1, wn = T64 ? 0 : (wave & 1);

  const int tiles_n = EDGE ? (p.N + TNW - 1) / TNW : p.N / TNW;
  const int splits = p.split_k > 1 ? p.split_k : 1;
  const int ntiles = nwork / splits;
  // this workgroup's items: one (the block id, XCD-aware: blocks b and b+8 share an L2), or a contiguous run of them
  // (consecutive items are the column tiles of one row block: its A rows are re-read from L2, not from HBM)
  int item_lo, item_hi;
  if (PERSIST) {
    const int per = (nwork + nblocks - 1) / nblocks;
    item_lo = min(block_id * per, nwork);
    item_hi = min(item_lo + per, nwork);
  } else {
    item_lo = block_id;
    if ((nwork & 7) == 0) item_lo = (block_id & 7) * (nwork >> 3) + (block_id >> 3);
    item_hi = item_lo + 1;
  }
  if (item_lo >= item_hi) return;
  const int per_slice = ((p.K / 32 + splits - 1) / splits) * 32;
  // item -> (m0, n0, slice, first k, k-tiles)
  auto decode = [&](const int w, int& m0, int& n0, int& slice, int& kbeg) {
    slice = w / ntiles;
    const int t = w - slice * ntiles;
    m0 = (t / tiles_n) * TM;
    n0 = (t % tiles_n) * TNW;
    kbeg = 0;
    int kend = p.K;
    if (splits > 1) {
      kbeg = min(slice * per_slice, p.K);
      kend = min(kbeg + per_slice, p.K);
    }
    return (kend - kbeg) / 32;
  };

  const int ga_step = A_KS ? 32 * p.lda * 2 : 64;
  const int gb_step = B_KS ? 32 * p.ldb * 2 : 64;
  auto barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // stage of k-tile g of the stream: (g % NST) * STAGE, kept as running offsets
  auto next_stage = [](const int o) { return o + STAGE == NST * STAGE ? 0 : o + STAGE; };
  int st[2] = {0, STAGE};                       // computing waves: the k-tile being read, the next one
  auto rotate = [&]() { st[0] = st[1]; st[1] = next_stage(st[1]); };

  if (loader) {
    static_assert(CONV == 0 || (CONV == 1 && !A_KS && !B_KS) || (CONV == 2 && A_KS && B_KS), "conv gathers: NT forward, TN wgrad");
    // total k-tiles of the stream (the computing waves run one barrier per k-tile of the same items)
    int total = 0;
    if (splits == 1) total = (item_hi - item_lo) * (p.K / 32);
    else for (int w = item_lo; w < item_hi; ++w) { int a_, b_, c_, d_; total += max(decode(w, a_, b_, c_, d_), 0); }
    // ---- state of the item whose k-tiles are being issued
    int it = item_lo, kt = 0, nk = 0, m0 = 0, n0 = 0, slice = 0, kbeg = 0;
    // one descriptor per plane (64-bit plane stride in the base: a plane of the conv head's 4.5 GB gradient is 2.3 GB away)
    __amdgpu_buffer_rsrc_t ra[NPL], rb[NPL];
    int oa[JA], ob[JB];
    // ---- convolution gathers: descriptors on the whole tensor, offsets rebuilt per K tile by the (otherwise idle) loader VALU
    __amdgpu_buffer_rsrc_t rx[NPL];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
      rx[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(CONV == 2 ? p.B + (size_t)pl * p.b_plane : p.A + (size_t)pl * p.a_plane),
                                                 0, 0x7fffffe0, 0x00020000);
    int cih0[JA] = {}, ciw0[JA] = {}, cbase[JA] = {};               // CONV 1: per DMA row of this lane
    int ctap = 0, cc0 = 0, ckh = 0, ckw = 0;                        // CONV 1: filter tap / channel offset of the next tile issued
    int wkh = 0, wkw = 0, wcol = 0;                                 // CONV 2: this lane's column chunk: tap and channel (fixed)
    auto setup = [&]() {
      nk = decode(it, m0, n0, slice, kbeg);
      const __bf16* ta = A_KS ? p.A + (size_t)kbeg * p.lda + m0 : p.A + (size_t)m0 * p.lda + kbeg;
      const __bf16* tb = B_KS ? p.B + (size_t)kbeg * p.ldb + n0 : p.B + (size_t)n0 * p.ldb + kbeg;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        ra[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(ta + (size_t)pl * p.a_plane), 0, 0x7fffffff, 0x00020000);
        rb[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(tb + (size_t)pl * p.b_plane), 0, 0x7fffffff, 0x00020000);
      }
#pragma unroll
      for (int j = 0; j < JA; ++j) oa[j] = (int)glds_lane_off16<A_KS>(lw + 4 * j, lane, p.lda, EDGE ? p.M - m0 : 0x7fffffff);
#pragma unroll
      for (int j = 0; j < JB; ++j) ob[j] = (int)glds_lane_off16<B_KS>(lw + 4 * j, lane, p.ldb, EDGE ? p.N - n0 : 0x7fffffff);
      if (CONV == 1) {
#pragma unroll
        for (int j = 0; j < JA; ++j) {
          const int row = (lw + 4 * j) * 16 + (lane >> 2);
          const int m = min(m0 + row, p.M - 1);
          const int ow = m % p.cv_wo, t2 = m / p.cv_wo;
          const int oh = t2 % p.cv_ho, b = t2 / p.cv_ho;
          cih0[j] = oh * p.cv_stride - p.cv_pad_h;
          ciw0[j] = ow * p.cv_stride_w - p.cv_pad_w;
          const int q = (lane & 3) ^ kc16_swz(row);          // the 16-byte chunk of the 64-byte tile row this lane fetches
          if (p.cv_cin == 8) {
            // 8-channel input (the stem's pixel pairs): the row's four chunks are four neighbouring taps (kw .. kw + 3) of
            // one kernel row, so the lane's own tap decides its source pixel and its padding test
            ciw0[j] += q;
            cbase[j] = ((b * p.cv_h + cih0[j]) * p.cv_w + ciw0[j]) * 16;
          } else {
            cbase[j] = ((b * p.cv_h + cih0[j]) * p.cv_w + ciw0[j]) * p.cv_cin * 2 + (q << 4);
          }
        }
        if (p.cv_cin == 8) {
          ctap = kbeg / 8; cc0 = 0;
          ckh = ctap / p.cv_kw; ckw = ctap - ckh * p.cv_kw;
        } else {
          ctap = kbeg / p.cv_cin; cc0 = kbeg - ctap * p.cv_cin;
          ckh = ctap / p.cv_kw; ckw = ctap - ckh * p.cv_kw;
        }
      }
      if (CONV == 2) {
        // the lane's 16-byte chunk = 8 consecutive columns n = (tap, ci .. ci+7) of the [k][N] image (cv_cin % 8 == 0)
        const int ch = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | (lw & 3));
        const int n = min(n0 + ch * 8, p.N - 8);
        const int tap = n / p.cv_cin;
        wcol = n - tap * p.cv_cin;
        wkh = tap / p.cv_kw; wkw = tap - wkh * p.cv_kw;
      }
    };
    auto issue = [&](const int stage_off) {
      if (p.abl & 2) return;
      const int sa = kt * ga_step, sb = kt * gb_step;
      char* d = lds + stage_off + lw * 1024;
      int va[JA], vb[JB];
#pragma unroll
      for (int j = 0; j < JA; ++j) va[j] = oa[j];
#pragma unroll
      for (int j = 0; j < JB; ++j) vb[j] = ob[j];
      if (CONV == 1) {                                   // tiles are issued in K order: the tap advances incrementally
        const int toff = ((ckh * p.cv_w + ckw) * p.cv_cin + cc0) * 2;
#pragma unroll
        for (int j = 0; j < JA; ++j) {
          const bool ok = (unsigned)(cih0[j] + ckh) < (unsigned)p.cv_h && (unsigned)(ciw0[j] + ckw) < (unsigned)p.cv_w;
          va[j] = ok ? cbase[j] + toff : kDmaOutOfRange;
        }
        if (p.cv_cin == 8) {
          ckw += 4;                                        // (four taps per tile; cv_kw % 4 == 0)
          if (ckw >= p.cv_kw) { ckw = 0; ++ckh; }
        } else {
          cc0 += 32;
          if (cc0 >= p.cv_cin) { cc0 = 0; if (++ckw == p.cv_kw) { ckw = 0; ++ckh; } }
        }
      }
      if (CONV == 2) {
        // k rows = output pixels kbeg + 32 kt + 4 (lw + 4 j) + (lane >> 4): pixel -> (b, oh, ow) -> the tap's input pixel
#pragma unroll
        for (int j = 0; j < JB; ++j) {
          const int pix = kbeg + kt * 32 + (lw + 4 * j) * 4 + (lane >> 4);
          const int ow = pix % p.cv_wo, t2 = pix / p.cv_wo;
          const int oh = t2 % p.cv_ho, b = t2 / p.cv_ho;
          const int ih = oh * p.cv_stride - p.cv_pad_h + wkh, iw = ow * p.cv_stride_w - p.cv_pad_w + wkw;
          const bool ok = (unsigned)ih < (unsigned)p.cv_h && (unsigned)iw < (unsigned)p.cv_w;
          vb[j] = ok ? (((b * p.cv_h + ih) * p.cv_w + iw) * p.cv_cin + wcol) * 2 : kDmaOutOfRange;
        }
      }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
        for (int j = 0; j < JA; ++j) {
          if (CONV == 1) PLP_BLDS16(rx[pl], d + pl * OPPA + j * 4096, va[j], 0);
          else PLP_BLDS16(ra[pl], d + pl * OPPA + j * 4096, va[j], sa);
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
          if (CONV == 2) PLP_BLDS16(rx[pl], d + NPL * OPPA + pl * OPPB + j * 4096, vb[j], 0);
          else PLP_BLDS16(rb[pl], d + NPL * OPPA + pl * OPPB + j * 4096, vb[j], sb);
        }
      }
    };
    // the stream: items in order, each one's k-tiles in order; an item without k-tiles (a K slice past the end) is skipped,
    // as the computing waves skip it
    auto open_item = [&]() {
      for (; it < item_hi; ++it) {
        setup();
        if (nk > 0) { kt = 0; return true; }
      }
      return false;
    };
    int sw = 0;                                       // stage the next k-tile is issued into
    auto issue_next = [&]() {                         // called exactly `total` times
      if (kt >= nk) { ++it; if (!open_item()) return; }
      issue(sw);
      sw = next_stage(sw);
      ++kt;
    };
    // at most n k-tiles (the youngest) still in flight
    auto wait_tiles = [&](const int n) {
      if (NST >= 7 && n >= 5) wait_vmcnt<5 * NDMA>();
      else if (NST >= 6 && n == 4) wait_vmcnt<4 * NDMA>();
      else if (NST >= 5 && n == 3) wait_vmcnt<3 * NDMA>();
      else if (NST >= 4 && n == 2) wait_vmcnt<2 * NDMA>();
      else if (n == 1) wait_vmcnt<NDMA>();
      else wait_vmcnt<0>();
    };
    static_assert(NST <= 7, "wait_tiles covers rings of up to seven stages");
    if (total <= 0 || !open_item()) return;
    const int pre = min(total, NST - 1);
    for (int i = 0; i < pre; ++i) issue_next();
    wait_tiles(pre - 1);                              // k-tile 0 has landed
    barrier();
    for (int g = 0; g < total; ++g) {
      if (g + NST - 1 < total) issue_next();          // into the stage k-tile g-1 left at the previous barrier
      wait_tiles(max(min(g + NST - 1, total - 1) - (g + 1), 0));      // k-tile g+1 has landed
      barrier();
    }
    return;
  }

  FragAddr16<A_KS> fra;
  FragAddr16<B_KS> frb;
  fra.init(wm, lane);
  frb.init(wn, lane);
  s16x8 fa[4][NPL];          // A fragments of the tile in flight (row tiles 0..3)
  s16x8 fb[2][4][NPL];       // B fragments: set P of tile kt, set 1-P of tile kt+1
  f32x4v acc[NACC][4][4];

#define PLP16_READ_A(stage_off, t0, t1)                                                      \
  do {                                                                                       \
    const char* q_ = lds + (stage_off);                                                      \
    _Pragma("unroll") for (int t2 = (t0); t2 < (t1); ++t2)                                   \
    _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                                       \
      fa[t2][pl] = read_frag16<A_KS>(q_ + pl * OPPA, fra, t2);                            \
  } while (0)
#define PLP16_READ_B(set, stage_off)                                                         \
  do {                                                                                       \
    const char* q_ = lds + (stage_off) + NPL * OPPA;                                         \
    _Pragma("unroll") for (int t2 = 0; t2 < 4; ++t2)                                         \
    _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                                       \
      fb[set][t2][pl] = read_frag16<B_KS>(q_ + pl * OPPB, frb, t2);                       \
  } while (0)
  // one 16x16 output tile, one 32-deep k step: f16x3 = main product on accumulator 0, the two cross terms on 1
#define PLP16_MFS(set, rt, ct)                                                                         \
  do {                                                                                                 \
    if constexpr (MODE == kF16x3) {                                                                    \
      acc[1][rt][ct] = mfma16x16<MODE>(fa[rt][0], fb[set][ct][1], acc[1][rt][ct]);                      \
      acc[1][rt][ct] = mfma16x16<MODE>(fa[rt][1], fb[set][ct][0], acc[1][rt][ct]);                      \
    }                                                                                                  \
    acc[0][rt][ct] = mfma16x16<MODE>(fa[rt][0], fb[set][ct][0], acc[0][rt][ct]);                        \
  } while (0)
#define PLP16_ROWS(set, r0, r1)                                                              \
  do {                                                                                       \
    if (p.abl & 4) break;                                                                    \
    _Pragma("unroll") for (int rt = (r0); rt < (r1); ++rt)                                   \
    _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) PLP16_MFS(set, rt, ct);                 \
  } while (0)

  constexpr int RA = NPL * (A_KS ? 2 : 1), RB = NPL * (B_KS ? 2 : 1);     // ds_reads per fragment tile
  constexpr int NMF = 8 * ModeCfg<MODE>::NPROD;                           // MFMAs per half step
  int nk = 0;
  const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
  // (Tried: the item's first k-tile as a step of its own whose MFMAs take a zero C operand, instead of 128 accumulator moves
  //  per item -- a fifth instance of the step body: hipcc then spills 360-650 VGPRs in the persistent kernels.  Not kept.)
  auto step = [&](const int kt, auto par, auto steady) {
    constexpr int P = decltype(par)::value;
    constexpr bool STEADY = decltype(steady)::value;
    const bool has_next = STEADY || kt + 1 < nk;
    PLP16_READ_A(st[0], 2, 4);                 // this tile's lower row tiles, under the MFMAs of the upper ones
    PLP16_ROWS(P, 0, 2);
    if (STEADY) sched_half<2 * RA, 0, NMF>();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this tile's fragment reads have left LDS (WAR on the stage)
    barrier();
    if (has_next) {
      PLP16_READ_A(st[1], 0, 2);
      PLP16_READ_B(1 - P, st[1]);
    }
    PLP16_ROWS(P, 2, 4);
    if (STEADY) sched_half<2 * RA + 4 * RB, 0, NMF>();
    __builtin_amdgcn_sched_barrier(0);
    rotate();
  };

  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using T_ = std::true_type;
  using F_ = std::false_type;
  bool first = true;
  // (m0, n0, slice) of the run's items without a division per item: the column tile advances, then the row block
  int m0, n0, slice, kbeg;
  nk = decode(item_lo, m0, n0, slice, kbeg);
  for (int w = item_lo; w < item_hi; ++w) {
    if (w > item_lo) {
      if (splits > 1) {
        nk = decode(w, m0, n0, slice, kbeg);
      } else {
        n0 += TNW;
        if (n0 >= tiles_n * TNW) { n0 = 0; m0 += TM; }
      }
    }
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[c][a][b] = zero4;
    if (nk > 0) {
      if (first) { barrier(); first = false; }   // the stream's first k-tile has landed (later ones: the previous step's barrier)
      PLP16_READ_A(st[0], 0, 2);
      PLP16_READ_B(0, st[0]);
      int kt = 0;
      for (; kt + 3 < nk; kt += 2) {
        step(kt, P0{}, T_{});
        step(kt + 1, P1{}, T_{});
      }
      for (; kt < nk; kt += 2) {
        step(kt, P0{}, F_{});
        if (kt + 1 < nk) step(kt + 1, P1{}, F_{});
      }
    }
    epi(acc, m0, n0, slice);
  }
#undef PLP16_READ_A
#undef PLP16_READ_B
#undef PLP16_MFS
#undef PLP16_ROWS
}

// ---------------------------------------------------------------------------------------------------------------------
// CHAIN (round 3): the backward pair of one layer -- dX = dz W (NN) and dW = dz^T a (TN, split-K) -- by ONE workgroup per
// CU, one item of each, the operand stream running on from the first into the second.  The dual launch of round 2 ran the
// two problems as two rounds of workgroups: when a CU's dX workgroup retired (after an epilogue that pulls the skip
// gradient, the saved z and the bitmap: 50 MB chip-wide, no MFMA running anywhere) a dW workgroup had to be dispatched,
// fetch its first k-tiles and fill its pipeline.  Here the loader waves issue dW's first k-tiles while the computing waves
// are in dX's epilogue (which stages through LDS of its own, behind the ring), and nothing is re-dispatched.
// Whole tiles, K0 and every K1 slice whole 32-k tiles, as many dX tiles as dW items (the host checks).
template <int MODE, class Epi0, class Epi1>
__device__ __forceinline__ void planes_run16_chain(const PlanesArgs& p0, const PlanesArgs& p1, const int block_id, const int nwork,
                                                   char* __restrict__ lds, Epi0&& epi0, Epi1&& epi1) {
  static_assert(MODE == kF16x3 || MODE == kBf16, "two fp16 planes or one bf16 plane");
  constexpr int NPL = ModeCfg<MODE>::NPL, NACC = ModeCfg<MODE>::NACC;
  using Cf = PlanesCfg<32, NPL, 3>;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int lw = wave & 3;
  const int wm = (wave & 3) >> 1, wn = wave & 1;
  int w = block_id;
  if ((nwork & 7) == 0) w = (block_id & 7) * (nwork >> 3) + (block_id >> 3);   // XCD-aware: blocks b and b+8 share an L2
  // item 0: tile w of the NN problem
  const int tn0 = p0.N / 128;
  const int m00 = (w / tn0) * 128, n00 = (w % tn0) * 128;
  const int nk0 = p0.K / 32;
  // item 1: (slice, tile) w of the TN problem, K-slice-major
  const int tn1 = p1.N / 128;
  const int splits = p1.split_k > 1 ? p1.split_k : 1;
  const int ntiles1 = nwork / splits;
  const int slice = w / ntiles1, t1 = w - slice * ntiles1;
  const int m01 = (t1 / tn1) * 128, n01 = (t1 % tn1) * 128;
  const int per = ((p1.K / 32 + splits - 1) / splits) * 32;
  const int kbeg1 = min(slice * per, p1.K);
  const int nk1 = (min(kbeg1 + per, p1.K) - kbeg1) / 32;
  const int total = nk0 + nk1;

  auto barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto next_stage = [](const int o) { return o + Cf::STAGE == 3 * Cf::STAGE ? 0 : o + Cf::STAGE; };

  if (loader) {
    __amdgpu_buffer_rsrc_t ra0[NPL], rb0[NPL], ra1[NPL], rb1[NPL];
    const __bf16* ta0 = p0.A + (size_t)m00 * p0.lda;                       // NN: A k-contiguous [M][K], B k-strided [K][N]
    const __bf16* tb0 = p0.B + n00;
    const __bf16* ta1 = p1.A + (size_t)kbeg1 * p1.lda + m01;               // TN: both k-strided
    const __bf16* tb1 = p1.B + (size_t)kbeg1 * p1.ldb + n01;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      ra0[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(ta0 + (size_t)pl * p0.a_plane), 0, 0x7fffffff, 0x00020000);
      rb0[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(tb0 + (size_t)pl * p0.b_plane), 0, 0x7fffffff, 0x00020000);
      ra1[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(ta1 + (size_t)pl * p1.a_plane), 0, 0x7fffffff, 0x00020000);
      rb1[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(tb1 + (size_t)pl * p1.b_plane), 0, 0x7fffffff, 0x00020000);
    }
    int oa0[2], ob0[2], oa1[2], ob1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      oa0[j] = (int)glds_lane_off16<false>(lw + 4 * j, lane, p0.lda);
      ob0[j] = (int)glds_lane_off16<true>(lw + 4 * j, lane, p0.ldb);
      oa1[j] = (int)glds_lane_off16<true>(lw + 4 * j, lane, p1.lda);
      ob1[j] = (int)glds_lane_off16<true>(lw + 4 * j, lane, p1.ldb);
    }
    const int ga0 = 64, gb0 = 32 * p0.ldb * 2, ga1 = 32 * p1.lda * 2, gb1 = 32 * p1.ldb * 2;
    // (Two issue routines, each with its own descriptors, in loops of their own: one routine choosing the item at run time
    //  made hipcc select between the descriptor sets per lane -- a waterfall loop around every DMA: 173 us per launch.)
    int sw = 0;
    auto issue0 = [&](const int kt) {
      char* d = lds + sw + lw * 1024;
      const int sa = kt * ga0, sb = kt * gb0;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          PLP_BLDS16(ra0[pl], d + pl * Cf::OPP + j * 4096, oa0[j], sa);
          PLP_BLDS16(rb0[pl], d + (NPL + pl) * Cf::OPP + j * 4096, ob0[j], sb);
        }
      sw = next_stage(sw);
    };
    auto issue1 = [&](const int kt) {
      char* d = lds + sw + lw * 1024;
      const int sa = kt * ga1, sb = kt * gb1;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          PLP_BLDS16(ra1[pl], d + pl * Cf::OPP + j * 4096, oa1[j], sa);
          PLP_BLDS16(rb1[pl], d + (NPL + pl) * Cf::OPP + j * 4096, ob1[j], sb);
        }
      sw = next_stage(sw);
    };
    // (host: nk0 >= 2, nk1 >= 1)  tile g+2 of the stream is issued before the barrier of step g
    issue0(0);
    issue0(1);
    wait_vmcnt<Cf::NDMA>();
    barrier();
    for (int g = 0; g + 2 < nk0; ++g) { issue0(g + 2); wait_vmcnt<Cf::NDMA>(); barrier(); }
    for (int kt = 0; kt < nk1; ++kt) { issue1(kt); wait_vmcnt<Cf::NDMA>(); barrier(); }       // steps nk0 - 2 .. total - 3
    wait_vmcnt<0>(); barrier();
    wait_vmcnt<0>(); barrier();
    return;
  }

  FragAddr16<false> fa_kc;      // A of the NN item
  FragAddr16<true> fa_ks;       // A of the TN item
  FragAddr16<true> fb_ks;       // B of both
  fa_kc.init(wm, lane);
  fa_ks.init(wm, lane);
  fb_ks.init(wn, lane);
  s16x8 fa[4][NPL];
  s16x8 fb[2][4][NPL];
  f32x4v acc[NACC][4][4];
  int st[2] = {0, Cf::STAGE};
  auto rotate = [&]() { st[0] = st[1]; st[1] = next_stage(st[1]); };

#define PLC_READ_A(AKS, stage_off, t0, t1)                                                   \
  do {                                                                                       \
    const char* q_ = lds + (stage_off);                                                      \
    _Pragma("unroll") for (int t2 = (t0); t2 < (t1); ++t2)                                   \
    _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                                     \
      if constexpr (AKS) fa[t2][pl] = read_frag16<true>(q_ + pl * Cf::OPP, fa_ks, t2);       \
      else fa[t2][pl] = read_frag16<false>(q_ + pl * Cf::OPP, fa_kc, t2);                    \
    }                                                                                        \
  } while (0)
#define PLC_READ_B(set, stage_off)                                                           \
  do {                                                                                       \
    const char* q_ = lds + (stage_off) + NPL * Cf::OPP;                                      \
    _Pragma("unroll") for (int t2 = 0; t2 < 4; ++t2)                                         \
    _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                                       \
      fb[set][t2][pl] = read_frag16<true>(q_ + pl * Cf::OPP, fb_ks, t2);                     \
  } while (0)
#define PLC_MFS(set, rt, ct)                                                                           \
  do {                                                                                                 \
    if constexpr (MODE == kF16x3) {                                                                    \
      acc[1][rt][ct] = mfma16x16<MODE>(fa[rt][0], fb[set][ct][1], acc[1][rt][ct]);                      \
      acc[1][rt][ct] = mfma16x16<MODE>(fa[rt][1], fb[set][ct][0], acc[1][rt][ct]);                      \
    }                                                                                                  \
    acc[0][rt][ct] = mfma16x16<MODE>(fa[rt][0], fb[set][ct][0], acc[0][rt][ct]);                        \
  } while (0)
#define PLC_ROWS(set, r0, r1)                                                                \
  do {                                                                                       \
    _Pragma("unroll") for (int rt = (r0); rt < (r1); ++rt)                                   \
    _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) PLC_MFS(set, rt, ct);                   \
  } while (0)

  constexpr int NMF = 8 * ModeCfg<MODE>::NPROD;
  int nk = 0;
  auto step = [&](const int kt, auto par, auto steady, auto aks) {
    constexpr int P = decltype(par)::value;
    constexpr bool STEADY = decltype(steady)::value;
    constexpr bool AKS = decltype(aks)::value;
    constexpr int RA = NPL * (AKS ? 2 : 1), RB = NPL * 2;
    const bool has_next = STEADY || kt + 1 < nk;
    PLC_READ_A(AKS, st[0], 2, 4);
    PLC_ROWS(P, 0, 2);
    if (STEADY) sched_half<2 * RA, 0, NMF>();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
    if (has_next) {
      PLC_READ_A(AKS, st[1], 0, 2);
      PLC_READ_B(1 - P, st[1]);
    }
    PLC_ROWS(P, 2, 4);
    if (STEADY) sched_half<2 * RA + 4 * RB, 0, NMF>();
    __builtin_amdgcn_sched_barrier(0);
    rotate();
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using T_ = std::true_type;
  using F_ = std::false_type;
  const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
  auto zero_acc = [&]() {
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[c][a][b] = zero4;
  };
  auto run_item = [&](auto aks) {
    constexpr bool AKS = decltype(aks)::value;
    PLC_READ_A(AKS, st[0], 0, 2);
    PLC_READ_B(0, st[0]);
    int kt = 0;
    for (; kt + 3 < nk; kt += 2) {
      step(kt, P0{}, T_{}, aks);
      step(kt + 1, P1{}, T_{}, aks);
    }
    for (; kt < nk; kt += 2) {
      step(kt, P0{}, F_{}, aks);
      if (kt + 1 < nk) step(kt + 1, P1{}, F_{}, aks);
    }
  };
  if (total <= 0) return;
  barrier();                                   // the stream's first k-tile has landed
  zero_acc();
  nk = nk0;
  if (nk > 0) run_item(F_{});
  epi0(acc, m00, n00, 0);
  zero_acc();
  nk = nk1;
  if (nk > 0) run_item(T_{});
  epi1(acc, m01, n01, slice);
#undef PLC_READ_A
#undef PLC_READ_B
#undef PLC_MFS
#undef PLC_ROWS
}

// ---------------------------------------------------------------------------------------------------------------------
// WIDE (round 3): the same loop with k-tiles staged in PAIRS, so that a k-contiguous operand is fetched in whole 128-byte
// lines.  With one 32-k tile per stage a k-contiguous row contributes 64 bytes per tile: a DMA instruction covers 16 rows x
// 64 B, i.e. 16 half lines, and every line crosses the L1 / texture-address path twice (once per tile).  The k-strided
// operands never had that problem (256-byte rows) -- and they were the faster loops: DMA alone 16 us for the TN problem of
// the lifter against 23.6 us for NT.  Here a stage holds TWO k-tiles:
//   k-contiguous image [128 rows][128 B] per plane (16 KB): one DMA instruction = 8 rows x 128 B; the 16-byte chunk a lane
//     FETCHES is XORed with (row >> 1) & 7, so that the 16 rows x 1 chunk column of a ds_read_b128 lane group fall on the 16
//     slots of the 256-byte bank row; k-tile h of the pair = chunks 4h .. 4h+3;
//   k-strided image: the pair's two [32 k][256 B] images back to back (unchanged).
// Two pair-stages (2 x 64 KB for two fp16 planes).  The pair p+2 is issued right after the mid-step barrier of the odd
// step 2p+1 (when the last fragment of pair p has left LDS) and has landed by the barrier of step 2p+3: two steps in
// flight, as the three-stage ring gives a single tile.  Every K slice must be a whole number of PAIRS (K % 64 == 0 per
// slice); CONV = 1 then needs Cin % 64 == 0 (a pair inside one filter tap).
__device__ __forceinline__ int kc16w_swz(int row) { return (row >> 1) & 7; }

template <bool KS>
__device__ __forceinline__ uint32_t glds_lane_off16w(int rb, int lane, int ld, int left = 0x7fffffff) {
  if (!KS) {   // 8 rows x 128 B per instruction
    const int row = rb * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ kc16w_swz(row);
    return (uint32_t)(min(row, left - 1) * ld * 2 + ch * 16);
  }
  const int krow = rb * 4 + (lane >> 4);               // 0 .. 63: both tiles of the pair
  const int ch = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | (rb & 3));
  return (uint32_t)(krow * ld * 2 + min(ch, (left >> 3) - 1) * 16);
}

template <bool KS>
struct FragAddr16W {
  uint32_t b[KS ? 8 : 2];
  __device__ __forceinline__ void init(int wq, int lane) {
    const int r = lane & 15, q = lane >> 4;
    if (!KS) {
      const int row = wq * 64 + r;                       // (+ 16 t rows: (row >> 1) & 7 does not depend on t)
#pragma unroll
      for (int h = 0; h < 2; ++h) b[h] = (uint32_t)(row * 128 + (((q + 4 * h) ^ kc16w_swz(row)) << 4));
    } else {
      const int qq = r >> 2, pp = r & 3;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int row = 8 * q + 4 * u + qq;
          const int ch = wq * 8 + t * 2 + (pp >> 1);
          const int x = ((row & 3) << 2) | ((row >> 2) & 3);
          b[t * 2 + u] = (uint32_t)(256 * row + 16 * (ch ^ x) + 8 * (pp & 1));
        }
    }
  }
};

// fragment of 16-row tile t of the wave's block, k-tile h of the pair, from one plane's pair image `op`
template <bool KS, int H>
__device__ __forceinline__ s16x8 read_frag16w(const char* op, const FragAddr16W<KS>& fa, int t) {
  if (!KS) return *reinterpret_cast<const s16x8*>(op + fa.b[H] + t * 2048);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(op + H * 8192 + fa.b[t * 2 + 0]));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(op + H * 8192 + fa.b[t * 2 + 1]));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int NPL>
struct WideCfg {
  static constexpr int OPP = 128 * 64 * 2;      // one plane of one operand, a PAIR of k-tiles: 16 KB
  static constexpr int STAGE = 2 * NPL * OPP;   // A planes then B planes
  static constexpr int LDS = 2 * STAGE;
  static constexpr int NDMA = 2 * NPL * 4;      // per loader wave and pair
};

template <bool A_KS, bool B_KS, int MODE, bool EDGE, int CONV, bool PERSIST, class Epi>
__device__ __forceinline__ void planes_run16w(const PlanesArgs& p, const int block_id, const int nblocks, const int nwork,
                                              char* __restrict__ lds, Epi&& epi) {
  static_assert(MODE == kF16x3 || MODE == kBf16, "two fp16 planes or one bf16 plane");
  static_assert(CONV == 0 || (CONV == 1 && !A_KS && !B_KS), "the gathered weight gradient keeps the one-tile stages");
  constexpr int NPL = ModeCfg<MODE>::NPL, NACC = ModeCfg<MODE>::NACC;
  using Cf = WideCfg<NPL>;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int lw = wave & 3;
  const int wm = (wave & 3) >> 1, wn = wave & 1;

  const int tiles_n = EDGE ? (p.N + 127) / 128 : p.N / 128;
  const int splits = p.split_k > 1 ? p.split_k : 1;
  const int ntiles = nwork / splits;
  int item_lo, item_hi;
  if (PERSIST) {
    const int per = (nwork + nblocks - 1) / nblocks;
    item_lo = min(block_id * per, nwork);
    item_hi = min(item_lo + per, nwork);
  } else {
    item_lo = block_id;
    if ((nwork & 7) == 0) item_lo = (block_id & 7) * (nwork >> 3) + (block_id >> 3);
    item_hi = item_lo + 1;
  }
  if (item_lo >= item_hi) return;
  const int per_slice = ((p.K / 64 + splits - 1) / splits) * 64;      // (host: every slice a whole number of pairs)
  auto decode = [&](const int w, int& m0, int& n0, int& slice, int& kbeg) {
    slice = w / ntiles;
    const int t = w - slice * ntiles;
    m0 = (t / tiles_n) * 128;
    n0 = (t % tiles_n) * 128;
    kbeg = 0;
    int kend = p.K;
    if (splits > 1) {
      kbeg = min(slice * per_slice, p.K);
      kend = min(kbeg + per_slice, p.K);
    }
    return (kend - kbeg) / 64;                         // PAIRS of k-tiles
  };
  auto barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  if (loader) {
    int npairs = 0;
    if (splits == 1) npairs = (item_hi - item_lo) * (p.K / 64);
    else for (int w = item_lo; w < item_hi; ++w) { int a_, b_, c_, d_; npairs += max(decode(w, a_, b_, c_, d_), 0); }
    int it = item_lo, pp = 0, np = 0, m0 = 0, n0 = 0, slice = 0, kbeg = 0;
    __amdgpu_buffer_rsrc_t ra[NPL], rb[NPL], rx[NPL];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
      rx[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.A + (size_t)pl * p.a_plane), 0, 0x7fffffe0, 0x00020000);
    int oa[4], ob[4];
    int cih0[4] = {0, 0, 0, 0}, ciw0[4] = {0, 0, 0, 0}, cbase[4] = {0, 0, 0, 0};
    int cc0 = 0, ckh = 0, ckw = 0;
    const int ga_step = A_KS ? 64 * p.lda * 2 : 128;
    const int gb_step = B_KS ? 64 * p.ldb * 2 : 128;
    auto setup = [&]() {
      np = decode(it, m0, n0, slice, kbeg);
      const __bf16* ta = A_KS ? p.A + (size_t)kbeg * p.lda + m0 : p.A + (size_t)m0 * p.lda + kbeg;
      const __bf16* tb = B_KS ? p.B + (size_t)kbeg * p.ldb + n0 : p.B + (size_t)n0 * p.ldb + kbeg;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        ra[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(ta + (size_t)pl * p.a_plane), 0, 0x7fffffff, 0x00020000);
        rb[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(tb + (size_t)pl * p.b_plane), 0, 0x7fffffff, 0x00020000);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        oa[j] = (int)glds_lane_off16w<A_KS>(lw + 4 * j, lane, p.lda, EDGE ? p.M - m0 : 0x7fffffff);
        ob[j] = (int)glds_lane_off16w<B_KS>(lw + 4 * j, lane, p.ldb, EDGE ? p.N - n0 : 0x7fffffff);
      }
      if (CONV == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = (lw + 4 * j) * 8 + (lane >> 3);
          const int m = min(m0 + row, p.M - 1);
          const int ow = m % p.cv_wo, t2 = m / p.cv_wo;
          const int oh = t2 % p.cv_ho, b = t2 / p.cv_ho;
          cih0[j] = oh * p.cv_stride - p.cv_pad_h;
          ciw0[j] = ow * p.cv_stride_w - p.cv_pad_w;
          cbase[j] = ((b * p.cv_h + cih0[j]) * p.cv_w + ciw0[j]) * p.cv_cin * 2 + (((lane & 7) ^ kc16w_swz(row)) << 4);
        }
        const int ctap = kbeg / p.cv_cin;
        cc0 = kbeg - ctap * p.cv_cin;
        ckh = ctap / p.cv_kw; ckw = ctap - ckh * p.cv_kw;
      }
    };
    auto issue = [&](const int stage_off) {
      if (p.abl & 2) return;
      const int sa = pp * ga_step, sb = pp * gb_step;
      char* d = lds + stage_off + lw * 1024;
      int va[4] = {oa[0], oa[1], oa[2], oa[3]};
      if (CONV == 1) {                                   // pairs are issued in K order: the tap advances incrementally
        const int toff = ((ckh * p.cv_w + ckw) * p.cv_cin + cc0) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = (unsigned)(cih0[j] + ckh) < (unsigned)p.cv_h && (unsigned)(ciw0[j] + ckw) < (unsigned)p.cv_w;
          va[j] = ok ? cbase[j] + toff : kDmaOutOfRange;
        }
        cc0 += 64;
        if (cc0 >= p.cv_cin) { cc0 = 0; if (++ckw == p.cv_kw) { ckw = 0; ++ckh; } }
      }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (CONV == 1) PLP_BLDS16(rx[pl], d + pl * Cf::OPP + j * 4096, va[j], 0);
          else PLP_BLDS16(ra[pl], d + pl * Cf::OPP + j * 4096, va[j], sa);
          PLP_BLDS16(rb[pl], d + (NPL + pl) * Cf::OPP + j * 4096, ob[j], sb);
        }
    };
    auto open_item = [&]() {
      for (; it < item_hi; ++it) {
        setup();
        if (np > 0) { pp = 0; return true; }
      }
      return false;
    };
    int sw = 0;
    auto issue_next = [&]() {                         // called exactly `npairs` times
      if (pp >= np) { ++it; if (!open_item()) return; }
      issue(sw);
      sw = Cf::STAGE - sw;
      ++pp;
    };
    if (npairs <= 0 || !open_item()) return;
    issue_next();
    if (npairs > 1) { issue_next(); wait_vmcnt<Cf::NDMA>(); } else { wait_vmcnt<0>(); }
    barrier();                                        // pair 0 has landed
    int issued = npairs > 1 ? 2 : 1;
    for (int g = 0; g < 2 * npairs; ++g) {
      if (g & 1) {
        wait_vmcnt<0>();                              // pair (g + 1) / 2 -- the only one in flight -- has landed
        barrier();                                    // ... and the last fragment of pair (g - 1) / 2 has left LDS:
        if (issued < npairs) { issue_next(); ++issued; }      // its stage takes pair (g + 3) / 2
      } else {
        barrier();
      }
    }
    return;
  }

  FragAddr16W<A_KS> fra;
  FragAddr16W<B_KS> frb;
  fra.init(wm, lane);
  frb.init(wn, lane);
  s16x8 fa[4][NPL];
  s16x8 fb[2][4][NPL];
  f32x4v acc[NACC][4][4];

#define PLW_READ_A(stage_off, H, t0, t1)                                                     \
  do {                                                                                       \
    const char* q_ = lds + (stage_off);                                                      \
    _Pragma("unroll") for (int t2 = (t0); t2 < (t1); ++t2)                                   \
    _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                                       \
      fa[t2][pl] = read_frag16w<A_KS, H>(q_ + pl * Cf::OPP, fra, t2);                        \
  } while (0)
#define PLW_READ_B(set, stage_off, H)                                                        \
  do {                                                                                       \
    const char* q_ = lds + (stage_off) + NPL * Cf::OPP;                                      \
    _Pragma("unroll") for (int t2 = 0; t2 < 4; ++t2)                                         \
    _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                                       \
      fb[set][t2][pl] = read_frag16w<B_KS, H>(q_ + pl * Cf::OPP, frb, t2);                   \
  } while (0)
#define PLW_MFS(set, rt, ct)                                                                           \
  do {                                                                                                 \
    if constexpr (MODE == kF16x3) {                                                                    \
      acc[1][rt][ct] = mfma16x16<MODE>(fa[rt][0], fb[set][ct][1], acc[1][rt][ct]);                      \
      acc[1][rt][ct] = mfma16x16<MODE>(fa[rt][1], fb[set][ct][0], acc[1][rt][ct]);                      \
    }                                                                                                  \
    acc[0][rt][ct] = mfma16x16<MODE>(fa[rt][0], fb[set][ct][0], acc[0][rt][ct]);                        \
  } while (0)
#define PLW_ROWS(set, r0, r1)                                                                \
  do {                                                                                       \
    if (p.abl & 4) break;                                                                    \
    _Pragma("unroll") for (int rt = (r0); rt < (r1); ++rt)                                   \
    _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) PLW_MFS(set, rt, ct);                   \
  } while (0)

  constexpr int RA = NPL * (A_KS ? 2 : 1), RB = NPL * (B_KS ? 2 : 1);
  constexpr int NMF = 8 * ModeCfg<MODE>::NPROD;
  int sc = 0, sn = Cf::STAGE;        // pair-stage of the k-tile being read, the other one
  int nk = 0;
  const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
  // step kt of parity P = kt & 1 = its half of the pair: the NEXT k-tile is half 1 of the same pair-stage (P = 0) or half 0 of
  // the other one (P = 1)
  auto step = [&](const int kt, auto par, auto steady) {
    constexpr int P = decltype(par)::value;
    constexpr bool STEADY = decltype(steady)::value;
    const bool has_next = STEADY || kt + 1 < nk;
    PLW_READ_A(sc, P, 2, 4);
    PLW_ROWS(P, 0, 2);
    if (STEADY) sched_half<2 * RA, 0, NMF>();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
    if (has_next) {
      if constexpr (P == 0) { PLW_READ_A(sc, 1, 0, 2); PLW_READ_B(1, sc, 1); }
      else { PLW_READ_A(sn, 0, 0, 2); PLW_READ_B(0, sn, 0); }
    }
    PLW_ROWS(P, 2, 4);
    if (STEADY) sched_half<2 * RA + 4 * RB, 0, NMF>();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (P == 1) { const int o = sc; sc = sn; sn = o; }
  };

  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using T_ = std::true_type;
  using F_ = std::false_type;
  bool first = true;
  int m0, n0, slice, kbeg;
  nk = 2 * decode(item_lo, m0, n0, slice, kbeg);
  for (int w = item_lo; w < item_hi; ++w) {
    if (w > item_lo) {
      if (splits > 1) {
        nk = 2 * decode(w, m0, n0, slice, kbeg);
      } else {
        n0 += 128;
        if (n0 >= tiles_n * 128) { n0 = 0; m0 += 128; }
      }
    }
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[c][a][b] = zero4;
    if (nk > 0) {
      if (first) { barrier(); first = false; }
      PLW_READ_A(sc, 0, 0, 2);                 // (an item is a whole number of pairs: it starts on half 0)
      PLW_READ_B(0, sc, 0);
      int kt = 0;
      for (; kt + 3 < nk; kt += 2) {
        step(kt, P0{}, T_{});
        step(kt + 1, P1{}, T_{});
      }
      for (; kt < nk; kt += 2) {
        step(kt, P0{}, F_{});
        if (kt + 1 < nk) step(kt + 1, P1{}, F_{});
      }
    }
    epi(acc, m0, n0, slice);
  }
#undef PLW_READ_A
#undef PLW_READ_B
#undef PLW_MFS
#undef PLW_ROWS
}

}  // namespace plp
