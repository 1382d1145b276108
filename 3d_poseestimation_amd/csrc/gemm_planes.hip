// GEMMs on pre-split 16-bit operand planes (gemm_planes.h): the 1024-wide Linears of the lifter in the
// PL_F16X3 (fp32-grade, three fp16 MFMAs per product term) and PL_BF16 (bf16 storage) modes.
//
// Replaces the ATen addmm calls behind nn.Linear forward / backward (reference phase1_lifting/baselineModel.py:33,39
// and their autograd), like gemm_f32.hip; same 128x128 tile per workgroup, same register epilogue (gemm_epilogue.h:
// bias, training-mode BatchNorm partial statistics, eval-mode BN fold + ReLU + skip, residual-gradient addend, split-K
// slabs).  512 threads: wavefronts 0-3 compute, wavefronts 4-7 drive the LDS-DMA.
#include <stdlib.h>

#include "gemm_epilogue.h"
#include "gemm_planes16.h"
#include "pl_internal.h"
#include "plane_store.h"

namespace pl {
namespace {

struct PlanesKern {
  plp::PlanesArgs p;
  GemmArgs e;
  float out_scale;
  const float* dyn_inv;
  int vec_addend;          // POSELIFT_ADDEND_SCALAR=1 (same-box A/B): the addend through gemm_epilogue's dword loads
  int nt_store;            // C leaves with nontemporal stores (an output far larger than the caches, read next from HBM anyway)
};

// x_l + x_(l ^ 16) and x_l + x_(l ^ 32) in every lane, by ONE vector instruction each (gfx950: v_permlane16_swap /
// v_permlane32_swap exchange 16- / 32-lane rows of two registers) instead of a ds_bpermute round trip through the LDS
// crossbar.  IEEE addition commutes, so the sums are bit for bit those of `x += __shfl_xor(x, 16 | 32)`.
__device__ __forceinline__ float xadd16(const float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xadd32(const float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float4 xadd_rows(float4 s) {       // the four lanes l, l^16, l^32, l^48: fixed order (16, then 32)
  s.x = xadd32(xadd16(s.x)); s.y = xadd32(xadd16(s.y)); s.z = xadd32(xadd16(s.z)); s.w = xadd32(xadd16(s.w));
  return s;
}

// ---- the epilogue on the wave's 64x64 block staged in its own 17 KB of LDS (row stride 68 floats: conflict-free both
// ways), everything as 16-byte accesses on 256-byte row segments.  Lane (lr = lane >> 4, lc = 4 (lane & 15)) owns columns
// col0 .. col0+3 of rows lr, lr+4, ..., lr+60.  Same operation order per element as gemm_epilogue (bias, skip-gradient
// addend, eval-BN fold, ReLU, residual, ReLU); statistics and the BatchNorm-backward sums per 64-row block in a fixed order.
template <bool EDGE>
__device__ __forceinline__ void staged_epilogue(const PlanesKern& k, float* __restrict__ C, float4 (&v)[16],
                                                const int m0, const int n0, const int wm, const int wn, const int lane) {
  const GemmArgs& e = k.e;
  const int lr = lane >> 4, lc = (lane & 15) * 4;
  const int row0 = m0 + wm * 64 + lr, col0 = n0 + wn * 64 + lc;
  const size_t o0 = (size_t)row0 * e.ldc + col0;
  const bool plain = e.split_k > 1;              // split-K slab: the bare partial product
  // EDGE: rows >= M / columns >= N of an overhanging tile hold garbage: never loaded for, never stored, never counted
  const int nrows = EDGE ? e.M - (m0 + wm * 64) : 64;          // valid rows of this wave's block (may be <= 0)
  const bool cok = !EDGE || col0 < e.N;
  auto ok = [&](int it) { return !EDGE || (cok && lr + 4 * it < nrows); };
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // (Tried: requesting bnr_z's sixteen float4 here, ahead of the addend, so that both arrive in one round trip instead of two in
  //  a row.  Same-box A/B over five interleaved runs: dual launch 63.1 us against 62.4 us as it is -- the epilogue's burst is
  //  bandwidth, not latency; more of it in flight at once only lengthens the queue.  Not kept.)
  if (!plain) {
    if (e.bias) {
      const float4 b = cok ? *reinterpret_cast<const float4*>(e.bias + col0) : zero4;
#pragma unroll
      for (int it = 0; it < 16; ++it) { v[it].x += b.x; v[it].y += b.y; v[it].z += b.z; v[it].w += b.w; }
    }
    if (e.addend) {
      float4 q[16];
#pragma unroll
      for (int it = 0; it < 16; ++it) q[it] = ok(it) ? *reinterpret_cast<const float4*>(e.addend + o0 + (size_t)it * 4 * e.ldc) : zero4;
#pragma unroll
      for (int it = 0; it < 16; ++it) { v[it].x += q[it].x; v[it].y += q[it].y; v[it].z += q[it].z; v[it].w += q[it].w; }
    }
    if (e.col_scale) {
      const float4 sc = cok ? *reinterpret_cast<const float4*>(e.col_scale + col0) : zero4;
      const float4 sh = cok ? *reinterpret_cast<const float4*>(e.col_shift + col0) : zero4;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        v[it].x = fmaf(v[it].x, sc.x, sh.x); v[it].y = fmaf(v[it].y, sc.y, sh.y);
        v[it].z = fmaf(v[it].z, sc.z, sh.z); v[it].w = fmaf(v[it].w, sc.w, sh.w);
      }
    }
    if (e.relu == 1) {
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        v[it].x = fmaxf(v[it].x, 0.f); v[it].y = fmaxf(v[it].y, 0.f); v[it].z = fmaxf(v[it].z, 0.f); v[it].w = fmaxf(v[it].w, 0.f);
      }
    }
    if (e.resid) {
      float4 q[16];
#pragma unroll
      for (int it = 0; it < 16; ++it) q[it] = ok(it) ? *reinterpret_cast<const float4*>(e.resid + o0 + (size_t)it * 4 * e.ldc) : zero4;
#pragma unroll
      for (int it = 0; it < 16; ++it) { v[it].x += q[it].x; v[it].y += q[it].y; v[it].z += q[it].z; v[it].w += q[it].w; }
    }
    if (e.relu == 2) {
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        v[it].x = fmaxf(v[it].x, 0.f); v[it].y = fmaxf(v[it].y, 0.f); v[it].z = fmaxf(v[it].z, 0.f); v[it].w = fmaxf(v[it].w, 0.f);
      }
    }
  }
  const PlaneDst cpd = {e.cpl_h, e.cpl_l, e.cpl_scale, plain ? 0 : e.cpl_kind, k.nt_store};      // the result as the next GEMM's operand planes
  if (EDGE && e.scat_on) {                       // one output parity of a transposed convolution (GemmArgs::scat_*)
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      if (!ok(it)) continue;
      const int m = row0 + 4 * it;
      const int c = m % e.conv_wo, t = m / e.conv_wo;
      const int a = t % e.conv_ho, b = t / e.conv_ho;
      const size_t orow = ((size_t)b * 2 * e.conv_ho + 2 * a + e.scat_ph) * 2 * e.conv_wo + 2 * c + e.scat_pw;
      if (C) *reinterpret_cast<float4*>(C + orow * e.ldc + col0) = v[it];
      if (cpd.kind) store_planes4(cpd, orow * e.ldc + col0, v[it]);
    }
    return;
  }
  if (k.nt_store && C) {
    typedef float f4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int it = 0; it < 16; ++it)
      if (ok(it)) {
        const f4v q = {v[it].x, v[it].y, v[it].z, v[it].w};
        __builtin_nontemporal_store(q, reinterpret_cast<f4v*>(C + o0 + (size_t)it * 4 * e.ldc));
        if (cpd.kind) store_planes4(cpd, o0 + (size_t)it * 4 * e.ldc, v[it]);
      }
  } else {
#pragma unroll
  for (int it = 0; it < 16; ++it)
    if (ok(it)) {
      if (C) *reinterpret_cast<float4*>(C + o0 + (size_t)it * 4 * e.ldc) = v[it];
      if (cpd.kind) store_planes4(cpd, o0 + (size_t)it * 4 * e.ldc, v[it]);
    }
  }
  if (plain) return;
  if (e.stat_sum) {
    // training-mode BatchNorm partial statistics of this 64-row block (sum and M2 about the block's own mean); lanes l,
    // l^16, l^32, l^48 hold the same four columns: fixed-order butterfly
    if (EDGE) {
#pragma unroll
      for (int it = 0; it < 16; ++it)
        if (!ok(it)) v[it] = zero4;
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < 16; ++it) { s.x += v[it].x; s.y += v[it].y; s.z += v[it].z; s.w += v[it].w; }
    s = xadd_rows(s);
    const int cnt = EDGE ? max(0, min(64, nrows)) : 64;
    const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
    const float4 mean = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
    float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      if (EDGE && !ok(it)) continue;
      const float dx = v[it].x - mean.x, dy = v[it].y - mean.y, dz = v[it].z - mean.z, dw = v[it].w - mean.w;
      m2.x = fmaf(dx, dx, m2.x); m2.y = fmaf(dy, dy, m2.y); m2.z = fmaf(dz, dz, m2.z); m2.w = fmaf(dw, dw, m2.w);
    }
    m2 = xadd_rows(m2);
    // (the consumers walk 2 ceil(M / 128) groups, empty ones included -- gemm_stat_groups; a wave block of a 256-row tile
    //  below that has no group)
    if (lr == 0 && cok && (!EDGE || (m0 >> 6) + wm < 2 * ((e.M + 127) >> 7))) {
      const size_t o = (size_t)((m0 >> 6) + wm) * e.N + col0;
      *reinterpret_cast<float4*>(e.stat_sum + o) = s;
      *reinterpret_cast<float4*>(e.stat_m2 + o) = m2;
    }
  }
  if (e.bnr_z) {
    // BatchNorm-backward pass 1 of the layer below on the block just produced (GemmArgs::bnr_*): the separate pass read
    // g (16.8 MB) back from HBM, here it is in registers.  One partial row per 64-row block, in a fixed order.
    const int N = e.N;
    const int wpr = ((N + 255) >> 8) * 4;
    const int strip = col0 >> 8, bit = (col0 & 255) >> 2;
    const float4 mu = *reinterpret_cast<const float4*>(e.bnr_mean + col0);
    const float4 rs = *reinterpret_cast<const float4*>(e.bnr_rstd + col0);
    const float ks = e.bnr_kscale;
    float4 zz[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) zz[it] = *reinterpret_cast<const float4*>(e.bnr_z + o0 + (size_t)it * 4 * e.ldc);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    float mxd = 0.f, mxz = 0.f;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const uint64_t* bw = e.bnr_bits + (size_t)(row0 + 4 * it) * wpr + strip * 4;
      const ulonglong2 w01 = *reinterpret_cast<const ulonglong2*>(bw), w23 = *reinterpret_cast<const ulonglong2*>(bw + 2);
      const float d0 = ((w01.x >> bit) & 1ull) ? v[it].x * ks : 0.f;
      const float d1 = ((w01.y >> bit) & 1ull) ? v[it].y * ks : 0.f;
      const float d2 = ((w23.x >> bit) & 1ull) ? v[it].z * ks : 0.f;
      const float d3 = ((w23.y >> bit) & 1ull) ? v[it].w * ks : 0.f;
      const float z0 = (zz[it].x - mu.x) * rs.x, z1 = (zz[it].y - mu.y) * rs.y;
      const float z2 = (zz[it].z - mu.z) * rs.z, z3 = (zz[it].w - mu.w) * rs.w;
      s1.x += d0; s1.y += d1; s1.z += d2; s1.w += d3;
      s2.x = fmaf(d0, z0, s2.x); s2.y = fmaf(d1, z1, s2.y); s2.z = fmaf(d2, z2, s2.z); s2.w = fmaf(d3, z3, s2.w);
      mxd = fmaxf(fmaxf(mxd, fmaxf(fabsf(d0), fabsf(d1))), fmaxf(fabsf(d2), fabsf(d3)));
      mxz = fmaxf(fmaxf(mxz, fmaxf(fabsf(z0), fabsf(z1))), fmaxf(fabsf(z2), fabsf(z3)));
    }
    s1 = xadd_rows(s1);
    s2 = xadd_rows(s2);
    const int rg = (m0 >> 6) + wm;                   // 64-row block of the whole matrix
    if (lr == 0) {
      *reinterpret_cast<float4*>(e.bnr_part_dy + (size_t)rg * N + col0) = s1;
      *reinterpret_cast<float4*>(e.bnr_part_dyz + (size_t)rg * N + col0) = s2;
    }
    if (e.bnr_amax) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { mxd = fmaxf(mxd, __shfl_xor(mxd, o)); mxz = fmaxf(mxz, __shfl_xor(mxz, o)); }
      if (lane == 0) {
        float* q = e.bnr_amax + ((size_t)rg * (N >> 6) + (n0 >> 6) + wn) * 2;
        q[0] = mxd; q[1] = mxz;
      }
    }
  }
}

// S16 = true (default): the 16x16x32 main loop (gemm_planes16.h) -- the chip holds a higher matrix clock on that shape
// under this loop's load -- and the staged epilogue for everything.  S16 = false (POSELIFT_MFMA32=1, same-box A/B): the
// 32x32x16 loop; its accumulator layout is gemm_epilogue's, so only the addend / BatchNorm-backward cases are staged.
// EDGE (16x16x32 loop only): M / N need not be multiples of 128 -- the conv path's 64-wide layers and ragged pixel counts
// CONV (16x16x32 loop only): 1 = A gathered as a convolution input (NT), 2 = B gathered for the weight gradient (TN)
// PERSIST (16x16x32 loop only): the workgroup walks a run of work items, operands streaming across the item boundaries
// (gemm_planes16.h); the epilogue then stages through LDS of its own behind the three-stage ring.
template <int MODE, int NST = 3> constexpr int ring_bytes() { return plp::PlanesCfg<32, plp::ModeCfg<MODE>::NPL, NST>::LDS; }
// the 256 x 64 tile (gemm_planes16.h T64): A image 256 rows, B image 64 rows of 64 B per plane and stage, three stages
template <int MODE> constexpr int t64_ring_bytes() { return 3 * plp::ModeCfg<MODE>::NPL * (256 + 64) * 64; }
// the staged epilogue takes the wave's 64x64 block through LDS in 64 / ER passes of ER rows x 68 floats per wave
template <int ER> constexpr int stage_bytes() { return 4 * ER * 68 * 4; }
// ring depth of the persistent form.  (Measured with 4 stages + 16-row staging passes for f16x3, 6 for bf16: the K = 64
// layer 372 us against 379 us with 3 -- the ring is not what those launches wait for)
template <int MODE> constexpr int persist_nst() { return 3; }
constexpr int kPersistRows = 32;
// LDS: the operand ring, and never less than the 4 x 17 KB the 32x32x16 loop's epilogue stages a whole block in
template <int MODE> constexpr int wide_ring_bytes() { return plp::WideCfg<plp::ModeCfg<MODE>::NPL>::LDS; }
constexpr int kWidePersistRows = 16;              // 2 x 64 KB of pair-stages leave 32 KB: the staged epilogue in 16-row passes
template <int MODE, bool PERSIST = false, bool WIDE = false, bool T64 = false>
constexpr int lds_bytes() {
  if (T64) return PERSIST ? t64_ring_bytes<MODE>() + stage_bytes<kPersistRows>()
                          : (t64_ring_bytes<MODE>() > stage_bytes<32>() ? t64_ring_bytes<MODE>() : stage_bytes<32>());
  if (WIDE) return PERSIST ? wide_ring_bytes<MODE>() + stage_bytes<kWidePersistRows>()
                           : (wide_ring_bytes<MODE>() > stage_bytes<32>() ? wide_ring_bytes<MODE>() : stage_bytes<32>());
  if (PERSIST) return ring_bytes<MODE, persist_nst<MODE>()>() + stage_bytes<kPersistRows>();
  return ring_bytes<MODE>() > 4 * 64 * 68 * 4 ? ring_bytes<MODE>() : 4 * 64 * 68 * 4;
}
static_assert(lds_bytes<plp::kF16x3, true>() <= 160 * 1024 && lds_bytes<plp::kBf16, true>() <= 160 * 1024, "LDS per CU");
static_assert(lds_bytes<plp::kF16x3, true, true>() <= 160 * 1024 && lds_bytes<plp::kF16x3, false, true>() <= 160 * 1024, "LDS per CU");
static_assert(lds_bytes<plp::kF16x3, true, false, true>() <= 160 * 1024, "LDS per CU");

template <bool A_KS, bool B_KS, int MODE, bool S16, bool EDGE = false, int CONV = 0, bool PERSIST = false, bool WIDE = false,
          bool T64 = false>
__device__ __forceinline__ void planes_body(const PlanesKern& k, const int block_id, const int nblocks, const int nwork, char* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = T64 ? (wave & 3) : (wave & 3) >> 1, wn = T64 ? 0 : (wave & 1);
  // main + low / 2048, then back from the operands' power-of-two scales (exact unless the result under/overflows)
  const float os = MODE == plp::kF16x3 ? (k.dyn_inv ? k.out_scale * k.dyn_inv[0] : k.out_scale) : 1.f;
  if constexpr (S16) {
    // the wave's 64x64 block goes through ER rows x 68 floats of LDS in 64 / ER passes: the lane that owned accumulator
    // elements leaves with float4 rows (v[it] = row 4 it + (lane >> 4), columns 4 (lane & 15) ..)
    constexpr int NST = PERSIST ? persist_nst<MODE>() : 3;
    constexpr int ER = PERSIST ? (WIDE ? kWidePersistRows : kPersistRows) : 32;
    constexpr int RING = T64 ? t64_ring_bytes<MODE>() : (WIDE ? wide_ring_bytes<MODE>() : ring_bytes<MODE, NST>());
    float* ldsw = reinterpret_cast<float*>(lds + (PERSIST ? RING : 0)) + (wave & 3) * ER * 68;
    const int q = lane >> 4, c = lane & 15, lc = c * 4;
    auto epi = [&](plp::f32x4v (&acc)[plp::ModeCfg<MODE>::NACC][4][4], const int m0, const int n0, const int slice) {
      if (!PERSIST) __syncthreads();                     // every computing wave is done reading operand tiles (the staging
                                                         // rows alias the ring); PERSIST: a region of its own, wave-private
      float4 v[16];
#pragma unroll
      for (int pass = 0; pass < 64 / ER; ++pass) {
#pragma unroll
        for (int r2 = 0; r2 < ER / 16; ++r2)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float x = acc[0][pass * (ER / 16) + r2][ct][r];
              if constexpr (MODE == plp::kF16x3) x = fmaf(acc[1][pass * (ER / 16) + r2][ct][r], 1.0f / plp::kF16LoScale, x) * os;
              ldsw[(r2 * 16 + 4 * q + r) * 68 + ct * 16 + c] = x;
            }
#pragma unroll
        for (int it = 0; it < ER / 4; ++it)
          v[pass * (ER / 4) + it] = *reinterpret_cast<const float4*>(ldsw + (it * 4 + q) * 68 + lc);
      }
      float* C = k.e.C ? k.e.C + (k.e.split_k > 1 ? (size_t)slice * k.e.M * k.e.ldc : 0) : nullptr;
      staged_epilogue<EDGE>(k, C, v, m0, n0, wm, wn, lane);
    };
    if constexpr (WIDE) plp::planes_run16w<A_KS, B_KS, MODE, EDGE, CONV, PERSIST>(k.p, block_id, nblocks, nwork, lds, epi);
    else plp::planes_run16<A_KS, B_KS, MODE, EDGE, CONV, PERSIST, NST, T64>(k.p, block_id, nblocks, nwork, lds, epi);
  } else {
    float* ldsw = reinterpret_cast<float*>(lds) + (wave & 3) * 64 * 68;
    int m0, n0, slice;
    f32x16 acc[plp::ModeCfg<MODE>::NACC][2][2];
    if (!plp::planes_mainloop<A_KS, B_KS, 32, MODE, 4, 0, 3>(k.p, block_id, nwork, lds, acc, m0, n0, slice)) return;
    const int i = lane & 31, h = lane >> 5;
    if constexpr (MODE == plp::kF16x3) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            acc[0][a][b][r] = fmaf(acc[1][a][b][r], 1.0f / plp::kF16LoScale, acc[0][a][b][r]) * os;
    }
    float* C = k.e.C + (k.e.split_k > 1 ? (size_t)slice * k.e.M * k.e.ldc : 0);
    // dX of a residual block's first Linear adds the skip gradient (addend): through gemm_epilogue's accumulator layout
    // that is 64 dword loads per lane on 128-byte row segments and cost +14 us per launch (rocprofv3: 68.8 / 72.6 us for the
    // two launches with an addend against 59.8 / 54.7 without); staged, the addend moves as 16-byte accesses.
    const bool plain_out = k.e.split_k <= 1 && !k.e.bias && !k.e.stat_sum && !k.e.col_scale && !k.e.resid && !k.e.relu;
    if (plain_out && ((k.vec_addend && k.e.addend) || k.e.bnr_z)) {
      __syncthreads();
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            ldsw[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 68 + b * 32 + i] = acc[0][a][b][r];
      float4 v[16];
      const int lr = lane >> 4, lc = (lane & 15) * 4;
#pragma unroll
      for (int it = 0; it < 16; ++it) v[it] = *reinterpret_cast<const float4*>(ldsw + (it * 4 + lr) * 68 + lc);
      staged_epilogue<false>(k, C, v, m0, n0, wm, wn, lane);
      return;
    }
    gemm_epilogue<false, 2>(k.e, C, acc[0], m0, n0, wm, wn, i, h);
  }
}

template <bool A_KS, bool B_KS, int MODE, bool S16, bool EDGE = false, int CONV = 0>
__global__ __launch_bounds__(512) void planes_gemm_kernel(PlanesKern k) {
  __shared__ __attribute__((aligned(16))) char lds[lds_bytes<MODE>()];
  planes_body<A_KS, B_KS, MODE, S16, EDGE, CONV>(k, blockIdx.x, gridDim.x, gridDim.x, lds);
}

// one workgroup per CU, each walking nwork / gridDim.x consecutive work items (NT, 16x16x32 loop)
template <int MODE, bool EDGE, int CONV>
__global__ __launch_bounds__(512) void planes_gemm_persistent_kernel(PlanesKern k, int nwork) {
  __shared__ __attribute__((aligned(16))) char lds[lds_bytes<MODE, true>()];
  planes_body<false, false, MODE, true, EDGE, CONV, true>(k, blockIdx.x, gridDim.x, nwork, lds);
}

// NT on 256-row x 64-column tiles (gemm_planes16.h T64): the 64-channel layers
template <int MODE, bool EDGE, int CONV, bool PERSIST>
__global__ __launch_bounds__(512) void planes_gemm_t64_kernel(PlanesKern k, int nwork) {
  __shared__ __attribute__((aligned(16))) char lds[lds_bytes<MODE, PERSIST, false, true>()];
  planes_body<false, false, MODE, true, EDGE, CONV, PERSIST, false, true>(k, blockIdx.x, gridDim.x, nwork, lds);
}

// NT with the k-tiles staged in pairs (whole 128-byte lines of both k-contiguous operands: gemm_planes16.h, WIDE);
// PERSIST: nwork items over gridDim.x workgroups, else one item per workgroup (nwork == gridDim.x)
template <int MODE, bool EDGE, int CONV, bool PERSIST>
__global__ __launch_bounds__(512) void planes_gemm_wide_kernel(PlanesKern k, int nwork) {
  __shared__ __attribute__((aligned(16))) char lds[lds_bytes<MODE, PERSIST, true>()];
  planes_body<false, false, MODE, true, EDGE, CONV, PERSIST, true>(k, blockIdx.x, gridDim.x, nwork, lds);
}

// the wave's 64x64 accumulator block -> float4 rows through ER rows x 68 floats of LDS (64 / ER passes), then the staged epilogue
template <int MODE, int ER, bool EDGE>
__device__ __forceinline__ void acc_epilogue(const PlanesKern& k, plp::f32x4v (&acc)[plp::ModeCfg<MODE>::NACC][4][4], float* ldsw,
                                             const int m0, const int n0, const int slice, const int wm, const int wn, const int lane) {
  const float os = MODE == plp::kF16x3 ? (k.dyn_inv ? k.out_scale * k.dyn_inv[0] : k.out_scale) : 1.f;
  const int q = lane >> 4, c = lane & 15, lc = c * 4;
  float4 v[16];
#pragma unroll
  for (int pass = 0; pass < 64 / ER; ++pass) {
#pragma unroll
    for (int r2 = 0; r2 < ER / 16; ++r2)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[0][pass * (ER / 16) + r2][ct][r];
          if constexpr (MODE == plp::kF16x3) x = fmaf(acc[1][pass * (ER / 16) + r2][ct][r], 1.0f / plp::kF16LoScale, x) * os;
          ldsw[(r2 * 16 + 4 * q + r) * 68 + ct * 16 + c] = x;
        }
#pragma unroll
    for (int it = 0; it < ER / 4; ++it)
      v[pass * (ER / 4) + it] = *reinterpret_cast<const float4*>(ldsw + (it * 4 + q) * 68 + lc);
  }
  float* C = k.e.C ? k.e.C + (k.e.split_k > 1 ? (size_t)slice * k.e.M * k.e.ldc : 0) : nullptr;
  staged_epilogue<EDGE>(k, C, v, m0, n0, wm, wn, lane);
}

// the backward pair chained in one workgroup per CU (gemm_planes16.h CHAIN): grid = the number of dX tiles = dW items
template <int MODE>
__global__ __launch_bounds__(512) void planes_gemm_chain_kernel(PlanesKern k0, PlanesKern k1) {
  __shared__ __attribute__((aligned(16))) char lds[ring_bytes<MODE>() + stage_bytes<32>()];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave & 3) >> 1, wn = wave & 1;
  float* ldsw = reinterpret_cast<float*>(lds + ring_bytes<MODE>()) + (wave & 3) * 32 * 68;
  auto epi0 = [&](plp::f32x4v (&acc)[plp::ModeCfg<MODE>::NACC][4][4], const int m0, const int n0, const int slice) {
    acc_epilogue<MODE, 32, false>(k0, acc, ldsw, m0, n0, slice, wm, wn, lane);
  };
  auto epi1 = [&](plp::f32x4v (&acc)[plp::ModeCfg<MODE>::NACC][4][4], const int m0, const int n0, const int slice) {
    acc_epilogue<MODE, 32, false>(k1, acc, ldsw, m0, n0, slice, wm, wn, lane);
  };
  plp::planes_run16_chain<MODE>(k0.p, k1.p, blockIdx.x, gridDim.x, lds, epi0, epi1);
}

// backward pair of one layer in one launch: workgroups [0, n0) run dX = dz W (NN), the rest dW = dz^T a (TN, split-K
// slabs).  One workgroup per CU at a time (96 KB of LDS each): what the single launch saves is the launch boundary
// and the tail of the first problem, which the second one's workgroups fill.
// (Tried: alternating the two problems in groups of 8 workgroups, so that half the CUs run dX's heavy epilogues beside the
//  other half's main loops -- same-box A/B 0.688 vs 0.629 ms per step: two working sets per XCD L2 cost far more.)
// (Tried in round 3, xsplit = 1 / POSELIFT_DUAL_XSPLIT=1: the two problems taking turns PER XCD -- XCDs 0-3 run their share
//  of dX first and of dW second, XCDs 4-7 the other way round (blocks b and b + 8 share an XCD and are dispatched in order), so
//  that each L2 still holds ONE problem's working set at a time while dX's 50 MB epilogue burst comes in two halves, each
//  beside the other half's main loops.  Same-box A/B over three interleaved runs: 68.5 us per dual launch against 65.1 us,
//  0.650 against 0.638 ms per step.  Not kept -- the burst is not what a half-chip of main loops can hide.)
template <int MODE, bool S16>
__global__ __launch_bounds__(512) void planes_gemm_dual_kernel(PlanesKern k0, PlanesKern k1, int n0, int xsplit) {
  __shared__ __attribute__((aligned(16))) char lds[lds_bytes<MODE>()];
  const int b = blockIdx.x;
  if (xsplit) {                                   // (host: n0 == gridDim.x - n0, n0 % 8 == 0)
    const bool second = b >= n0;
    const int bb = second ? b - n0 : b;
    if (((b & 7) < 4) != second)
      planes_body<false, true, MODE, S16>(k0, bb, n0, n0, lds);
    else
      planes_body<true, true, MODE, S16>(k1, bb, n0, n0, lds);
    return;
  }
  if (b < n0)
    planes_body<false, true, MODE, S16>(k0, b, n0, n0, lds);
  else
    planes_body<true, true, MODE, S16>(k1, b - n0, gridDim.x - n0, gridDim.x - n0, lds);
}

bool mfma16_shape() {
  static const bool s16 = [] { const char* e = getenv("POSELIFT_MFMA32"); return !(e && e[0] == '1'); }();
  return s16;
}

int npl_of(int mode) { return mode == plp::kF16x3 ? 2 : (mode == plp::kBf16x6 ? 3 : 1); }

PlanesKern kern_of(GemmLayout layout, const PlanesGemmArgs& a) {
  PlanesKern k = {};
  k.p.A = reinterpret_cast<const __bf16*>(a.A);
  k.p.B = reinterpret_cast<const __bf16*>(a.B);
  k.p.C = a.e.C;
  k.p.M = a.e.M; k.p.N = a.e.N; k.p.K = a.e.K;
  k.p.lda = a.lda; k.p.ldb = a.ldb; k.p.ldc = a.e.ldc;
  k.p.split_k = a.e.split_k;
  k.p.a_plane = (size_t)a.a_plane; k.p.b_plane = (size_t)a.b_plane;
  k.e = a.e;
  k.p.cv_cin = a.e.conv_cin; k.p.cv_h = a.e.conv_h; k.p.cv_w = a.e.conv_w; k.p.cv_ho = a.e.conv_ho; k.p.cv_wo = a.e.conv_wo;
  k.p.cv_kw = a.e.conv_kw; k.p.cv_stride = a.e.conv_stride; k.p.cv_pad_h = a.e.conv_pad_h; k.p.cv_pad_w = a.e.conv_pad_w;
  k.p.cv_stride_w = a.e.conv_stride_w > 0 ? a.e.conv_stride_w : a.e.conv_stride;
  k.out_scale = a.out_scale;
  k.dyn_inv = a.dyn_inv;
  static const int vec = [] { const char* e = getenv("POSELIFT_ADDEND_SCALAR"); return (e && e[0] == '1') ? 0 : 1; }();
  k.vec_addend = vec;
  // POSELIFT_ABL (timing-only, wrong results): 1 = the epilogue stores nothing to C, 2 = no operand DMA, 4 = no MFMAs
  static const int abl = [] { const char* e = getenv("POSELIFT_ABL"); return e ? atoi(e) : 0; }();
  k.p.abl = abl;
  static const int ntc = [] { const char* e = getenv("POSELIFT_NT_C"); return e ? atoi(e) : 1; }();   // =0: same-box A/B
  k.nt_store = (ntc && nontemporal_on() && (int64_t)a.e.M * a.e.N * 4 >= kNontemporalBytes) ? 1 : 0;
  if (abl & 1) { k.e.C = nullptr; k.p.C = nullptr; }
  (void)layout;
  return k;
}

// CUs of the current device (the persistent form launches one workgroup per CU)
int cu_count() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cached[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

int grid_of(const PlanesGemmArgs& a) {
  return ((a.e.M + 127) / 128) * ((a.e.N + 127) / 128) * (a.e.split_k > 1 ? a.e.split_k : 1);
}
bool is_edge(const PlanesGemmArgs& a) { return (a.e.M % 128) != 0 || (a.e.N % 128) != 0; }

}  // namespace

// whole 128x128 tiles, every K slice a whole number of 32-k tiles, 16-byte aligned plane rows, and every byte
// offset the DMA's 32-bit scalar offset has to hold below 2^31
static int splits_of(const GemmArgs& e) { return e.split_k > 1 ? e.split_k : 1; }
bool planes_gemm_ok(GemmLayout layout, const PlanesGemmArgs& a) {
  const GemmArgs& e = a.e;
  if (a.mode != plp::kBf16 && a.mode != plp::kF16x3) return false;
  if (!a.A || !a.B || (!e.C && !e.cpl_kind) || e.M <= 0 || e.N <= 0 || e.K <= 0) return false;
  if (e.cpl_kind && (!mfma16_shape() || splits_of(e) > 1 || !e.cpl_h || (e.cpl_kind == 2 && !e.cpl_l) ||
                     ((reinterpret_cast<uintptr_t>(e.cpl_h) | reinterpret_cast<uintptr_t>(e.cpl_l)) & 7)))
    return false;
  const int splits = e.split_k > 1 ? e.split_k : 1;
  if (e.K % (32 * splits)) return false;
  if (e.conv_cin) {
    // implicit-GEMM convolution: NT with A gathered (forward / data gradient) or TN with B gathered (weight gradient);
    // the gathered tensor's bytes (every plane) must fit the DMA's 32-bit offsets
    if (!mfma16_shape() || (layout != kNT && layout != kTN) || e.bnr_z) return false;
    if (e.conv_h <= 0 || e.conv_w <= 0 || e.conv_ho <= 0 || e.conv_wo <= 0 || e.conv_kw <= 0 || e.conv_stride <= 0) return false;
    const int64_t pixels = layout == kNT ? e.M : e.K;
    if (pixels % ((int64_t)e.conv_ho * e.conv_wo)) return false;
    const int64_t xb = pixels / ((int64_t)e.conv_ho * e.conv_wo) * e.conv_h * e.conv_w * e.conv_cin * 2;
    const int npl_ = a.mode == plp::kF16x3 ? 2 : 1;
    if (xb >= 0x7fffffe0ll) return false;
    (void)npl_;
    // (8 channels: four taps of a kernel row per 32-k tile -- the stem on pixel pairs; one K slice, no pair staging)
    const bool cin8 = e.conv_cin == 8 && e.conv_kw % 4 == 0 && e.split_k <= 1;
    if (layout == kNT && !cin8 && ((e.conv_cin & 31) || e.K % e.conv_cin)) return false;
    if (e.scat_on && (layout != kNT || e.split_k > 1 || e.addend || e.resid || e.relu == 2 || e.stat_sum)) return false;
    if (layout == kTN && ((e.conv_cin & 7) || e.N % e.conv_cin)) return false;
  }
  if (e.M % 128 || e.N % 128) {
    // overhanging tiles (16x16x32 loop, guarded epilogue): the source of an overhanging lane is clamped to the last valid
    // row / 8-column chunk, so a k-strided operand's extent must be a multiple of 8; no BatchNorm-backward epilogue there
    if (!mfma16_shape() || (e.N & 7) || (layout == kTN && (e.M & 7)) || e.bnr_z) return false;
  }
  if ((a.lda & 7) || (a.ldb & 7)) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B)) & 15) return false;
  if (((a.a_plane | a.b_plane) & 7) != 0) return false;
  const int npl = npl_of(a.mode);
  const bool a_ks = layout == kTN, b_ks = layout != kNT;
  // byte offsets inside ONE plane and ONE K slice, from the tile origin (the 16x16x32 loop: plane stride and slice origin are
  // in the descriptor's 64-bit base; the 32x32x16 loop adds the plane stride to the 32-bit scalar offset)
  const int64_t ks = e.K / splits;
  const int64_t pa = mfma16_shape() ? 0 : (int64_t)(npl - 1) * a.a_plane * 2, pb = mfma16_shape() ? 0 : (int64_t)(npl - 1) * a.b_plane * 2;
  const int64_t a_ext = pa + (a_ks ? ks * a.lda * 2 : (int64_t)128 * a.lda * 2 + ks * 2);
  const int64_t b_ext = pb + (b_ks ? ks * a.ldb * 2 : (int64_t)128 * a.ldb * 2 + ks * 2);
  if (a_ext >= (1ll << 31) || b_ext >= (1ll << 31)) return false;
  if ((e.stat_sum != nullptr) != (e.stat_m2 != nullptr)) return false;
  if ((e.col_scale != nullptr) != (e.col_shift != nullptr)) return false;
  // the staged epilogue moves everything as 16-byte accesses
  if (e.ldc & 3) return false;
  const uintptr_t al = reinterpret_cast<uintptr_t>(e.C) | reinterpret_cast<uintptr_t>(e.bias) | reinterpret_cast<uintptr_t>(e.addend) |
                       reinterpret_cast<uintptr_t>(e.resid) | reinterpret_cast<uintptr_t>(e.col_scale) |
                       reinterpret_cast<uintptr_t>(e.col_shift) | reinterpret_cast<uintptr_t>(e.stat_sum) |
                       reinterpret_cast<uintptr_t>(e.stat_m2) | reinterpret_cast<uintptr_t>(e.bnr_z) |
                       reinterpret_cast<uintptr_t>(e.bnr_mean) | reinterpret_cast<uintptr_t>(e.bnr_rstd) |
                       reinterpret_cast<uintptr_t>(e.bnr_part_dy) | reinterpret_cast<uintptr_t>(e.bnr_part_dyz);
  if (al & 15) return false;
  return true;
}

int launch_gemm_planes(GemmLayout layout, const PlanesGemmArgs& a, hipStream_t s) {
  if (!planes_gemm_ok(layout, a)) PL_FAIL(PL_ESHAPE, "gemm_planes: unsupported problem %dx%dx%d (layout %d, mode %d)", a.e.M, a.e.N, a.e.K, (int)layout, a.mode);
  const PlanesKern k = kern_of(layout, a);
  const dim3 grid(grid_of(a)), block(512);
  void* prof = prof_begin_flops(2.0 * a.e.M * a.e.N * a.e.K, s);
#define PL_PLANES_LAUNCH(MODE, S16)                                                                              \
  switch (layout) {                                                                                              \
    case kNT: hipLaunchKernelGGL((planes_gemm_kernel<false, false, MODE, S16>), grid, block, 0, s, k); break;    \
    case kNN: hipLaunchKernelGGL((planes_gemm_kernel<false, true, MODE, S16>), grid, block, 0, s, k); break;     \
    case kTN: hipLaunchKernelGGL((planes_gemm_kernel<true, true, MODE, S16>), grid, block, 0, s, k); break;      \
    default: PL_FAIL(PL_EINVAL, "gemm_planes: bad layout %d", (int)layout);                                     \
  }
  // persistent form (round 3): NT problems with at least two work items per CU -- the conv path's 1x1 / 3x3 / transposed
  // convolutions over 16K .. 1M pixels.  POSELIFT_PERSIST=0: one workgroup per item, as round 2 (same-box A/B).
  static const int persist_env = [] { const char* e = getenv("POSELIFT_PERSIST"); return e ? atoi(e) : 1; }();
  static const int wide_env = [] { const char* e = getenv("POSELIFT_WIDE"); return e ? atoi(e) : 1; }();   // =0: same-box A/B
  const int ncu = cu_count();
  const bool persist = persist_env && mfma16_shape() && layout == kNT && (int)grid.x >= 2 * ncu;
  // pair-staged k-tiles: NT, every K slice a whole number of 64-k pairs, a gathered pair inside one filter tap -- and a long
  // contraction: measured same-box (tools/bench_conv_gemm.py, bench.py) K = 1024 / 2048 take 4-6 % less time (the lifter's
  // forward GEMM 33.2 -> 32.7 us in the step), K <= 512 0-3 % MORE (a pair is a coarser unit at an item boundary)
  const bool wide = wide_env && mfma16_shape() && layout == kNT && splits_of(a.e) == 1 && a.e.K % 64 == 0 && a.e.K >= 1024 &&
                    (a.e.conv_cin == 0 || a.e.conv_cin % 64 == 0);
  // 64-channel outputs: 256-row x 64-column tiles (half of a 128-wide tile would be padding)
  static const int t64_env = [] { const char* e = getenv("POSELIFT_T64"); return e ? atoi(e) : 1; }();       // =0: same-box A/B
  const bool t64 = t64_env && mfma16_shape() && layout == kNT && a.e.N <= 64 && splits_of(a.e) == 1 && !a.e.scat_on;
  if (t64) {
    const int nwork = (int)(((a.e.M + 255) / 256) * ((a.e.N + 63) / 64));
    const bool pers = persist_env && nwork >= 2 * ncu;
    const dim3 tg(pers ? ncu : nwork);
    const bool edge = (a.e.M % 256) != 0 || (a.e.N % 64) != 0 || a.e.conv_cin;
#define PL_T64(MODE, P)                                                                                                  \
    if (a.e.conv_cin) hipLaunchKernelGGL((planes_gemm_t64_kernel<MODE, true, 1, P>), tg, block, 0, s, k, nwork);          \
    else if (edge) hipLaunchKernelGGL((planes_gemm_t64_kernel<MODE, true, 0, P>), tg, block, 0, s, k, nwork);             \
    else hipLaunchKernelGGL((planes_gemm_t64_kernel<MODE, false, 0, P>), tg, block, 0, s, k, nwork);
    if (pers) { if (a.mode == plp::kF16x3) { PL_T64(plp::kF16x3, true) } else { PL_T64(plp::kBf16, true) } }
    else { if (a.mode == plp::kF16x3) { PL_T64(plp::kF16x3, false) } else { PL_T64(plp::kBf16, false) } }
#undef PL_T64
  } else if (wide) {
    const int nwork = (int)grid.x;
    const dim3 wg(persist ? ncu : nwork);
    const bool edge = is_edge(a) || a.e.conv_cin;
#define PL_WIDE(MODE, P)                                                                                                  \
    if (a.e.conv_cin) hipLaunchKernelGGL((planes_gemm_wide_kernel<MODE, true, 1, P>), wg, block, 0, s, k, nwork);          \
    else if (edge) hipLaunchKernelGGL((planes_gemm_wide_kernel<MODE, true, 0, P>), wg, block, 0, s, k, nwork);             \
    else hipLaunchKernelGGL((planes_gemm_wide_kernel<MODE, false, 0, P>), wg, block, 0, s, k, nwork);
    if (persist) { if (a.mode == plp::kF16x3) { PL_WIDE(plp::kF16x3, true) } else { PL_WIDE(plp::kBf16, true) } }
    else { if (a.mode == plp::kF16x3) { PL_WIDE(plp::kF16x3, false) } else { PL_WIDE(plp::kBf16, false) } }
#undef PL_WIDE
  } else if (persist) {
    const dim3 pg(ncu);
    const int nwork = (int)grid.x;
    const bool edge = is_edge(a) || a.e.conv_cin;
#define PL_PERSIST(MODE)                                                                                               \
    if (a.e.conv_cin) hipLaunchKernelGGL((planes_gemm_persistent_kernel<MODE, true, 1>), pg, block, 0, s, k, nwork);   \
    else if (edge) hipLaunchKernelGGL((planes_gemm_persistent_kernel<MODE, true, 0>), pg, block, 0, s, k, nwork);      \
    else hipLaunchKernelGGL((planes_gemm_persistent_kernel<MODE, false, 0>), pg, block, 0, s, k, nwork);
    if (a.mode == plp::kF16x3) { PL_PERSIST(plp::kF16x3) } else { PL_PERSIST(plp::kBf16) }
#undef PL_PERSIST
  } else if (a.e.conv_cin) {
    if (layout == kNT) {
      if (a.mode == plp::kF16x3) hipLaunchKernelGGL((planes_gemm_kernel<false, false, plp::kF16x3, true, true, 1>), grid, block, 0, s, k);
      else hipLaunchKernelGGL((planes_gemm_kernel<false, false, plp::kBf16, true, true, 1>), grid, block, 0, s, k);
    } else {
      if (a.mode == plp::kF16x3) hipLaunchKernelGGL((planes_gemm_kernel<true, true, plp::kF16x3, true, true, 2>), grid, block, 0, s, k);
      else hipLaunchKernelGGL((planes_gemm_kernel<true, true, plp::kBf16, true, true, 2>), grid, block, 0, s, k);
    }
  } else if (is_edge(a)) {
#define PL_PLANES_LAUNCH_EDGE(MODE)                                                                                    \
  switch (layout) {                                                                                                    \
    case kNT: hipLaunchKernelGGL((planes_gemm_kernel<false, false, MODE, true, true>), grid, block, 0, s, k); break;   \
    case kNN: hipLaunchKernelGGL((planes_gemm_kernel<false, true, MODE, true, true>), grid, block, 0, s, k); break;    \
    case kTN: hipLaunchKernelGGL((planes_gemm_kernel<true, true, MODE, true, true>), grid, block, 0, s, k); break;     \
    default: PL_FAIL(PL_EINVAL, "gemm_planes: bad layout %d", (int)layout);                                           \
  }
    if (a.mode == plp::kF16x3) { PL_PLANES_LAUNCH_EDGE(plp::kF16x3) } else { PL_PLANES_LAUNCH_EDGE(plp::kBf16) }
#undef PL_PLANES_LAUNCH_EDGE
  } else if (mfma16_shape()) {
    if (a.mode == plp::kF16x3) { PL_PLANES_LAUNCH(plp::kF16x3, true) } else { PL_PLANES_LAUNCH(plp::kBf16, true) }
  } else {
    if (a.mode == plp::kF16x3) { PL_PLANES_LAUNCH(plp::kF16x3, false) } else { PL_PLANES_LAUNCH(plp::kBf16, false) }
  }
#undef PL_PLANES_LAUNCH
  prof_end(prof, s);
  PL_CHECK_LAUNCH("gemm_planes");
  return PL_OK;
}

int launch_gemm_planes_pair(const PlanesGemmArgs& nn, const PlanesGemmArgs& tn, hipStream_t s) {
  if (!planes_gemm_ok(kNN, nn) || !planes_gemm_ok(kTN, tn) || nn.mode != tn.mode || is_edge(nn) || is_edge(tn))
    PL_FAIL(PL_ESHAPE, "gemm_planes_pair: unsupported problems");
  const PlanesKern k0 = kern_of(kNN, nn), k1 = kern_of(kTN, tn);
  const int g0 = grid_of(nn), g1 = grid_of(tn);
  void* prof = prof_begin_flops(2.0 * nn.e.M * nn.e.N * nn.e.K + 2.0 * tn.e.M * tn.e.N * tn.e.K, s);
  // chained form (round 3): one workgroup per dX tile that goes on to a dW item, operands streaming across
  static const int chain_env = [] { const char* e = getenv("POSELIFT_CHAIN"); return e ? atoi(e) : 1; }();   // =0: same-box A/B
  if (chain_env && mfma16_shape() && g0 == g1 && nn.e.K % 32 == 0 && nn.e.K >= 64 && tn.e.K % (32 * splits_of(tn.e)) == 0 &&
      tn.e.K >= 32 * splits_of(tn.e)) {
    const dim3 cg(g0), cb(512);
    if (nn.mode == plp::kF16x3) hipLaunchKernelGGL((planes_gemm_chain_kernel<plp::kF16x3>), cg, cb, 0, s, k0, k1);
    else hipLaunchKernelGGL((planes_gemm_chain_kernel<plp::kBf16>), cg, cb, 0, s, k0, k1);
    prof_end(prof, s);
    PL_CHECK_LAUNCH("gemm_planes_chain");
    return PL_OK;
  }
  const dim3 grid(g0 + g1), block(512);
  static const int xs_env = [] { const char* e = getenv("POSELIFT_DUAL_XSPLIT"); return e ? atoi(e) : 0; }();   // =1: same-box A/B (not kept)
  const int xs = (xs_env && g0 == g1 && (g0 & 7) == 0) ? 1 : 0;
  if (mfma16_shape()) {
    if (nn.mode == plp::kF16x3) hipLaunchKernelGGL((planes_gemm_dual_kernel<plp::kF16x3, true>), grid, block, 0, s, k0, k1, g0, xs);
    else hipLaunchKernelGGL((planes_gemm_dual_kernel<plp::kBf16, true>), grid, block, 0, s, k0, k1, g0, xs);
  } else {
    if (nn.mode == plp::kF16x3) hipLaunchKernelGGL((planes_gemm_dual_kernel<plp::kF16x3, false>), grid, block, 0, s, k0, k1, g0, xs);
    else hipLaunchKernelGGL((planes_gemm_dual_kernel<plp::kBf16, false>), grid, block, 0, s, k0, k1, g0, xs);
  }
  prof_end(prof, s);
  PL_CHECK_LAUNCH("gemm_planes_dual");
  return PL_OK;
}

}  // namespace pl
