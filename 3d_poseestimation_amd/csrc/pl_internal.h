// Internal declarations shared by the HIP translation units of libposelift.so.
// gfx950 (MI355X) only: 64-lane wavefronts, MFMA, 160 KB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/poselift.h"

namespace pl {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

#define PL_FAIL(code, ...)        \
  do {                            \
    ::pl::set_error(__VA_ARGS__); \
    return (code);                \
  } while (0)

#define PL_CHECK_LAUNCH(what)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess)                                                 \
      PL_FAIL(PL_EHIP, "%s: launch failed: %s", what, hipGetErrorString(e__)); \
  } while (0)

#define PL_TRY(expr)        \
  do {                      \
    int rc__ = (expr);      \
    if (rc__ != PL_OK) return rc__; \
  } while (0)

// ---------------------------------------------------------------------------------
// fp32 MFMA GEMM  (gemm_f32.hip)
// ---------------------------------------------------------------------------------
enum GemmLayout { kNT = 0, kNN = 1, kTN = 2 };

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  int split_k;              // >1: slice z writes C + z*M*ldc (slab), no epilogue extras
  // epilogue (all optional)
  const float* bias;        // [N]     v += bias[col]
  const float* addend;      // [M][ldc] v += addend[row][col]   (may alias C)
  float* stat_sum;          // [2*ceil(M/128)][N] per-64-row-group column sums of v
  float* stat_m2;           //   ... and sums of squares about the group mean
  const float* col_scale;   // [N]     v = v*scale[col] + shift[col]  (eval-mode BN fold)
  const float* col_shift;
  int relu;                 // 1: v = max(v, 0) before the resid add; 2: after it (Bottleneck order)
  const float* resid;       // [M][ldc] v += resid[row][col] after relu (may alias C)
  int arith;                // 0 fp32 MFMA; 1 PL_BF16: operands rounded to bf16 on the way to the MFMA;
                            // 2 PL_BF16X6: three-way bf16 split, six MFMAs, fp32-grade products
                            // (whole-tile problems only; edge problems stay fp32)
  // Implicit-GEMM convolution (launch_conv_nhwc only; conv_cin == 0 otherwise): A is not a matrix but the
  // NHWC input x [B][H][W][Cin]; row m = (b, oh, ow), k = (kh*KW + kw)*Cin + ci, element
  // x[b][oh*stride - pad + kh][ow*stride - pad + kw][ci] or 0 outside the image.  B = weights [N][K] (OHWI).
  int conv_cin, conv_h, conv_w, conv_ho, conv_wo, conv_kw, conv_stride, conv_pad_h, conv_pad_w;
  int conv_stride_w;        // planes convolutions: stride along w when it differs from conv_stride (along h); 0 = the same
  // BatchNorm-backward pass 1 of the layer BELOW, folded into the epilogue of the dX GEMM that produces its incoming
  // gradient (planes GEMM only; bnr_z == NULL otherwise): C is g = d(act of that layer); with its saved z, ReLU/dropout
  // bitmap, batch mean and rstd the epilogue also emits, per 64-row block and column, sum dy and sum dy*zhat
  // (dy = g * bit * bnr_kscale) -- [M/64][N] each -- and per (64-row block, 64-column block) max |dy| and max |zhat|.
  const float* bnr_z;
  const uint64_t* bnr_bits;
  const float* bnr_mean;
  const float* bnr_rstd;
  float bnr_kscale;
  float* bnr_part_dy;       // [M/64][N]
  float* bnr_part_dyz;      // [M/64][N]
  float* bnr_amax;          // [(M/64) * (N/64)][2] or NULL
  // planes GEMM (staged epilogue): the result ALSO (C != NULL) or ONLY (C == NULL) as operand planes of the next GEMM --
  // cpl_kind 0 none, 1 one bf16 plane, 2 fp16 pair (h, l) of cpl_scale * value; addressed like C (row * ldc + col)
  unsigned short* cpl_h; unsigned short* cpl_l; float cpl_scale; int cpl_kind;
  // planes convolution launches only: row m = (b, a, c) of a conv_ho x conv_wo grid is stored at pixel
  // (b, 2a + scat_ph, 2c + scat_pw) of a [B][2 conv_ho][2 conv_wo][ldc] map (one output parity of a 4x4 stride-2
  // transposed convolution); plain stores only
  int scat_on, scat_ph, scat_pw;
  // host side only: scratch for the small-batch split-K path (gemm_thin.hip); NULL = never take it
  float* thin_scratch;
  size_t thin_scratch_floats;
};

// y = conv2d(x, w) as an implicit GEMM on the PL_BF16X6 planes pipeline; a.A = x, a.B = w [Cout][KH*KW*Cin],
// a.M = B*Ho*Wo, a.N = Cout, a.K = KH*KW*Cin and the conv_* fields filled in; whole tiles only.
int launch_conv_nhwc(const GemmArgs& a, hipStream_t s);
// four independent convolutions of identical shape (the four output parities of a stride-2 transposed
// convolution) in ONE launch: at 8x8 / 16x16 input maps a single one fills a quarter of the CUs
int launch_conv_nhwc_group4(const GemmArgs* a, hipStream_t s);
// weight gradient of a convolution: the TN planes GEMM over pixels with the B operand gathered from x
int launch_conv_wgrad(const GemmArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------------
// planes GEMM (gemm_planes.hip / gemm_planes.h): operands pre-split into 16-bit planes by their producers
// ---------------------------------------------------------------------------------
// Where a streaming kernel writes the planes of the tensor it produces ([rows][cols] row-major, like the tensor):
//   kind 0 none; 1 one bf16 plane (PL_BF16 storage); 2 fp16 pair (PL_F16X3): h = fp16(S x), l = fp16((S x - h) * 2048)
// S = scale, or -- dyn != NULL -- the device value dyn[0] (a power of two written by an earlier kernel).
struct PlaneOut {
  unsigned short* h;
  unsigned short* l;
  float scale;
  const float* dyn;
  int kind;
  int nt;             // the tensor (planes and its fp32 twin) leaves with nontemporal stores: far larger than the caches
};
// outputs at least this large are streamed past the caches (their consumer reads them from HBM whatever the store policy)
constexpr long long kNontemporalBytes = 64ll << 20;
bool nontemporal_on();     // POSELIFT_NT=0: plain stores everywhere (same-box A/B)
constexpr float kActPlaneScale = 1.0f;      // activations: fp16 covers 6e-5 .. 65504 in h, the remainder in l
constexpr float kWeightPlaneScale = 16.0f;  // weights (|w| ~ 0.03 at init): 3.8e-6 .. 4094
// conv path: eval-mode feature maps of an unnormalised network reach 1e5 (seeded test weights: 6.7e4 after the first
// transposed convolution) -- 1/64 keeps h finite up to 4.2e6; below |x| = 4e-3 h goes subnormal and l carries the value to
// an absolute 1e-9 (relative 2^-22 above)
constexpr float kConvActPlaneScale = 1.0f / 64.0f;

struct PlanesGemmArgs {
  GemmArgs e;                 // M, N, K, C, ldc, split_k and the epilogue fields (A, B, lda, ldb, arith unused)
  const unsigned short* A;    // planes of A: k-contiguous [M][K] (NT, NN) or k-strided [K][M] (TN)
  const unsigned short* B;    // planes of B: k-contiguous [N][K] (NT) or k-strided [K][N] (NN, TN)
  int64_t a_plane, b_plane;   // elements between two planes of one tensor
  int lda, ldb;               // row strides (elements) of the plane matrices
  int mode;                   // plp::PlanesMode: 0 bf16 (one plane), 2 f16x3 (two planes)
  float out_scale;            // acc * out_scale [* dyn_inv[0]] before the epilogue: 1 / (S_A S_B)
  const float* dyn_inv;       // device scalar or NULL
};
bool planes_gemm_ok(GemmLayout layout, const PlanesGemmArgs& a);
int launch_gemm_planes(GemmLayout layout, const PlanesGemmArgs& a, hipStream_t s);
// dX = dz W (NN) and dW = dz^T a (TN, split-K slabs) of one layer in ONE launch
int launch_gemm_planes_pair(const PlanesGemmArgs& nn, const PlanesGemmArgs& tn, hipStream_t s);
// x [n] fp32 -> planes (static scale); n % 4 == 0, 16-byte aligned
int launch_split_planes(const float* x, int64_t n, const PlaneOut& out, hipStream_t s);
// PlaneOut of a caller's planes buffer ([2][n] fp16 for PL_F16X3, [n] bf16 for PL_BF16; NULL -> kind 0)
int plane_out_of(int mode, void* planes, int64_t n, float scale, const float* dyn, PlaneOut* po, const char* who);

int launch_gemm_f32(GemmLayout layout, const GemmArgs& a, hipStream_t s);
// M <= thin_gemm_max_m() rows (NT, NN): the contraction split over the chip, exact fp32 MFMA, slabs + one reduce/epilogue
// launch (gemm_thin.hip); launch_gemm_f32 takes it for problems that are not whole tiles when a.thin_scratch is set
int thin_gemm_max_m();
bool thin_gemm_ok(GemmLayout layout, const GemmArgs& a);
size_t thin_gemm_scratch_floats(int M, int N, int K);
int launch_gemm_thin(GemmLayout layout, const GemmArgs& a, float* scratch, size_t scratch_floats, hipStream_t s);
int launch_gemm_f32_pair(const GemmArgs& nn, const GemmArgs& tn, hipStream_t s);
int gemm_stat_groups(int M);  // number of 64-row groups the stats epilogue emits
int prof_enable(int on);
void* prof_begin_flops(double flops, hipStream_t s);   // NULL when this launch is not sampled
void prof_end(void* rec, hipStream_t s);
int prof_read(double min_flops, double max_flops, double* ms_total, int64_t* launches, double* flops_total);

// ---------------------------------------------------------------------------------
// streaming kernels (elementwise.hip)
// ---------------------------------------------------------------------------------
inline int bitmap_words_per_row(int H) { return ((H + 255) / 256) * 4; }

// BN statistics: merge per-group (sum, M2) partials -> mean/rstd/scale/shift, update running
// (stat = [world][2][G][H]: per rank G rows of sums then G rows of M2; B = rows per rank)
int launch_bn_finalize(const float* stat, int G, int world, int B, int H,
                       const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, int64_t* batches,
                       float* mean, float* rstd, float* scale, float* shift, hipStream_t s, int group_rows = 64);

// act = [resid +] dropout(relu(z*scale + shift)); bits = keep&positive bitmap
// planes: also (act == NULL: only) write the activation as GEMM operand planes
// finalize != NULL: the statistics finalize (launch_bn_finalize's job: local statistics, at most 16 groups) inside this launch;
// scale / shift are then not read
struct BnFinalizeArgs {
  const float* stat;            // [2][G][H] partials of this process
  int G, group_rows;
  const float *gamma, *beta;
  float eps, momentum;
  float *running_mean, *running_var;
  int64_t* batches;
  float *mean, *rstd;
};
int launch_bn_apply(const float* z, const float* scale, const float* shift, const float* resid,
                    float* act, uint64_t* bits, int B, int H, float p, uint64_t seed,
                    uint64_t step, int layer, const uint64_t* inject_keep, hipStream_t s,
                    const PlaneOut* planes = nullptr, const uint64_t* step_dev = nullptr,
                    const BnFinalizeArgs* finalize = nullptr);
// pl_mse_fwd_bwd + (tick != NULL) tick[0] += 1 once the loss is written (PLDesc.step_dev, graph replay)
int mse_fwd_bwd_tick(const float* pred, const float* tgt, int64_t n, float grad_scale, float* dpred, float* loss_out,
                     void* scratch, uint64_t* tick, void* stream);

int bwd_row_chunks(int B, int H);
// pass 1: partial column sums of dy and dy*zhat, dy = g * bits * keep_scale
// part_amax (optional): [strips * RC][2] per-workgroup maxima of |dy| and |zhat| (the dz range bound, below)
int launch_bn_bwd_reduce(const float* g, const uint64_t* bits, const float* z, const float* mean,
                         const float* rstd, float keep_scale, int B, int H, float* part_dy,
                         float* part_dyz, hipStream_t s, int Hc = 0, float* part_amax = nullptr, int rc = 0,
                         const float* join_g2 = nullptr, float* join_dx = nullptr);

// dz_scale (optional, with part_amax of n_amax workgroups): {S, 1/S}, S the power of two that maps the bound
// max|c0| max|dy| (2 + max|zhat|) >= max|dz| to at most 2^14 (fp16 planes of dz, PL_F16X3)
int launch_bn_bwd_finalize(const float* part, int RC, int world, int rank, int B, int H,
                           const float* gamma, const float* rstd, float* coef, float* dgamma,
                           float* dbeta, hipStream_t s, const float* part_amax = nullptr, int n_amax = 0,
                           float* dz_scale = nullptr, int eval_mode = 0, int64_t rstride = 0, int amax_world = 1);
// Small batches (B <= kBnSmallRows, local statistics, fp32 outputs): statistics + finalize + apply of a hidden layer in ONE
// launch (a workgroup owns a 256-column strip for all rows, held in registers), and pass 1 + finalize + dz + bias gradient
// of its backward.  Measured same-box (bench.py --batch B): B = 64 0.305 -> 0.280 ms per step; B = 100 / 127 with sixteen
// rows per wave 0.42 / 0.45 -> 0.47 / 0.52 ms (slower: kept to 64 rows).
constexpr int kBnSmallRows = 64;
int launch_bn_small_fwd(const float* z, const float* gamma, const float* beta, float eps, float momentum, float* rm, float* rv,
                        int64_t* nbt, float* mean, float* rstd, const float* resid, float* act, uint64_t* bits, int B, int H,
                        float p, uint64_t seed, uint64_t step, int layer, const uint64_t* inject_keep, hipStream_t s,
                        const uint64_t* step_dev);
int launch_bn_small_bwd(const float* g, const uint64_t* bits, const float* z, const float* mean, const float* rstd,
                        const float* gamma, float keep_scale, int B, int H, float* dz, float* dgamma, float* dbeta, float* dbias,
                        hipStream_t s, bool tile_bits = false);
// Small batches, hidden layers behind the first (small_layer.hip): a workgroup owns 16 columns for all B <= 64 rows, so one
// launch is Linear + BatchNorm1d (batch statistics) + ReLU + Dropout (+ skip) forward, and one launch is dX = dz W (+ skip
// gradient) followed by the BatchNorm backward of the layer below.  Their ReLU & keep bitmap is in the "tile format" of
// small_layer.hip (tile_bits above: bn_small_bwd reading such a layer).  POSELIFT_SMALL_LAYER=0 switches them off (A/B).
bool small_layer_ok(int B, int H, int K);
// ... the first layer (K = in_dim <= 256 inputs: contraction on the vector unit) forward, its weight gradient in the backward
// launch of the layer above, and the whole top of the backward pass (g = dy W2, BatchNorm backward of the last hidden layer,
// dW2, db2; out_dim <= 64) as one launch.  POSELIFT_SMALL_ENDS=0 switches these off (A/B).
bool small_first_ok(int K);
bool small_top_ok(int O);
int launch_small_layer_fwd(const float* a, const float* W, const float* bias, const float* gamma, const float* beta, float eps,
                           float momentum, float* rm, float* rv, int64_t* nbt, float* mean, float* rstd, const float* resid,
                           float* z, float* act, uint64_t* bits, int B, int H, int K, float pdrop, uint64_t seed, uint64_t step,
                           int layer, const uint64_t* inject_keep, hipStream_t s, const uint64_t* step_dev, bool first = false,
                           const float* W2 = nullptr, float* ypart = nullptr, int O = 0,
                           const unsigned short* a_planes = nullptr, unsigned short* out_planes = nullptr);
// a_planes (PL_F16X3 descriptors, not the first layer): the input as two fp16 planes [B][K] (h, l) -- the contraction then
// runs as three fp16 MFMAs per product (fp32-grade, a fifth of the MFMA time of the exact fp32 form); out_planes: the output
// also as planes [B][H], for the next layer's launch
// ypart != NULL (the last hidden layer of a fused train step): the launch also leaves its share of the output Linear,
// ypart [H / 16][B][64] (columns < O); launch_small_mse adds the slabs up: y, dpred = grad_scale * 2 (y - t) / (B O) and
// small_mse_partials(B, O) partial sums of (y - t)^2 -- no launch for the output layer, none for its slab reduce
// evaluation forward (model.eval()): the same layer kernels with the BatchNorm fold on the running statistics as tail, the
// grid also over 64-row blocks (any M; every row the same bits whatever the batch), and launch_small_out for the output layer
int launch_small_layer_eval(const float* a, const float* W, const float* bias, const float* gamma, const float* beta, float eps,
                            const float* rm, const float* rv, const float* resid, float* act, int M, int H, int K, hipStream_t s,
                            bool first = false, const float* W2 = nullptr, float* ypart = nullptr, int O = 0,
                            const unsigned short* a_planes = nullptr, unsigned short* out_planes = nullptr);
int launch_small_out(const float* ypart, int NS, int M, int O, const float* bias, float* y, hipStream_t s);
// the Linear alone with the statistics partials of a tile GEMM's epilogue, for TRAINING batches of 65 ... 512 rows: the tile
// GEMMs have 8 ... 32 tiles for 256 CUs there (22 us per forward GEMM at any of these sizes; this form: 9-17 us)
int launch_small_linear_stats(const float* a, const unsigned short* a_planes, const float* W, const float* bias, float* z, int M,
                              int H, int K, float* stat_sum, float* stat_m2, int groups, hipStream_t s);
int small_mse_partials(int B, int O);     // partial sums launch_small_mse leaves in mpart (<= 64)
int launch_small_mse(const float* ypart, int NS, int B, int O, const float* bias, const float* tgt, float grad_scale, float* y,
                     float* dpred, float* mpart, hipStream_t s);
// what the BatchNorm backward of a layer reads and writes (besides its incoming gradient and dz)
struct SmallBnLayer {
  const float *z, *mean, *rstd, *gamma;
  const uint64_t* bits;
  bool rowbits;                           // bitmap in the row format (the layer's forward was bn_small_fwd_kernel)
  float *dgamma, *dbeta, *dbias;
};
// g = dz W (+ addend) -> gout (or NULL), then the BatchNorm backward of the layer below -> dz_lo;
// dW != NULL: + dW = dz^T a_in (extra workgroups); dW1 != NULL: the layer below is the first one, + dW1 = dz_lo^T x1
int launch_small_layer_bwd(const float* dz, const float* W, const float* addend, float* gout, int B, int H, int K,
                           const SmallBnLayer& below, float kscale, float* dz_lo, hipStream_t s, const float* a_in = nullptr,
                           float* dW = nullptr, const float* x1 = nullptr, float* dW1 = nullptr, int K1 = 0,
                           const struct AdamWRide* adam = nullptr);   // adam: a slice of the AdamW step on spare workgroups (adamw.h)
int launch_small_top_bwd(const float* dy, const float* W2, const float* h, int B, int H, int O, float* gout, float* dW2,
                         float* db2, const SmallBnLayer& top, float kscale, float* dz_top, hipStream_t s,
                         const float* mpart = nullptr, int np = 0, float inv_n = 0.f, float* loss = nullptr,
                         uint64_t* tick = nullptr);   // loss != NULL: + loss = inv_n * sum of the np partials, tick += 1
// eval-mode BatchNorm for the saved-state forward: mean := running mean, rstd := rsqrt(running var + eps), scale, shift
int launch_bn_eval_stats(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int H,
                         float* mean, float* rstd, float* scale, float* shift, hipStream_t s);
// pass 2: dz = c0*(dy - c1 - zhat*c2)  (bn) or dz = dy (no bn); partial column sums of dz
// planes: also (dz == NULL: only) write dz as GEMM operand planes
int launch_bn_bwd_dz(const float* g, const uint64_t* bits, const float* z, const float* mean,
                     const float* rstd, const float* coef, float keep_scale, int bn, int B, int H,
                     float* dz, float* part_db, hipStream_t s, int Hc = 0, const PlaneOut* planes = nullptr, int rc = 0);

// out[i] = sum_s slabs[s*n + i]
int launch_reduce_slabs(const float* slabs, int nslab, int64_t n, float* out, hipStream_t s);
// njobs independent out[c] = sum_r part[r][c] reductions in one launch
// kind 2 (at most one job): out[0] = (sum of the R MSE partials) * loss_inv_n, then loss_tick[0] += 1 -- mse_final_kernel
int launch_reduce_rows_multi(const float* const* part, const int* R, const int* H, float* const* out, int njobs,
                             hipStream_t s, const int* kind = nullptr, const int* transK = nullptr, float loss_inv_n = 0.f,
                             uint64_t* loss_tick = nullptr);
int mse_partial_only(const float* pred, const float* tgt, int64_t n, float grad_scale, float* dpred, void* scratch, void* stream);
int mse_partials(int64_t n);
// the output Linear's slab reduce folded into the MSE partial pass (fused train step; bit-identical to the two launches)
bool mse_from_slabs_supported(int splits, int N);
int mse_partial_from_slabs(const float* part, int splits, int B, int N, const float* bias, const float* tgt, float grad_scale,
                           float* y, float* dpred, void* scratch, void* stream);
// out[r][c] = bias[c] + sum_s slabs[s][r][c]
int launch_reduce_slabs_bias(const float* slabs, int nslab, int rows, int cols, const float* bias,
                             float* out, hipStream_t s);
// partial column sums of X[rows][cols] -> part[RC][cols], RC = colsum_chunks(rows)
int colsum_chunks(int rows);
int launch_colsum_partial(const float* X, int rows, int cols, float* part, hipStream_t s);
// eval-mode fold: scale = gamma*rsqrt(rv+eps), shift = (bias - rm)*scale + beta  (bn) or 1, bias
int launch_bn_fold_eval(const float* bias, const float* gamma, const float* beta, const float* rm,
                        const float* rv, float eps, int bn, int H, float* scale, float* shift,
                        hipStream_t s);
int launch_fill(float* p, int64_t n, float v, hipStream_t s);

// ---------------------------------------------------------------------------------
// skinny layers (skinny.hip): 34 -> H and H -> 51, MFMA straight from registers
// ---------------------------------------------------------------------------------
bool skinny_supported(int K, int H);           // narrow dimension specialised (34, 51), H % 128 == 0
int skinny_chunks(int B);                      // row tasks (64 rows each)
int skinny_stat_groups(int B);                 // 64-row BN statistics groups it emits
// bnr (optional, w_transposed only): its bnr_* fields = BatchNorm-backward pass 1 of the layer that consumes `out` as its
// incoming gradient, done on the block just produced (amax per (64-row group, 32-column strip): (B/64) * (H/32) pairs)
int launch_skinny_wide_out(const float* X, const float* W, const float* bias, float* out, int B, int K,
                           int H, bool w_transposed, float* stat_sum, float* stat_m2, hipStream_t s,
                           const GemmArgs* bnr = nullptr);
// reduce == false: only the partials are written (part: skinny_in_chunks(B) x K x H floats); the caller combines them
// (launch_reduce_rows_multi: R = skinny_in_chunks(B), H = K*H, transK = K when out_transposed)
// xsum (optional): skinny_in_chunks(B) x K partial column sums of X (the output layer's bias gradient: X = dy), to be combined
// like any other partial (R = skinny_in_chunks(B), H = K)
int launch_skinny_wide_in(const float* X, const float* D, float* out, int B, int K, int H,
                          bool out_transposed, float* part, hipStream_t s, bool reduce = true, float* xsum = nullptr);
int skinny_in_chunks(int B);
bool skinny_narrow_out_supported(int H, int N);
size_t skinny_narrow_out_part_floats(int B, int H);
// part: [splits][B][64] slabs; reduce == false: only the slabs are written (y = bias + their sum is left to the caller)
int launch_skinny_narrow_out(const float* h, const float* W, const float* bias, float* y, int B, int H,
                             int N, float* part, hipStream_t s, bool reduce = true);
int skinny_narrow_out_splits(int B, int H);

}  // namespace pl
