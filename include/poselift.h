/* poselift.h -- C ABI of the MI355X (gfx950) 2D->3D pose-lifting train path.
 *
 * The reference (RHnejad/3D_PoseEstimation) has no FFI: its hot path is a plain
 * PyTorch nn.Module plus a script-level train step.  This header is therefore the
 * boundary a maintainer binds with ctypes; every entry point
 * names the reference lines whose computation it replaces.  All paths are relative
 * to the reference root.
 *
 * Conventions
 *   - plain C, POD arguments, no torch types; device pointers are hipMalloc'ed
 *     (or torch-owned) fp32 unless stated; every tensor is dense row-major.
 *   - the caller owns every buffer including the workspace; the library never
 *     allocates or frees device memory and keeps no device-side global state.
 *   - every call enqueues work on `stream` and returns; nothing synchronises.
 *   - return value: PL_OK (0) or a negative PLStatus; pl_last_error() gives the
 *     thread-local message.  Nothing throws across the ABI.
 */
#ifndef POSELIFT_H_
#define POSELIFT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PL_VERSION 109 /* 0.1.9: + pl_conv2d_planes_fwd_hw, pl_conv2d_planes_wgrad_hw (the stem on the planes GEMM); 0.1.8: + pl_lifter_train_step, pl_lifter_step_carries_adamw; 0.1.7: + pl_workspace_bitmap_format (small-batch layer kernels); 0.1.6: + pl_counter_add; 0.1.5: + pl_flip_pose_ex, pl_flip_w_nhwc (phase5 Flip branch); 0.1.4: + pl_planes_split_strided; 0.1.3: + pl_bn_join_bwd (0.1.2: operand-plane outputs of the BatchNorm / join kernels, pl_gemm_planes_raw) */

typedef enum PLStatus {
  PL_OK = 0,
  PL_EINVAL = -1,       /* null pointer / bad enum / misaligned arena            */
  PL_ESHAPE = -2,       /* unsupported shape (hidden % 4 != 0, dims <= 0, ...)   */
  PL_EDTYPE = -3,       /* unsupported compute dtype                             */
  PL_EBATCH = -4,       /* B < 2 with training-mode BatchNorm (torch raises too) */
  PL_EHIP = -5,         /* a HIP runtime call or launch failed                   */
  PL_EWORKSPACE = -6,   /* workspace pointer null or too small                   */
  PL_ESYNC = -7         /* the PLSync gather callback reported a failure         */
} PLStatus;

/* arithmetic (and operand storage) of the 1024-wide GEMMs:
 *   PL_F32     v_mfma_f32_32x32x2_f32, exact fp32 products                  (meets the 1e-3 mm gate)
 *   PL_BF16    operands rounded to bf16, v_mfma_f32_32x32x16_bf16           (~1 mm MPJPE).  On whole 128-tiles with
 *              BatchNorm the operands are STORED as bf16 by their producers (activations, dz, a bf16 weight shadow)
 *              and staged by LDS-DMA; z, the gradient flow and every reduction stay fp32.  Other shapes keep fp32
 *              storage and round while staging: the same values, the same products.
 *   PL_BF16X6  each fp32 operand split into 3 bf16 pieces, 6 bf16 MFMAs per product term:
 *              fp32-grade results (meets the gate too) at 6/16 of the fp32 matrix time
 *   PL_F16X3   fp32-grade on HALF the matrix work of PL_BF16X6: every operand tensor is written by its producing
 *              kernel as two fp16 planes, S x = h + l / 2048 (h = fp16(S x), l = fp16((S x - h) 2048), S a per-tensor
 *              power of two: 22-23 significant bits), and a b = h_a h_b + (h_a l_b + l_a h_b) / 2048 is accumulated on
 *              two fp32 accumulators by three fp16 MFMAs; the GEMM stages the planes with LDS-DMA and does no split
 *              work.  Whole 128-tiles with BatchNorm and local statistics; other shapes run PL_BF16X6 arithmetic. */
typedef enum PLDtype { PL_F32 = 0, PL_BF16 = 1, PL_BF16X6 = 2, PL_F16X3 = 3 } PLDtype;

/* Cross-rank BatchNorm statistics ("SyncBN"; data-parallel extension, SURVEY 8e -- the reference is
 * single-process and has no counterpart).  With PLDesc.sync set and world > 1, training-mode BatchNorm
 * normalises with the statistics of the GLOBAL batch (world x B rows; every rank must pass the same B):
 * after a layer's partial statistics are computed the library calls
 *     gather(user, buf, floats_per_rank, stream)
 * where buf = [world][floats_per_rank] device floats inside the caller's workspace and slab `rank` is
 * already filled by work enqueued on `stream`.  The callee must enqueue an all-gather that fills the
 * other slabs, ordered after that work and before anything enqueued on `stream` later (RCCL:
 * ncclAllGather in place on that stream), and return 0 (non-zero -> PL_ESYNC).  One gather per hidden
 * layer in the training forward (2 x ceil(B/64)-ish x hidden floats) and one in backward.  The merged
 * result does not depend on how the global batch is cut into ranks: forward activations and running
 * statistics are bit-identical to one process running the concatenated batch. */
typedef int (*PLGatherFn)(void* user, float* buf, int64_t floats_per_rank, void* stream);
typedef struct PLSync {
  int32_t world;       /* number of ranks (1 = local statistics)              */
  int32_t rank;        /* this process, 0 <= rank < world                     */
  PLGatherFn gather;
  void* user;          /* passed back to gather                               */
} PLSync;

/* Lifter descriptor: LinearModel(i_dim, o_dim, linear_size, num_stage, p_dropout, BN)
 * phase1_lifting/baselineModel.py:50-85.
 *
 * params     flat fp32 parameter arena.  Tensors appear in the reference's
 *            parameters() order -- per hidden layer: Linear.weight [out][in],
 *            Linear.bias, BatchNorm.weight, BatchNorm.bias (BN tensors are present
 *            even when bn == 0, as in the reference) -- then w2.weight, w2.bias.
 *            Every tensor starts at a multiple of 64 floats: pl_param_offset().
 * bn_running [n_hidden][2][hidden] fp32: running_mean then running_var per layer.
 * bn_batches [n_hidden] int64 num_batches_tracked, or NULL.
 */
typedef struct PLDesc {
  int32_t in_dim;      /* 34 = 17 joints x 2                                  */
  int32_t hidden;      /* linear_size, multiple of 4                          */
  int32_t out_dim;     /* 51 = 17 joints x 3                                  */
  int32_t num_stage;   /* residual blocks; hidden layers = 1 + 2*num_stage    */
  int32_t bn;          /* BN flag of the reference constructor                */
  int32_t dtype;       /* PLDtype: arithmetic of the 1024-wide GEMMs          */
  float p_dropout;     /* nn.Dropout p                                        */
  float bn_eps;        /* 1e-5 (nn.BatchNorm1d default)                       */
  float bn_momentum;   /* 0.1                                                 */
  int32_t reserved;
  float* params;
  float* bn_running;
  int64_t* bn_batches;
  const PLSync* sync;  /* NULL = BatchNorm over the local batch (the reference's behaviour) */
  /* Graph replay (hipGraph / torch.cuda.graph capture of the train step; no reference counterpart): a device counter of
   * completed steps.  When set, pl_lifter_fwd_train's dropout stream uses step + *step_dev (the kernels read it), and
   * pl_lifter_train_fwd_bwd increments it once per call, after the forward's last reader -- so a captured step draws fresh
   * dropout masks on every replay.  NULL: the `step` argument alone (the eager behaviour). */
  const uint64_t* step_dev;
  /* Operand planes of the 1024-wide weights kept ACROSS calls (optional; caller-owned, pl_wplanes_bytes() bytes).
   * NULL: every forward call splits the weights into its workspace first.  Set: the planes live here; with
   * wplanes_valid != 0 the caller asserts they hold the current parameters (pl_adamw_flat_planes wrote them and nothing
   * changed the parameters since) and the forward reads them as they are; with 0 the forward refreshes them first. */
  void* wplanes;
  int32_t wplanes_valid;
  int32_t reserved2;
} PLDesc;

int pl_version(void);
const char* pl_last_error(void);

/* ---- arena layout (host only, no device work) ---------------------------------- */
int64_t pl_num_hidden(const PLDesc* d);               /* 1 + 2*num_stage            */
int64_t pl_param_tensors(const PLDesc* d);            /* 4*n_hidden + 2             */
int64_t pl_param_offset(const PLDesc* d, int64_t i);  /* floats, multiple of 64     */
int64_t pl_param_numel(const PLDesc* d, int64_t i);
int64_t pl_param_arena_floats(const PLDesc* d);       /* padded arena length        */
/* Workspace for B rows: saved activations, masks, BN statistics, gradient scratch. */
size_t pl_workspace_bytes(const PLDesc* d, int64_t B);
/* Debug/test view into the workspace.  which: 0 z (pre-BN GEMM output), 1 act (layer
 * output incl. residual), 2 keep&relu bitmap, 3 batch mean, 4 batch rstd.           */
int pl_workspace_view(const PLDesc* d, int64_t B, int which, int64_t layer,
                      size_t* offset_bytes, size_t* size_bytes);
/* Layout of the bitmap of hidden layer `layer` after a training forward of B rows (which = 2 above):
 *   0  row format: [B][4*ceil(H/256)] words; row r, 256-column strip q, word j: bit l <-> column 256 q + 4 l + j;
 *   1  tile format (batches of <= 64 rows whose hidden layers run as ONE launch each, csrc/small_layer.hip): H words;
 *      element (r, c) is bit (r & 15) * 4 + ((c >> 2) & 3) of word (c >> 4) * 16 + (r >> 4) * 4 + (c & 3).
 * Negative = error code.  (Debug / test only: the backward pass knows by itself.) */
int pl_workspace_bitmap_format(const PLDesc* d, int64_t B, int64_t layer);

/* ---- forward ------------------------------------------------------------------- */
/* model.eval(); model(x)   phase1_lifting/baselineModel.py:87-102 under
 * train_1.py:112-126 (BatchNorm on running statistics, Dropout = identity).
 * x [B][in_dim], y [B][out_dim]. */
int pl_lifter_fwd_eval(const PLDesc* d, const float* x, float* y, int64_t B,
                       void* workspace, size_t workspace_bytes, void* stream);

/* model.train(); model(x)  phase1_lifting/baselineModel.py:87-102 under train_1.py:86.
 * Updates bn_running / bn_batches like nn.BatchNorm1d (momentum, unbiased running
 * var).  Dropout keep decisions come from the Philox4x32-10 stream keyed by
 * (seed, step, layer, element) documented in csrc/philox.h, or -- parity mode --
 * from inject_keep: n_hidden bitmaps laid out like workspace view 2 (NULL = Philox).
 * Saves what pl_lifter_bwd needs in the workspace. */
int pl_lifter_fwd_train(const PLDesc* d, const float* x, float* y, int64_t B,
                        void* workspace, size_t workspace_bytes,
                        uint64_t seed, uint64_t step, const uint64_t* inject_keep,
                        void* stream);

/* loss.backward() through the module (autograd of baselineModel.py:87-102).
 * dy [B][out_dim]; flat_grads: arena shaped like d->params, OVERWRITTEN with the
 * gradient of every parameter; dx [B][in_dim] or NULL (phase5_loop needs it,
 * train_1.py does not). Must follow pl_lifter_fwd_train on the same workspace. */
int pl_lifter_bwd(const PLDesc* d, const float* x, const float* dy, int64_t B,
                  void* workspace, size_t workspace_bytes, float* dx,
                  float* flat_grads, void* stream);

/* model.eval() with gradients flowing through it (phase5_loop/train_5.py:120 runs the lifter in eval mode inside the cycle
 * graph): the eval forward computed so that its backward can follow -- BatchNorm on the running statistics (left
 * untouched), Dropout = identity, pre-activations and ReLU bitmaps saved in the workspace -- and that backward: dx and
 * the gradient of every parameter (dgamma, dbeta included, as torch computes them in eval mode).  Same numbers as
 * pl_lifter_fwd_eval to fp32 round-off; pl_lifter_fwd_eval stays the fast path when nothing needs a gradient. */
int pl_lifter_fwd_eval_saved(const PLDesc* d, const float* x, float* y, int64_t B,
                             void* workspace, size_t workspace_bytes, void* stream);
int pl_lifter_bwd_eval(const PLDesc* d, const float* x, const float* dy, int64_t B,
                       void* workspace, size_t workspace_bytes, float* dx, float* flat_grads, void* stream);

/* The same backward restricted to a range of layers (data-parallel overlap, no reference
 * counterpart).  Layers are numbered along the chain: hidden Linears 0 .. L-1 (L = pl_num_hidden)
 * and the output Linear = L.  The call runs layers hi, hi-1, .., lo (0 <= lo <= hi <= L); the
 * gradient flowing between two calls lives in the workspace, so consecutive ranges
 * (L..a), (a-1..b), .., (c-1..0) == pl_lifter_bwd, bit for bit.  After the range ending at lo the
 * gradients of every tensor from pl_param_offset(d, 4*lo) to the end of the arena are final:
 * their all-reduce can run while the lower layers compute.  dx is written by the range with lo = 0. */
int pl_lifter_bwd_layers(const PLDesc* d, const float* x, const float* dy, int64_t B,
                         void* workspace, size_t workspace_bytes, float* dx,
                         float* flat_grads, int hi, int lo, void* stream);

/* One train_1.py:75-95 iteration body up to the optimiser in ONE call: model.train() forward
 * (baselineModel.py:87-102), MSELoss(mean) against target [B][out_dim] (train_1.py:94) and
 * loss.backward() (:95).  y [B][out_dim] and loss (1 float) are outputs; flat_grads is overwritten.
 * hi, lo as in pl_lifter_bwd_layers: the call with hi = L runs forward + loss first, then
 * backward over layers hi..lo; (L, 0) is the whole step.  Identical results to the separate calls. */
int pl_lifter_train_fwd_bwd(const PLDesc* d, const float* x, const float* target, int64_t B,
                            void* workspace, size_t workspace_bytes, uint64_t seed, uint64_t step,
                            float* y, float* loss, float* flat_grads, int hi, int lo, void* stream);

/* The WHOLE train_1.py:75-100 iteration body in one call: pl_lifter_train_fwd_bwd(.., L, 0, ..) and optimizer.step()
 * (train_1.py:39,89: torch.optim.AdamW) on d->params IN PLACE -- m, v: the exp_avg / exp_avg_sq arenas laid out like the
 * parameters; t the step number (>= 1); lr_dev / t_dev != NULL (graph replay): lr = *lr_dev and t = t + *t_dev, read on
 * the device.  At batches of <= 64 rows (the reference's own 64) whose hidden layers run as one launch each
 * (pl_lifter_step_carries_adamw = 1) the update has no launch of its own: each backward launch carries, on spare
 * workgroups, the slice of the arena whose gradients the launches before it finished and which nothing reads any more, and
 * one small launch updates the bottom of the arena (first layer, first residual Linear).  Otherwise: one pl_adamw_flat(_dev)
 * launch behind the backward pass.  Same element arithmetic as pl_adamw_flat either way.  The caller's persistent weight
 * planes (PLDesc.wplanes) are stale afterwards. */
typedef struct PLAdamWStep {
  float* m;
  float* v;
  float lr;
  const float* lr_dev;
  float beta1, beta2, eps, weight_decay;
  int64_t t;
  const uint64_t* t_dev;
} PLAdamWStep;
int pl_lifter_step_carries_adamw(const PLDesc* d, int64_t B);
int pl_lifter_train_step(const PLDesc* d, const float* x, const float* target, int64_t B, void* workspace,
                         size_t workspace_bytes, uint64_t seed, uint64_t step, float* y, float* loss, float* flat_grads,
                         const PLAdamWStep* opt, void* stream);

/* The 3-D head forward on NHWC logits [B][H][W][J*64] (depth_dim 64, the conv path's layout): same
 * coords [B*J][3] and stats [B*J][5] as pl_softargmax_fwd(ncoord 3, centred 1).  Model.py:94-133. */
int pl_softargmax3d_nhwc_fwd(const float* logits, int64_t B, int64_t J, int64_t H, int64_t W,
                             float* coords, float* stats, void* stream);
/* Backward of pl_softargmax3d_nhwc_fwd: dlogits [B][H*W][J*64] from the saved logits, the forward's stats [B*J][5]
 * and the coordinate gradients gcoords [B*J][3]; one streaming pass (4 B read + 4 B written per voxel), so a
 * training step needs no NHWC <-> NCHW pass around the head (phase4_joined/Model.py:94-133 under autograd). */
int pl_softargmax3d_nhwc_bwd(const float* logits, const float* stats, const float* gcoords, int64_t B, int64_t J,
                             int64_t H, int64_t W, float* dlogits, void* stream);

/* ---- convolution path (first slice: forward) ------------------------ */
/* nn.Conv2d forward in NHWC with the Bottleneck's eval-mode epilogue folded in: phase4_joined/Resnet.py:51-95
 * (conv1/2/3 + bn + relu + residual), :112-118 (7x7 stem), :151-158 (downsample), phase4_joined/Model.py:66-69
 * (final 1x1 conv with bias).
 *   v = sum_{kh,kw,ci} x[b][oh*stride - pad + kh][ow*stride - pad + kw][ci] * w[co][kh][kw][ci]  (+ bias[co])
 *   v = v * scale[co] + shift[co]          (BatchNorm2d on running statistics, folded; or NULL)
 *   relu 1: v = max(v, 0), then v += resid;   relu 2: v += resid, then max(v, 0);   relu 0: v += resid
 * x [B][H][W][Cin], w [Cout][KH][KW][Cin] (OHWI), resid / y [B][Ho][Wo][Cout].
 * Cin % 32 == 0: implicit GEMM on the PL_BF16X6 pipeline (fp32-grade products, no im2col buffer; ragged last
 * row / column tiles are clamped and masked); 1x1 stride 1: a plain GEMM; anything else (the stem with Cin = 3):
 * explicit im2col into `scratch` (>= pl_conv2d_nhwc_scratch_bytes, 0 for the others).
 * arith: PL_BF16X6 (fp32-grade) or PL_BF16 (operands rounded to bf16 while staged, one MFMA product: the
 * throughput mode; storage stays fp32). */
size_t pl_conv2d_nhwc_scratch_bytes(int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int KH,
                                    int KW, int stride, int pad);
/* the same knowing the epilogue of the call (the 7x7 / Cin 3 stem needs no scratch unless it carries a bias, a residual
 * or relu == 2; the plain query answers for that worst case) */
size_t pl_conv2d_nhwc_scratch_bytes_ex(int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int KH,
                                       int KW, int stride, int pad, int has_bias, int has_resid, int relu);
int pl_conv2d_nhwc_fwd(const float* x, int64_t B, int64_t H, int64_t W, int64_t Cin, const float* w,
                       int64_t Cout, int KH, int KW, int stride, int pad, const float* scale,
                       const float* shift, const float* bias, int relu, const float* resid, float* y,
                       int arith, void* scratch, size_t scratch_bytes, void* stream);

/* Weight gradient of the same convolution (autograd of nn.Conv2d, phase4_joined/train.py:80 loss.backward()):
 * dw [Cout][KH][KW][Cin] = sum_{b,oh,ow} dy[b][oh][ow][co] * x[b][oh*stride - pad + kh][ow*stride - pad + kw][ci],
 * the TN GEMM over pixels with x gathered on the fly, split over the pixels so that it fills the chip (fixed-order
 * slab combine, no float atomics).  Cout and Cin even, Wo % 8 == 0 and B*Ho*Wo % 32 == 0 take the gathered kernel
 * (ragged tiles clamped and masked); other shapes an explicit im2col + the generic GEMM in scratch; the 7x7 / Cin = 3
 * stem its own exact-fp32 kernel.  arith: PL_BF16X6 (fp32-grade) or PL_BF16 (operands rounded to bf16 while staged,
 * fp32 accumulate and output).  The input gradient needs no kernel of its own: stride 1 = pl_conv2d_nhwc_fwd on dy with
 * the flipped, transposed filter; stride 2 = pl_deconv4x4s2_nhwc_fwd (conv.py conv2d_nhwc_dgrad). */
size_t pl_conv2d_nhwc_wgrad_scratch_bytes(int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                          int KH, int KW, int stride, int pad);
int pl_conv2d_nhwc_wgrad(const float* x, int64_t B, int64_t H, int64_t W, int64_t Cin, const float* dy,
                         int64_t Cout, int KH, int KW, int stride, int pad, float* dw, int arith, void* scratch,
                         size_t scratch_bytes, void* stream);

/* Training-mode BatchNorm (+ ReLU) over the rows of a [rows][C] fp32 matrix: nn.BatchNorm2d on an NHWC feature map
 * ([B*H*W][C]; phase4_joined/Resnet.py:56-63, Model.py:52-58) -- or BatchNorm1d -- with batch statistics, running
 * statistics updated as torch does (momentum, unbiased variance, num_batches += 1).  C % 4 == 0, rows >= 2.
 *   fwd: y = [relu](gamma * (z - mean) * rstd + beta); bits [rows][4*ceil(C/256)] 64-bit words (the layout of
 *        pl_workspace_view which = 2): 1 where the output passed the ReLU (all ones without it); mean, rstd [C] saved.
 *   bwd: dz, dgamma [C], dbeta [C] from dy and what fwd saved.
 * scratch >= pl_bn_train_scratch_bytes(rows, C) for both. */
size_t pl_bn_train_scratch_bytes(int64_t rows, int64_t C);
int pl_bn_train_fwd(const float* z, int64_t rows, int64_t C, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, int64_t* batches, int relu,
                    float* y, uint64_t* bits, float* mean, float* rstd, void* scratch, void* stream);
int pl_bn_train_bwd(const float* dy, const uint64_t* bits, const float* z, const float* mean, const float* rstd,
                    const float* gamma, int64_t rows, int64_t C, float* dz, float* dgamma, float* dbeta,
                    void* scratch, void* stream);
/* The residual join of a Bottleneck in training mode (Resnet.py:90-91): out = relu(a + b) and its bitmap;
 * backward: both inputs receive pl_mask_by_bits(g). */
int pl_add_relu_fwd(const float* a, const float* b, int64_t rows, int64_t C, float* out, uint64_t* bits, void* stream);
int pl_mask_by_bits(const float* g, const uint64_t* bits, int64_t rows, int64_t C, float* dx, void* stream);

/* The same building blocks writing GEMM OPERAND PLANES (the conv path's 1x1 convolutions on the planes GEMM, as the
 * lifter's 1024-wide Linears: the kernel that produces a tensor also writes it as 16-bit planes, the GEMM stages them by
 * LDS-DMA).  planes of an n-element tensor: planes_mode PL_F16X3 -> [2][n] fp16 (h = fp16(S x), l = fp16((S x - h) 2048)),
 * PL_BF16 -> [n] bf16; 16-byte aligned, n % 8 == 0; NULL = none.  Replaces the same reference code as the plain forms.
 *   pl_planes_split     : planes of an fp32 tensor with the static scale S (weights: 16; conv-path activations:
 *                         pl_conv_act_plane_scale() = 1/64 -- eval-mode maps of an unnormalised network reach 1e5)
 *   pl_bn_train_fwd_ex  : y may be NULL when only the planes are wanted (S = pl_conv_act_plane_scale()); gemm_stat (optional): the batch statistics
 *                         as the producing GEMM's epilogue left them -- [2][pl_gemm_stat_groups(rows)][C]: per 64-row group
 *                         the column sums, then the sums of squares about the group mean -- instead of a pass over z
 *                         join (optional, relu != 0): y = relu(bn(z) + join) in the same pass -- the Bottleneck's bn3 and
 *                         residual join (Resnet.py:81-91); bits then is the join's ReLU bitmap
 *   pl_bn_train_bwd_ex  : dz may be NULL; PL_F16X3 planes hold S dz with S a power of two chosen on the device from a range
 *                         bound of dz; dz_scale (device, 2 floats) receives {S, 1/S} -- pass dz_scale + 1 as dyn_inv below
 *   pl_add_relu_fwd_ex  : out (fp32, the next join reads it) AND its planes (S = pl_conv_act_plane_scale())
 *   pl_mask_add_by_bits : dx = (g + g2) masked (g2 may be NULL): the join's backward with the gradient sum folded in */
float pl_conv_act_plane_scale(void);   /* S of every activation-plane output of the conv-path entries below (1/64) */
int pl_planes_split(const float* x, int64_t n, int planes_mode, float scale, void* planes, void* stream);
/* The same from a strided 4-D view of the source (a convolution weight in the layout its GEMM wants -- OIHW -> OHWI, the
 * transpose, the spatial flip of conv.py's data gradients -- straight from the nn.Conv2d parameter, one launch): element
 * (i0,i1,i2,i3) of the contiguous result is x[i0 s0 + i1 s1 + i2 s2 + i3 s3]; x points at the view's first element, strides are
 * in elements and signed (a flipped axis has a negative one); dims[3] % 4 == 0. */
int pl_planes_split_strided(const float* x, const int64_t* dims, const int64_t* strides, int planes_mode, float scale,
                            void* planes, void* stream);
int pl_bn_train_fwd_ex(const float* z, int64_t rows, int64_t C, const float* gamma, const float* beta, float eps,
                       float momentum, float* running_mean, float* running_var, int64_t* batches, int relu,
                       float* y, uint64_t* bits, float* mean, float* rstd, void* scratch, void* y_planes,
                       int planes_mode, const float* gemm_stat, const float* join, void* stream);
int pl_bn_train_bwd_ex(const float* dy, const uint64_t* bits, const float* z, const float* mean, const float* rstd,
                       const float* gamma, int64_t rows, int64_t C, float* dz, float* dgamma, float* dbeta,
                       void* scratch, void* dz_planes, int planes_mode, float* dz_scale, void* stream);
int pl_add_relu_fwd_ex(const float* a, const float* b, int64_t rows, int64_t C, float* out, uint64_t* bits,
                       void* out_planes, int planes_mode, void* stream);
int pl_mask_add_by_bits(const float* g, const float* g2, const uint64_t* bits, int64_t rows, int64_t C, float* dx,
                        void* stream);
/* Backward of bn3 + residual join in one go (Resnet.py:81-91 under autograd): dx = (g + g2) where the join's bitmap is set
 * (g2 may be NULL) -- the identity's gradient and bn3's dy -- written by the pass that also takes BatchNorm-backward's column
 * sums; then the finalize and dz as pl_bn_train_bwd_ex (same scratch, same dz / dz_planes / dz_scale meaning).  C >= 256. */
int pl_bn_join_bwd(const float* g, const float* g2, const uint64_t* bits, const float* z, const float* mean,
                   const float* rstd, const float* gamma, int64_t rows, int64_t C, float* dx, float* dz, float* dgamma,
                   float* dbeta, void* scratch, void* dz_planes, int planes_mode, float* dz_scale, void* stream);

/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1)  phase4_joined/Resnet.py:119.  x [B][H][W][C], C % 4 == 0;
 * y [B][(H-1)/2+1][(W-1)/2+1][C]. */
int pl_maxpool3x3s2_nhwc(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* y, void* stream);
/* its backward: dx [B][H][W][C] from the forward input x and dy [B][Ho][Wo][C]; a window's gradient goes to its first
 * maximum in (kh, kw) order, recomputed from x (no index tensor, no atomics). */
int pl_maxpool3x3s2_nhwc_bwd(const float* x, const float* dy, int64_t B, int64_t H, int64_t W, int64_t C,
                             float* dx, void* stream);
/* The training-mode pair (what autograd's MaxPool2d keeps is an index tensor too): the forward also writes, one
 * byte per output element, the tap kh*3 + kw of the first maximum; the backward reads those bytes and dy only. */
int pl_maxpool3x3s2_nhwc_idx(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* y,
                             unsigned char* idx, void* stream);
int pl_maxpool3x3s2_nhwc_bwd_idx(const unsigned char* idx, const float* dy, int64_t B, int64_t H, int64_t W,
                                 int64_t C, float* dx, void* stream);
/* y [B][2Hi][2Wi][C]: x on the even pixels, zero elsewhere -- the input gradient of a 1x1 stride-2 convolution
 * (the downsample branches, Resnet.py:151-158) is dy W (a GEMM) placed this way. */
int pl_upsample2x_zero_nhwc(const float* x, int64_t B, int64_t Hi, int64_t Wi, int64_t C, float* y, void* stream);
/* out[c] = sum_r X[r][c] (fixed order): the bias gradient of the final 1x1 convolution (Model.py:66-69). */
size_t pl_colsum_scratch_bytes(int64_t rows, int64_t cols);
int pl_colsum(const float* X, int64_t rows, int64_t cols, float* out, void* scratch, void* stream);

/* nn.ConvTranspose2d(kernel_size=4, stride=2, padding=1, bias=False) + folded BatchNorm2d + ReLU
 * phase4_joined/Model.py:47-63: x [B][Hi][Wi][Cin] -> y [B][2Hi][2Wi][Cout].  Each output parity
 * (oh&1, ow&1) is a 2x2-tap convolution over the input: four implicit GEMMs at the INPUT resolution (no
 * zero insertion, no wasted MACs) and one interleave pass.  w_sub [4][Cout][2][2][Cin]: parity
 * (ph, pw) = index ph*2 + pw, tap (th, tw) = weight[ci][co][kh][kw] with kh = (ph ? 2 : 3) - 2*th, kw
 * likewise.  Cin % 32 == 0.  scratch >= ..._scratch_bytes. */
size_t pl_deconv4x4s2_nhwc_scratch_bytes(int64_t B, int64_t Hi, int64_t Wi, int64_t Cout);
int pl_deconv4x4s2_nhwc_fwd(const float* x, int64_t B, int64_t Hi, int64_t Wi, int64_t Cin,
                            const float* w_sub, int64_t Cout, const float* scale, const float* shift,
                            int relu, float* y, int arith, void* scratch, size_t scratch_bytes, void* stream);

/* [B][P][C] -> [B][C][P]: the head's NHWC logits to the [B][J*D][H*W] layout of pl_softargmax_fwd. */
int pl_nhwc_to_nchw(const float* in, int64_t B, int64_t P, int64_t C, float* out, void* stream);

/* ---- batch feed ------------------------------------------------------------------- */
/* One batch of the DataLoader path (train_1.py:26-31,75-81: shuffle, collate, .float(), .to(device))
 * from tables resident in HBM: oa[i] = a[idx[i]] (wa floats per row, the (17,2) keypoints),
 * ob[i] = b[idx[i]] (wb floats, the (17,3) targets), i < n.  idx: device int64, every value in
 * [0, table_rows) (the caller's permutation; not checked on the device). */
int pl_gather_rows2(const float* a, int64_t wa, const float* b, int64_t wb, const int64_t* idx,
                    int64_t n, int64_t table_rows, float* oa, float* ob, void* stream);

/* ---- loss / metric / optimiser -------------------------------------------------- */
/* torch.nn.MSELoss(reduction="mean") + its backward  train_1.py:37,94-95.
 * n elements; dpred = grad_scale * 2 (pred - tgt) / n; loss_out: 1 device float.
 * scratch: >= pl_mse_scratch_bytes(n). dpred may be NULL (validation). */
size_t pl_mse_scratch_bytes(int64_t n);
int pl_mse_fwd_bwd(const float* pred, const float* tgt, int64_t n, float grad_scale,
                   float* dpred, float* loss_out, void* scratch, void* stream);

/* torch.nn.L1Loss(reduction="mean") terms of TriangleLoss (phase5_loop/losses.py:10-62; the LinearModel-era
 * copy phase5_loop/train_5 copy.py:34-86), ALL terms of one call in one launch pair:
 * losses[t] = mean |a_t - b_t| over n_t elements; gradients, where wanted, da_t = grad_scale *
 * sign(a_t - b_t) / n_t and db_t = -da_t (sign(0) = 0, as torch).  terms: HOST array of nterms <=
 * PL_L1_MAX_TERMS descriptors holding device pointers; losses: nterms device floats;
 * scratch >= pl_l1_scratch_bytes(nterms). */
#define PL_L1_MAX_TERMS 8
typedef struct PLL1Term {
  const float* a;
  const float* b;
  int64_t n;
  float* da; /* or NULL */
  float* db; /* or NULL */
} PLL1Term;
size_t pl_l1_scratch_bytes(int nterms);
int pl_l1_terms_fwd_bwd(const PLL1Term* terms, int nterms, float grad_scale, float* losses,
                        void* scratch, void* stream);

/* loss_MPJPE  train_1.py:19-23,100: metric[j] += sum_b ||pred[b][j] - tgt[b][j]||_2.
 * pred/tgt [B][joints][3]; metric [joints] device fp32, accumulated in place. */
size_t pl_mpjpe_scratch_bytes(int64_t B, int64_t joints);
int pl_mpjpe_accum(const float* pred, const float* tgt, int64_t B, int64_t joints,
                   float* metric, void* scratch, void* stream);

/* *counter += delta on the stream (one thread).  The optimizer of a step replayed from a hipGraph keeps its step count on
 * the device: pl_adamw_flat_dev reads t = t_base + *t_dev, this call ticks it behind the update (arena.FlatAdam for the
 * conv models: torch.optim.Adam(model.parameters(), lr) of phase4_joined/train.py:39, train_5 copy.py:105-109). */
int pl_counter_add(uint64_t* counter, int64_t delta, void* stream);

/* flip_pose  phase3_direct/my_HybrIK/utils.py:372-396 (used by train_1.py:89-92,128-134 when Flip):
 * out = horizontal flip of in, both [B][17][D], D = 2 (x -> 1-x) or 3 (x -> -x), left/right joints
 * [4,5,6,11,12,13] <-> [1,2,3,14,15,16] swapped.  Out of place. */
int pl_flip_pose(const float* in, float* out, int64_t B, int64_t joints, int64_t D, void* stream);

/* The training-mode Flip branch of the phase5 cycle step, phase5_loop/train_5 copy.py:174-199:
 *   y = (flip_pose(a) + b) / 2      -> pl_flip_pose_ex(a, b, y, ..., x_offset = (D == 2 ? 1 : 0), scale = 0.5)
 *   its backward da = flip'(g) / 2  -> pl_flip_pose_ex(g, NULL, da, ..., x_offset = 0, scale = 0.5)
 * out[b][j][d] = ((d == 0 ? x_offset - in[b][src(j)][0] : in[b][src(j)][d]) + (addend ? addend[b][j][d] : 0)) * scale,
 * src = the left/right joint swap of pl_flip_pose.  addend may be NULL.  Out of place. */
int pl_flip_pose_ex(const float* in, const float* addend, float* out, int64_t B, int64_t joints, int64_t D,
                    float x_offset, float scale, void* stream);

/* torch.flip(frame, (3,)) of train_5 copy.py:176 (NCHW dim 3 = width) on NHWC frames: out[b][h][w][c] = in[b][h][W-1-w][c]. */
int pl_flip_w_nhwc(const float* in, float* out, int64_t B, int64_t H, int64_t W, int64_t C, void* stream);

/* Flip test-time augmentation, (flip_pose(model(flip_pose(x))) + model(x)) / 2 -- the intent of
 * train_1.py:128-134 and phase5_loop/train_5 copy.py:160-171 -- around ONE eval forward of 2B rows:
 *   pack : xx [2B][17][D]: rows [0,B) = x, rows [B,2B) = flip_pose(x)
 *   merge: y [B][17][D] = (yy[0:B) + flip_pose(yy[B:2B))) / 2 */
int pl_flip_tta_pack(const float* x, float* xx, int64_t B, int64_t joints, int64_t D, void* stream);
int pl_flip_tta_merge(const float* yy, float* y, int64_t B, int64_t joints, int64_t D, void* stream);

/* torch.optim.AdamW.step  train_1.py:39,96 over one flat arena (p, g, m, v of n floats).
 * t = 1-based step count; g is multiplied by grad_scale first (1/world_size after a
 * sum all-reduce). */
int pl_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay,
                  int64_t t, float grad_scale, void* stream);

/* The same step with its two per-step inputs read from device memory, so that a captured graph advances by itself:
 * t = t_base + *t_dev (t_dev: a step counter such as PLDesc.step_dev, incremented elsewhere in the graph) and
 * lr = *lr_dev (the host-side LR scheduler writes it).  Bias corrections are derived on the device from t by the
 * code pl_adamw_flat runs, so a replayed step equals the eager one bit for bit. */
int pl_adamw_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev,
                      float beta1, float beta2, float eps, float weight_decay, int64_t t_base,
                      const uint64_t* t_dev, float grad_scale, void* stream);

/* The step that ALSO refreshes the GEMM operand planes of the 1024-wide weight matrices (PL_F16X3: two fp16 planes,
 * PL_BF16: the bf16 shadow) while the updated parameters are in registers: the forward of the next step then needs
 * no weight split of its own (PLDesc.wplanes / wplanes_valid).  seg[q]: floats [offset, offset + numel) of the arena
 * (both multiples of 4) -> plane h (and l) of numel 16-bit elements each.  lr_dev / t_dev as in pl_adamw_flat_dev, or
 * both NULL (lr, t by value). */
#define PL_ADAMW_MAX_SEGS 8
typedef struct PLAdamWSeg {
  int64_t offset, numel;
  void* h;
  void* l;              /* kind 2 only */
} PLAdamWSeg;
typedef struct PLAdamWPlanes {
  int32_t nseg;
  int32_t kind;         /* 1 bf16, 2 fp16 pair */
  float scale;          /* the weight planes' power-of-two scale: pl_weight_plane_scale() */
  int32_t reserved;
  PLAdamWSeg seg[PL_ADAMW_MAX_SEGS];
} PLAdamWPlanes;
int pl_adamw_flat_planes(float* p, const float* g, float* m, float* v, int64_t n, float lr, const float* lr_dev,
                         float beta1, float beta2, float eps, float weight_decay, int64_t t, const uint64_t* t_dev,
                         float grad_scale, const PLAdamWPlanes* planes, void* stream);
/* Layout of the caller-owned weight-plane buffer PLDesc.wplanes: pl_wplanes_bytes() bytes (0: this descriptor has no
 * planes path); the planes of hidden layer l (1 <= l < n_hidden) start at byte (l - 1) * pl_wplanes_layer_bytes(d):
 * plane h, then (PL_F16X3) plane l, hidden*hidden 16-bit elements each, row-major like the weight. */
size_t pl_wplanes_bytes(const PLDesc* d);
size_t pl_wplanes_layer_bytes(const PLDesc* d);
float pl_weight_plane_scale(void);
int pl_wplanes_refresh(const PLDesc* d, void* stream);   /* planes <- current parameters (wplanes_valid is ignored) */

/* ---- building blocks exported for tests ------------------------------------------ */
/* C[M][N] = op(A) op(B) on the fp32 MFMA path.
 * layout 0 (NT): A [M][K], B [N][K]   (forward  z = a W^T)
 * layout 1 (NN): A [M][K], B [K][N]   (backward da = dz W)
 * layout 2 (TN): A [K][M], B [K][N]   (backward dW = dz^T a)
 * bias [N] or NULL is added to every row. split_k > 1 is allowed for layout 2 only and
 * needs slab scratch of split_k*M*N floats (NULL otherwise). */
int pl_gemm_f32(int layout, const float* A, const float* B, float* C, int64_t M,
                int64_t N, int64_t K, const float* bias, int split_k, float* slabs,
                void* stream);
/* the same with the arithmetic chosen (PLDtype); bf16 modes apply to whole 128x128x32 tiles only */
int pl_gemm_arith(int layout, int arith, const float* A, const float* B, float* C, int64_t M,
                  int64_t N, int64_t K, const float* bias, int split_k, float* slabs,
                  void* stream);

/* The same product computed the way the PL_F16X3 / PL_BF16-storage lifter computes its 1024-wide Linears: A and B are
 * split into 16-bit operand planes in `scratch` (the lifter's own producers -- BatchNorm apply / backward, the weight
 * split -- write planes directly), then one planes GEMM (LDS-DMA staging, no split work in the loop).  mode PL_F16X3:
 * scale_a, scale_b are the power-of-two tensor scales (|scale * x| must stay below 65504); mode PL_BF16: one bf16 plane
 * per operand, scales ignored.  M, N % 128 == 0, K % 32 == 0. */
size_t pl_gemm_planes_scratch_bytes(int64_t M, int64_t N, int64_t K);
int pl_gemm_planes(int layout, int mode, const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K,
                   const float* bias, float scale_a, float scale_b, void* scratch, void* stream);
/* The planes GEMM on operands that already ARE planes (1x1 convolutions of the conv path: forward NT on
 * [pixels][Cin] x [Cout][Cin], data gradient NT on dz [pixels][Cout] x W^T [Cin][Cout], weight gradient TN over the
 * pixels; reference phase4_joined/Resnet.py:56-63 and their autograd).  Layout as pl_gemm_f32 (0 NT, 1 NN, 2 TN);
 * a_plane / b_plane: elements between the two fp16 planes of an operand (unused for PL_BF16); lda / ldb: row strides
 * of the plane matrices in elements (% 8 == 0).  C = out_scale * [dyn_inv[0] *] (A B) (+ bias): out_scale = 1 / (S_A S_B),
 * dyn_inv a device scalar or NULL.  Any M; N % 8 == 0 (TN: M % 8 == 0 too); K % (32 * splits) == 0.  splits =
 * pl_gemm_planes_splits(M, N, K) > 1 needs slabs of splits * M * N floats (K slices summed in order into C).
 * stat (optional, unsplit problems; also on pl_conv2d_planes_fwd): [2][pl_gemm_stat_groups(M)][N] -- the epilogue also emits
 * training-mode BatchNorm partial statistics of C, for pl_bn_train_fwd_ex. */
int pl_gemm_planes_splits(int64_t M, int64_t N, int64_t K);
int pl_gemm_stat_groups(int64_t M);   /* 64-row statistics groups a GEMM epilogue emits for M rows (2 per 128-row tile) */
/* KxK convolutions the same way (implicit GEMM: the loader waves gather the NHWC input planes tap by tap, padding pixels
 * read as zeros; reference Resnet.py:58-60 conv2 and its autograd).  x_planes: planes of x [B][H][W][Cin], Cin % 32 == 0;
 * w_planes: planes of the OHWI kernel [Cout][KH*KW*Cin]; y [B][Ho][Wo][Cout] fp32.  The data gradient of a stride-1
 * convolution is the same call on the planes of dz with the kernel flipped and transposed ([Cin][KH][KW][Cout], pad
 * KH-1-pad).  wgrad: dw [Cout][KH*KW*Cin] = sum over output pixels of dz (planes [B][Ho][Wo][Cout]) x gathered x; slabs:
 * pl_gemm_planes_splits(Cout, KH*KW*Cin, B*Ho*Wo) * Cout * KH*KW*Cin floats when that is > 1.  out_scale / dyn_inv as above. */
int pl_conv2d_planes_fwd(int mode, const void* x_planes, int64_t x_plane, int64_t B, int64_t H, int64_t W, int64_t Cin,
                         const void* w_planes, int64_t w_plane, int64_t Cout, int KH, int KW, int stride, int pad,
                         float* y, float out_scale, const float* dyn_inv, float* stat, void* stream);
/* The head's last link on planes: the 3-D soft-argmax backward writing dlogits (4.5 GB at B = 256) as operand planes for the
 * final 1x1 convolution's gradient GEMMs (dlogits may be NULL), and the bias gradient as column sums of those planes.
 * PL_F16X3: dl_scale = {S, 1/S} on the device, S a power of two the caller derives from |dlogit| <= 2 max sum_c |g_c|. */
int pl_softargmax3d_nhwc_bwd_ex(const float* logits, const float* stats, const float* gcoords, int64_t B, int64_t J,
                                int64_t H, int64_t W, float* dlogits, void* dl_planes, int planes_mode,
                                const float* dl_scale, void* stream);
int pl_colsum_planes(const void* planes, int planes_mode, int64_t rows, int64_t cols, const float* inv_scale, float* out,
                     void* scratch, void* stream);
/* nn.ConvTranspose2d(4, 2, 1, bias=False) forward the same way (Model.py:47-63): four 2x2-tap gathers at the input
 * resolution, one per output parity, stored straight into y [B][2H][2W][Cout]; wsub_planes = planes of
 * conv.deconv_subkernels(weight) [4][Cout][2][2][Cin].  Its data gradient is pl_conv2d_planes_fwd (4x4, stride 2, pad 1)
 * on the planes of dy, its weight gradient pl_conv2d_planes_wgrad with the roles of x and dy exchanged. */
int pl_deconv4x4s2_planes_fwd(int mode, const void* x_planes, int64_t x_plane, int64_t B, int64_t H, int64_t W, int64_t Cin,
                              const void* wsub_planes, int64_t wsub_plane, int64_t Cout, float* y, float out_scale,
                              const float* dyn_inv, void* stream);
/* Eval-mode forms of the two: the Bottleneck's / the head's folded epilogue (BatchNorm on running statistics as
 * scale / shift, bias, ReLU, residual: pl_conv2d_nhwc_fwd's) applied to the convolution's result, which is written as fp32 (y,
 * may be NULL) and / or as operand planes for the next convolution (ep->y_planes, scale pl_conv_act_plane_scale(); NULL = none).
 * relu: 0 none, 1 ReLU then + resid, 2 + resid then ReLU.  Reference: Resnet.py:65-93, Model.py:47-69 under model.eval(). */
typedef struct PLPlanesEpilogue {
  const float* bias;      /* [Cout] or NULL */
  const float* scale;     /* [Cout] or NULL (with shift) */
  const float* shift;
  const float* resid;     /* [pixels][Cout] fp32 or NULL */
  int32_t relu;
  int32_t reserved;
  void* y_planes;         /* planes of the result (mode as the call's), or NULL */
} PLPlanesEpilogue;
int pl_conv2d_planes_fwd_ep(int mode, const void* x_planes, int64_t x_plane, int64_t B, int64_t H, int64_t W, int64_t Cin,
                            const void* w_planes, int64_t w_plane, int64_t Cout, int KH, int KW, int stride, int pad,
                            float* y, float out_scale, const PLPlanesEpilogue* ep, void* stream);
int pl_deconv4x4s2_planes_fwd_ep(int mode, const void* x_planes, int64_t x_plane, int64_t B, int64_t H, int64_t W,
                                 int64_t Cin, const void* wsub_planes, int64_t wsub_plane, int64_t Cout, float* y,
                                 float out_scale, const PLPlanesEpilogue* ep, void* stream);
int pl_conv2d_planes_wgrad(int mode, const void* dz_planes, int64_t dz_plane, const void* x_planes, int64_t x_plane,
                           int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int KH, int KW, int stride,
                           int pad, float* dw, float out_scale, const float* dyn_inv, float* slabs, void* stream);
/* The same two with separate strides and paddings along h and w, and -- forward -- an 8-channel input when KW % 4 == 0 (a 32-k
 * tile is then four neighbouring taps of one kernel row).  What this is for: the 7x7 / stride 2 / pad 3 stem on 3 input
 * channels (phase4_joined/Resnet.py:112-113,137) as a planes GEMM -- the frame padded to 4 channels and viewed as pixel PAIRS
 * [B][H][W/2][8], the kernel as 7 x 4 taps of 8 (zero where the pair window overhangs the 7 real taps), stride (2, 1),
 * padding 3 above and below, 2 pairs left and 1 right (pad_w / pad_w_right: W/2 outputs per row): conv.stem_planes. */
int pl_conv2d_planes_fwd_hw(int mode, const void* x_planes, int64_t x_plane, int64_t B, int64_t H, int64_t W, int64_t Cin,
                            const void* w_planes, int64_t w_plane, int64_t Cout, int KH, int KW, int stride_h, int stride_w,
                            int pad_h, int pad_w, int pad_w_right, float* y, float out_scale, const float* dyn_inv, float* stat,
                            void* stream);
int pl_conv2d_planes_fwd_ep_hw(int mode, const void* x_planes, int64_t x_plane, int64_t B, int64_t H, int64_t W, int64_t Cin,
                               const void* w_planes, int64_t w_plane, int64_t Cout, int KH, int KW, int stride_h, int stride_w,
                               int pad_h, int pad_w, int pad_w_right, float* y, float out_scale, const PLPlanesEpilogue* ep,
                               void* stream);      /* pl_conv2d_planes_fwd_ep with that geometry: the stem in eval mode */
int pl_conv2d_planes_wgrad_hw(int mode, const void* dz_planes, int64_t dz_plane, const void* x_planes, int64_t x_plane,
                              int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int KH, int KW, int stride_h,
                              int stride_w, int pad_h, int pad_w, int pad_w_right, float* dw, float out_scale,
                              const float* dyn_inv, float* slabs, void* stream);
int pl_gemm_planes_raw(int layout, int mode, const void* A, int64_t a_plane, int64_t lda, const void* B, int64_t b_plane,
                       int64_t ldb, float* C, int64_t M, int64_t N, int64_t K, const float* bias, float out_scale,
                       const float* dyn_inv, float* slabs, float* stat, void* stream);

/* ---- next row N1: fused softmax + integral soft-argmax ----------------------------------- */
/* Tail of Model_3D.forward  phase4_joined/Model.py:94-133 (ncoord 3, centred 1: (E/dim - 0.5)*2)
 * and of Model_2D.forward  phase5_loop/Model_2d.py:96-134 (ncoord 2, D 1, centred 0: E/dim).
 * logits [BJ][D][H][W] (BJ = batch*joints, W % 4 == 0): softmax over the D*H*W voxels of each
 * (batch, joint), then the expectation of the w / h / d index.  coords [BJ][ncoord] in (x, y, z)
 * order; stats [BJ][5] = {max, sum exp, Ex, Ey, Ez} is what the backward needs.
 * Backward: dlogits [BJ][D][H][W] from gcoords [BJ][ncoord]; re-reads logits, materialises nothing. */
/* {S, 1/S} (device, two floats) for pl_softargmax3d_nhwc_bwd_ex's fp16 planes of dlogits: S = the power of two that maps
 * the bound 2 max_rows sum_c |gcoords[row][c]| >= max |dlogit| into (2^13, 2^14]; a zero / non-finite bound gives 1. */
int pl_softargmax_dl_scale(const float* gcoords, int64_t rows, int ncoord, float* scale2, void* stream);
int pl_softargmax_fwd(const float* logits, int64_t BJ, int64_t D, int64_t H, int64_t W, int ncoord,
                      int centred, float* coords, float* stats, void* stream);
int pl_softargmax_bwd(const float* logits, const float* stats, const float* gcoords, int64_t BJ,
                      int64_t D, int64_t H, int64_t W, int ncoord, int centred, float* dlogits,
                      void* stream);

/* ---- measurement hook (bench.py; not part of the reference interface) ------------------ */
/* While enabled (on = n > 0), every n-th GEMM launch (n = 1: every one) is bracketed by two HIP events
 * recorded on the launch stream -- a pair costs ~2.5 us of stream time, so bench.py samples every 7th launch
 * of its timed region rather than all of them.  pl_prof_read waits for them and sums the durations of the launches whose
 * algorithmic work (2*M*N*K, summed over both problems of a dual launch) lies in
 * [min_flops, max_flops].  Enabling resets the record.  Not capturable. */
int pl_prof_enable(int on);
int pl_prof_read(double min_flops, double max_flops, double* ms_total, int64_t* launches,
                 double* flops_total);

#ifdef __cplusplus
}
#endif
#endif /* POSELIFT_H_ */
