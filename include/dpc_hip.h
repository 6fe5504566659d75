/* dpc_hip.h -- C ABI of libdpc_hip.so, the MI355X (gfx950) kernels of the DPC-RNN
 * training step.
 *
 * The reference (TengdaHan/DPC) has NO native/FFI layer: every op on its hot path is
 * a stock torch op reached through the Python nn.Module API.  This
 * header therefore *defines* the boundary a maintainer binds instead of those torch
 * ops; each entry cites the reference call site it replaces (paths relative to the
 * reference checkout).  INTEGRATION.md shows the ctypes stub on the reference side.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never
 *     allocates, never synchronises; workspaces are caller-owned;
 *   - return 0 on success, <0 on error (DPC_ERR_*); nothing is printed;
 *   - activations are channels-last [N][T][H][W][C]; dtype codes: 0 = f32, 1 = bf16.
 */
#ifndef DPC_HIP_H
#define DPC_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPC_F32 0
#define DPC_BF16 1

#define DPC_ERR_ARG (-1)
#define DPC_ERR_LAUNCH (-2)
#define DPC_ERR_UNSUPPORTED (-3)

typedef void* dpc_stream_t;

int dpc_abi_version(void);

/* ---- implicit-GEMM convolution / GEMM on the matrix cores ----------------------
 * Replaces nn.Conv3d forward and its autograd input-gradient in
 *   backbone/resnet_2d3d.py:14-32 (conv3x3x3 / conv1x3x3), :211 (stem, after the
 *   space-to-depth repack of dpc_pack_input_s2d), :241-244 (1x1x1 downsample),
 * nn.Conv2d 1x1 (ConvGRU gates backbone/convrnn.py:13-15,29-33; network_pred
 *   dpc/model_3d.py:36-40) and torch.matmul for the score (dpc/model_3d.py:83).
 *
 *   out[m][co] = sum_{tap,ci} src[gather(m,tap)][ci] * wgt[co][tap*Ci+ci]  (+ addend[m][co])
 * rows m enumerate (n, rt, rh, rw); mode 0 gathers rt*st-pt+kt (forward), mode 1
 * gathers (rt+pt-kt)/st when divisible (input-gradient; wgt is then the
 * [Ci][tap][Co] transpose produced by dpc_pack_weight).  A plain NT GEMM is the
 * case N=M, all spatial dims 1, one tap.  Strides must be 1 or 2; Ci a power of two
 * when there is more than one tap; Ci, src_ld, ldw multiples of 16 bytes.
 * If `stats` != NULL it receives per-program partial sums [rows][2][Co] (sum, sum of
 * squares of the stored outputs) for the batch-norm that follows; the row count is
 * dpc_conv_stats_rows(desc).
 * Aliasing: `addend` MAY be the same buffer as `out` (an in-place residual accumulation, dx += conv): every kernel behind
 * this entry and dpc_conv_igemm_ex reads a 16-byte addend unit in the lane that afterwards stores exactly that unit.  For a
 * strided input-gradient (mode 1, a stride of 2) with addend == out the positions no tap reaches are left as they are instead
 * of being rewritten with the addend.  No other pair of arguments may overlap.  tests/kcases.py::case_conv_dgrad_alias
 * runs every dispatch variant both ways (case_conv_dgrad_inplace: the untouched positions).
 */
typedef struct dpc_conv_desc {
    int32_t dtype_in, dtype_out; /* DPC_F32 / DPC_BF16 */
    int32_t mode;                /* 0 forward gather, 1 input-gradient gather */
    int32_t N, RT, RH, RW;       /* output positions                         */
    int32_t ST, SH, SW;          /* source tensor spatial dims               */
    int32_t Ci, src_ld;          /* channels per tap, source row stride      */
    int32_t Co, ldw, ldo;        /* output cols, weight row stride, out ld   */
    int32_t KT, KH, KW;
    int32_t st, sh, sw;
    int32_t pt, ph, pw;
} dpc_conv_desc;

int dpc_conv_stats_rows(const dpc_conv_desc* d);
int dpc_conv_igemm(const dpc_conv_desc* d, const void* src, const void* wgt, void* out,
                   const void* addend, float* stats, dpc_stream_t stream);

/* dpc_conv_igemm with backward pieces of the BasicBlock fused into the epilogue of an INPUT-GRADIENT launch (autograd of
 * backbone/resnet_2d3d.py:67-80,105-116), so that they stop being separate passes over HBM:
 *  - addend_mask: out = conv + (bit ? addend : 0), addend = the gradient arriving at the block output, bit = that output's
 *    ReLU sign mask (the byte-per-16-byte-unit mask dpc_bn_apply writes): the masked gradient dz is never materialised;
 *  - bn_raw (+ bn_mask, bn_mean, bn_invstd): the reduction of dpc_bn_bwd_reduce for the unit whose output gradient `out` is --
 *    dz = out gated by bn_mask, stats rows [dpc_conv_stats_rows(d)][2][Co] = (sum dz, sum dz * xhat), xhat = (bn_raw-mean)*invstd,
 *    taken from the stored (rounded) values; finish with dpc_bn_bwd_finalize, then dpc_bn_bwd_apply as before.
 * Without bn_raw `stats` means what it means for dpc_conv_igemm.  Requires ldo == Co, dtype_in == dtype_out, 16-byte aligned
 * tensors; DPC_ERR_UNSUPPORTED otherwise (run the separate kernels).  dpc_conv_plan with DPC_PLAN_ADDEND_MASK / DPC_PLAN_BNRED
 * tells which kernel serves the combination. */
typedef struct dpc_conv_epilogue {
    const void* addend;          /* optional [rows][ldo], dtype_out */
    const uint8_t* addend_mask;  /* optional, needs addend */
    const void* bn_raw;          /* optional [rows][ldo], dtype_out: raw conv output of the unit being differentiated */
    const uint8_t* bn_mask;      /* optional, needs bn_raw: ReLU mask of that unit's activation */
    const float* bn_mean;
    const float* bn_invstd;
    float* stats;
} dpc_conv_epilogue;
int dpc_conv_igemm_ex(const dpc_conv_desc* d, const void* src, const void* wgt, void* out, const dpc_conv_epilogue* epi,
                      dpc_stream_t stream);

/* NT GEMM with the reduction split over workgroups: part[ks][M][N] (f32) = A[M][K range ks] @ B[N][K range ks]^T, A / B row-major
 * with leading dimensions lda / ldb (dtype elements).  For products with a small output and a long reduction -- d_pred = dS @
 * feature_inf of the contrastive loss (dpc/main.py:217 backward: M = R, N = 256, K = R).  *nsplit = number of slabs (query with
 * part == NULL: capacity nsplit*M*N floats); sum them with dpc_reduce_unpack(part, nsplit, out, M, 1, N, N, 0, 1, 0). */
int dpc_gemm_nt_splitk(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* B, int32_t ldb,
                       float* part, int32_t* nsplit, dpc_stream_t stream);
/* The same with A given K-major: part[ks][M][N] = sum_{k in slice ks} A[k][m] * B[n][k] (A rows are lda elements apart, bf16 only).
 * d_feature_inf = dS^T @ pred of the contrastive loss (autograd of dpc/model_3d.py:83): A = dS [R][ld], B = pred^T [256][ld].
 * A rows beyond K are never read; lda >= M and ldb >= K rounded up to 8 elements, the padding of B's rows holding finite values.
 * DPC_ERR_UNSUPPORTED outside N % 128 == 0, lda % 8 == ldb % 8 == 0, M, K >= 1024 (use dpc_conv_wgrad's GEMM form there). */
int dpc_gemm_tn_splitk(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* B, int32_t ldb,
                       float* part, int32_t* nsplit, dpc_stream_t stream);

/* Weight gradient: part[ks][co][tap*Ci+ci] = sum_{m in split ks} dy[m][co]*src[gather(m,tap)][ci]
 * (autograd of the same convs / 1x1 convs / matmul; f32 partials, reduced by
 * dpc_reduce_unpack).  Returns the number of K-splits through *nsplit; capacity in
 * floats of `part` must be >= nsplit*Co*KT*KH*KW*Ci (query with part == NULL). */
int dpc_conv_wgrad(const dpc_conv_desc* d, const void* src, const void* dy, int32_t dy_ld,
                   float* part, int32_t* nsplit, dpc_stream_t stream);

/* ---- kernel selection, observable --------------------------------------------------------------------------------------------
 * dpc_conv_igemm / dpc_conv_wgrad pick one of several gfx950 kernels per shape (role-specialised loader/compute kernels,
 * staged-patch kernels, the generic implicit GEMM).  dpc_conv_plan returns the name of the kernel such a call WOULD launch
 * (the same dispatch code runs with the launch skipped; pointers are assumed 16-byte aligned): op = DPC_PLAN_IGEMM with
 * flags DPC_PLAN_ADDEND / DPC_PLAN_STATS for the optional operands, or DPC_PLAN_WGRAD with dy_ld.  dpc_last_kernel returns the
 * kernel the calling thread's most recent C-ABI call launched last.  Both write a NUL-terminated string (e.g.
 * "igemm_ws_kernel<true>", "wgrad2_kernel<T,NWM,NWN,true>[T=bf16 nwm=2 nwn=3 padded=1]") and return its length, or < 0. */
#define DPC_PLAN_IGEMM 0
#define DPC_PLAN_WGRAD 1
#define DPC_PLAN_ADDEND 1
#define DPC_PLAN_STATS 2
#define DPC_PLAN_ADDEND_MASK 4 /* dpc_conv_igemm_ex: gated addend */
#define DPC_PLAN_BNRED 8       /* dpc_conv_igemm_ex: fused BatchNorm-backward reduction */
int dpc_conv_plan(const dpc_conv_desc* d, int32_t op, int32_t flags, int32_t dy_ld, char* name, int32_t cap);
int dpc_last_kernel(char* name, int32_t cap);

/* CU carve-out for a concurrent collective: the kernels that launch one persistent workgroup per CU (layer1's patch kernel, the
 * loader/compute implicit GEMMs) shrink their grids by n workgroups (multiples of 8 are kept) for launches planned while the
 * setting is in force, so that RCCL's channel kernels find CUs during the overlapped gradient all-reduce (dpc/main.py:65's
 * DataParallel reduce, here one all-reduce per step).  Process-wide host state; returns the previous value.  dpc_conv_stats_rows
 * follows the setting: query it under the same value the launch sees. */
int dpc_set_reserved_cus(int32_t n);

/* Arithmetic of the f32 kernels' contractions (dtype_in == DPC_F32 in dpc_conv_igemm / dpc_conv_wgrad / dpc_gemm_nt_splitk):
 *   0  exact f32 MFMA chains (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak) -- the default, bitwise an fmaf chain;
 *   1  "bf16x6": each f32 operand split into three bf16 pieces in registers, the six products down to 2^-24 on the bf16 matrix
 *      pipe, f32 accumulation -- f32-grade results (score within 2e-4 of the reference's fp32 CPU path, north_star asks 1e-3) at up
 *      to 2.7x the f32 matrix rate.  Operands, activations and outputs stay f32 in HBM.
 * Process-wide host state read when a launch is planned; returns the previous mode. */
int dpc_set_f32_matmul(int32_t mode);

/* Diagnostic co-tenant ("squatter"): n_wg workgroups of `waves` (1..4) waves that hold lds_bytes of LDS (multiple of 16, up to
 * 160 KB) for usec microseconds and do nothing (mode 0), LDS traffic over their own allocation (1), 16-byte loads over
 * scratch[scratch_bytes] (2) or VALU work (3).  where[n_wg] (optional) receives 0x80000000 | XCC_ID << 16 | HW_ID[15:0] per
 * workgroup.  Emulates, on ONE GPU, what the train step meets in a multi-GPU run -- RCCL's channel kernels during the overlapped
 * all-reduce that replaces nn.DataParallel's gradient reduce (dpc/main.py:65,229) -- and any side-stream neighbour: used by
 * tests/test_cotenant_gpu.py and scripts/probes/squat_probe.py, never by the training path. */
int dpc_diag_squat(int32_t n_wg, int32_t waves, int32_t lds_bytes, int32_t mode, int32_t usec, const void* scratch,
                   int64_t scratch_bytes, uint32_t* where, uint32_t* sink, dpc_stream_t stream);

/* ---- weight / operand repacking --------------------------------------------------
 * out[i0][i1][i2] (dtype_out, dense) = in[i0*s0 + i1*s1 + i2*s2] (f32).  Turns the
 * reference's [Co][Ci][kT][kH][kW] parameters (state_dict layout, §8b) into the
 * K-contiguous operand layouts above without touching the parameters themselves. */
int dpc_pack3d(const float* in, void* out, int32_t dtype_out, int32_t d0, int32_t d1, int32_t d2,
               int64_t s0, int64_t s1, int64_t s2, dpc_stream_t stream);
/* the same for a whole table of tensors in one launch (all per-step weight repacks of the conv stack): entry e is served by
 * blocks [block0[e], block0[e+1]) of a grid of total_blocks x 256 threads; the table lives in DEVICE memory. */
typedef struct dpc_pack_entry {
    const void* in;   /* f32 source */
    void* out;        /* destination, dtype_out elements, contiguous [d0][d1][d2] */
    int32_t d0, d1, d2, block0;
    int64_t s0, s1, s2;
} dpc_pack_entry;
int dpc_pack3d_multi(const dpc_pack_entry* table_dev, int32_t n_entries, int32_t total_blocks, int32_t dtype_out, dpc_stream_t stream);
/* out[i0*s0+i1*s1+i2*s2] (f32) = sum_{k<nsplit} part[k][i0][i1][i2]  (+ out if accumulate) */
int dpc_reduce_unpack(const float* part, int32_t nsplit, float* out, int32_t d0, int32_t d1, int32_t d2,
                      int64_t s0, int64_t s1, int64_t s2, int32_t accumulate, dpc_stream_t stream);
/* two bf16 matrices of one shape in one launch: outK[j][i] = inK[i][j] (in1 / out1 may be NULL); 16-byte aligned, ld % 8 == 0.
 * pred and feature_inf -> the K-contiguous operands of the score's backward products (autograd of dpc/model_3d.py:83). */
int dpc_transpose2d_bf16x2(const void* in0, const void* in1, int32_t ld_in, void* out0, void* out1, int32_t ld_out, int32_t rows,
                           int32_t cols, dpc_stream_t stream);
/* 2-D transpose/convert between element types: out[j][i] = in[i][j] */
int dpc_transpose2d(const void* in, int32_t dtype_in, int32_t ld_in, void* out, int32_t dtype_out, int32_t ld_out,
                    int32_t rows, int32_t cols, dpc_stream_t stream);

/* Stem: Conv3d(3,64,(1,7,7),s(1,2,2),p(0,3,3)) (resnet_2d3d.py:211) is run as a 1x4x4
 * stride-1 conv over a 2x2 space-to-depth image with 16 channels (12 used).
 * dpc_pack_input_s2d: block [BN][3][T][H][W] f32 NCDHW (dpc/model_3d.py:49-50) ->
 *   [BN][T][H/2][W/2][16]; channel = (sy*2+sx)*3 + c.
 * dpc_pack_stem_weight: [Co][3][1][7][7] f32 -> [Co][16 taps][16]; dpc_unpack_stem_wgrad the inverse for grads. */
int dpc_pack_input_s2d(const float* block, void* out, int32_t dtype_out, int32_t BN, int32_t T, int32_t H, int32_t W,
                       dpc_stream_t stream);
int dpc_pack_stem_weight(const float* w, void* out, int32_t dtype_out, int32_t Co, dpc_stream_t stream);
int dpc_unpack_stem_wgrad(const float* part, int32_t nsplit, float* dw, int32_t Co, dpc_stream_t stream);

/* ---- batch norm with batch statistics (track_running_stats=False, dpc/model_3d.py:28;
 *      nn.BatchNorm3d at resnet_2d3d.py:55,59,93,97,212,243) -------------------------
 * finalize: partial sums [rows][2][C] -> mean, invstd, scale=gamma*invstd, shift=beta-mean*scale
 *           (f64 accumulation; count = elements per channel). */
int dpc_bn_finalize(const float* partials, int32_t rows, int32_t C, double count, const float* gamma,
                    const float* beta, float eps, float* mean, float* invstd, float* scale, float* shift,
                    dpc_stream_t stream);
/* y = act( x*scale+shift (+ res*rscale+rshift | + res) ), act = relu if relu!=0
 * (BasicBlock tail, resnet_2d3d.py:67-80,105-116) */
int dpc_bn_apply(const void* x, void* y, int32_t dtype, int64_t rows, int32_t C, const float* scale,
                 const float* shift, const void* res, const float* rscale, const float* rshift, int32_t relu,
                 uint8_t* mask, dpc_stream_t stream);
/* backward: dz = dy * (y>0 if relu); partial sums of dz and dz*xhat -> [rows][2][C].  The ReLU pattern comes
 * from `mask` (the byte-per-16-byte-unit sign mask dpc_bn_apply wrote: 1/16 of the bytes of y) or, when mask
 * is NULL, from y itself. */
int dpc_bn_bwd_reduce(const void* dy, const void* y, const uint8_t* mask, const void* x, int32_t dtype, int64_t rows, int32_t C,
                      const float* mean, const float* invstd, int32_t relu, float* partials, int32_t* prow,
                      dpc_stream_t stream);
/* sums [prow][2][C] -> dgamma , dbeta, and coefficients c1=sum_dz/count, c2=sum_dzxhat/count */
int dpc_bn_bwd_finalize(const float* partials, int32_t prow, int32_t C, double count, float* dgamma, float* dbeta,
                        float* coef, dpc_stream_t stream);
/* dx = gamma*invstd*(dz - c1 - xhat*c2); optionally also writes dz (residual branch grad) */
int dpc_bn_bwd_apply(const void* dy, const void* y, const uint8_t* mask, const void* x, int32_t dtype, int64_t rows, int32_t C,
                     const float* mean, const float* invstd, const float* gamma, const float* coef, int32_t relu,
                     void* dx, void* dz, dpc_stream_t stream);

/* ---- stem tail: BN+ReLU+MaxPool3d((1,3,3),s(1,2,2),p(0,1,1)) (resnet_2d3d.py:212-214,260-263) */
int dpc_bn_relu_maxpool_fwd(const void* x, int32_t dtype, int32_t NT, int32_t H, int32_t W, int32_t C,
                            const float* scale, const float* shift, void* y, uint8_t* argmax, dpc_stream_t stream);
/* dz (grad at the BN output, ReLU mask applied) from the pooled gradient and the saved argmax */
int dpc_maxpool_bwd(const void* dy, const uint8_t* argmax, int32_t dtype, int32_t NT, int32_t H, int32_t W, int32_t C,
                    void* dz, dpc_stream_t stream);

/* stem backward without materialising dz: BN backward whose incoming gradient is routed through the
 * max-pool on the fly (autograd of resnet_2d3d.py:260-263 in two passes instead of five) */
int dpc_pool_bn_bwd_reduce(const void* dy, const uint8_t* argmax, const void* x, int32_t dtype, int32_t NT, int32_t H,
                           int32_t W, int32_t C, const float* mean, const float* invstd, float* partials, int32_t* prow,
                           dpc_stream_t stream);
/* the same partial sums from the POOLED tensors only (y at an argmax position is the pooled value):
 * sum dz = sum_p dp[p], sum dz*xhat = sum_p dp[p]*(ypool[p]-beta)/gamma over valid argmax bytes */
int dpc_pooled_bn_bwd_reduce(const void* dy, const uint8_t* argmax, const void* ypool, int32_t dtype, int64_t rows,
                             int32_t C, const float* gamma, const float* beta, float* partials, int32_t* prow,
                             dpc_stream_t stream);
int dpc_pool_bn_bwd_apply(const void* dy, const uint8_t* argmax, const void* x, int32_t dtype, int32_t NT, int32_t H,
                          int32_t W, int32_t C, const float* mean, const float* invstd, const float* gamma,
                          const float* coef, void* dx, dpc_stream_t stream);
/* the stem's weight gradient straight from the gradient at the pooled output (throughput mode): pool_bn_bwd_apply's arithmetic
 * runs inside the weight-gradient kernel, the 2.7 GB full-resolution dz tensor is never written or re-read.  desc = the stem's
 * weight-gradient descriptor (space-to-depth form, dtype_in bf16, dtype_out f32).  part / nsplit as dpc_conv_wgrad (finish with
 * dpc_unpack_stem_wgrad).  DPC_ERR_UNSUPPORTED -> run dpc_pool_bn_bwd_apply + dpc_conv_wgrad instead. */
int dpc_stem_wgrad_fused(const dpc_conv_desc* d, const void* src_s2d, const void* raw, const void* dpool, const uint8_t* argmax,
                         const float* mean, const float* invstd, const float* gamma, const float* coef, float* part,
                         int32_t* nsplit, dpc_stream_t stream);

/* ---- temporal mean + ReLU split (dpc/model_3d.py:53-59) -----------------------------
 * x [B*N][T][SQ][D] -> feat_relu [N][B*SQ][D] (GRU input, time-major) and
 * feat_inf [B][P][SQ][D] (pre-ReLU, last P blocks, score operand). */
int dpc_tpool_split_fwd(const void* x, int32_t dtype, int32_t B, int32_t N, int32_t T, int32_t SQ, int32_t D,
                        int32_t P, void* feat_relu, void* feat_inf, dpc_stream_t stream);
int dpc_tpool_split_bwd(const void* x, const void* d_relu, const float* d_inf, int32_t dtype, int32_t B, int32_t N,
                        int32_t T, int32_t SQ, int32_t D, int32_t P, void* dx, dpc_stream_t stream);

/* ---- pieces around the fused ConvGRU recurrence (dpc_gru_chain_* below) -------------------
 * colsum of [M][D] (leading dimension ld) into out[D] (+= when accumulate): bias gradients of the ConvGRU / network_pred */
int dpc_colsum(const void* x, int32_t dtype, int32_t ld, int32_t M, int32_t D, float* out, int32_t accumulate,
               float* ws, int64_t ws_floats /* >= 64*D */, dpc_stream_t stream);

/* ---- contrastive loss head (dpc/main.py:178-185,213-218; utils/utils.py:38-55) -------
 * mask: closed form of dpc/model_3d.py:86-96, int8 [B][P][SQ][B][P][SQ] contiguous.
 * ce_topk: rows x cols logits (ld), target[i] = i (single-GPU closed form of process_output):
 *   result[0..3] = mean CE loss, top1, top3, top5; dscore (optional, dtype/ld given) = (softmax-onehot)/rows. */
int dpc_mask_gen(int8_t* mask, int32_t B, int32_t P, int32_t SQ, dpc_stream_t stream);
int dpc_ce_topk(const float* score, int32_t rows, int32_t cols, int32_t ld, float* row_ws, float* result,
                void* dscore, int32_t dtype_d, int32_t ld_d, dpc_stream_t stream);
/* the same on bf16 logits with a bf16 gradient: the materialised score of a train step that does not return it (dpc/main.py:198-218
 * reads `score_` only through the loss).  cols % 8 == 0, ld % 8 == 0, 16-byte aligned rows, cols <= 16384; else DPC_ERR_UNSUPPORTED. */
int dpc_ce_topk_bf16(const void* score_bf16, int32_t rows, int32_t cols, int32_t ld, float* row_ws, float* result,
                     void* dscore_bf16, int32_t ld_d, dpc_stream_t stream);

/* ---- fused score + loss (throughput mode, bf16 operands): the [R][R] score and its gradient never touch HBM --------
 * score_fwd: S = pred @ finf^T (dpc/model_3d.py:83) tile by tile; per row the running max / sum-exp / #{s > s_target}
 *   (target = the diagonal, the closed form of process_output on one GPU) -> row_ws[R][2] = (lse - s_target, rank),
 *   diag[R] = s_target, lse2[R] = (lse + ln R) log2 e for the backward; `score` (optional, f32 [R][R]) materialises S for a
 *   caller that reads it.  Follow with dpc_ce_finalize(row_ws, R, result) for loss / top-1/3/5 (dpc/main.py:217-218).
 * score_bwd: one of the two products of the backward with dS = (softmax(S) - onehot)/R recomputed from the operands:
 *   by_owner = 1, own = pred, oth = finf, othT = finf^T  -> partial slabs of d_pred = dS @ finf
 *   by_owner = 0, own = finf, oth = pred, othT = pred^T  -> partial slabs of d_finf = dS^T @ pred
 *   othT is [D][ldT] (ldT % 8 == 0, zero beyond column R).  Returns the number of [R][D] f32 slabs written to out_part
 *   (sum them with dpc_reduce_unpack).  dpc_score_ws_floats: workspace sizes (returns the number of slabs / column splits).
 * D must be 256 (or 32, the width-reduced test networks); other widths: DPC_ERR_UNSUPPORTED (use the materialised path). */
int dpc_score_ws_floats(int32_t R, int32_t D, int64_t* fwd_floats, int64_t* bwd_floats);
int dpc_score_fwd(const void* pred, const void* finf, int32_t R, int32_t D, float* diag, float* lse2, float* row_ws,
                  float* score, float* ws, dpc_stream_t stream);
int dpc_score_bwd(const void* own, const void* oth, const void* othT, int32_t ldT, int32_t R, int32_t D, const float* lse2,
                  int32_t by_owner, float* out_part, dpc_stream_t stream);
int dpc_ce_finalize(const float* row_ws, int32_t rows, float* result, dpc_stream_t stream);

/* ---- Adam with L2 weight decay on flat f32 buffers (dpc/main.py:80-81) --------------- */
int dpc_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
             float wd, float bias_corr1, float bias_corr2, float grad_scale, dpc_stream_t stream);

/* graph-capturable optimizer step (the whole train step is replayed as ONE hipGraph, dpc_amd/engine.py): the step
 * counter t and Adam's bias corrections 1-beta^t live in device memory.  step_advance: t += 1, bc = {1-b1^t, 1-b2^t}.
 * adam_dev == dpc_adam with the two corrections read from bias_corr_dev[0..1]. */
int dpc_step_advance(int32_t* step_dev, float* bias_corr_dev, double beta1, double beta2, dpc_stream_t stream);
int dpc_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, double beta1, double beta2, float eps,
                 float wd, const float* bias_corr_dev, float grad_scale, dpc_stream_t stream);
/* counter_dev[0] += 1 on the stream: the dropout draw counter (one draw per train-mode forward; the `step_dev` the recurrence and
 * the classifier head key their Philox masks on), graph-capturable like dpc_step_advance */
int dpc_counter_advance(int32_t* counter_dev, dpc_stream_t stream);
/* f32 [rows][cols] window copy between two leading dimensions (ConvGRU gate-gradient scatter) */
int dpc_copy2d_f32(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int32_t rows, int32_t cols, dpc_stream_t stream);
/* the same for a device-resident table of windows in ONE launch (the nine gate slices of the ConvGRU's weight / bias gradients,
 * backbone/convrnn.py:13-15: [D][2D] parameters filled from batched [3D][D] GEMM results); block0 = first workgroup of an entry */
typedef struct dpc_copy2d_entry {
    const float* src;
    float* dst;
    int64_t src_ld, dst_ld;
    int32_t rows, cols, block0, pad;
} dpc_copy2d_entry;
int dpc_copy2d_multi(const dpc_copy2d_entry* table_dev, int32_t n_entries, int32_t total_blocks, dpc_stream_t stream);

/* ---- ConvGRU dropout (nn.Dropout(p=0.1) on the carried hidden state, backbone/convrnn.py:39,59,78) ---------------
 * mask[i] = 1/(1-p) w.p. 1-p else 0: Philox4x32-10, key = seed, counter = (i/4, step_dev[0], stream id (0 here; 1 = the LC head's dropout), 0), word i%4,
 * keep iff (word >> 8) >= round(p * 2^24).  One launch draws the masks of all recurrence steps of one train step. */
int dpc_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, const int32_t* step_dev, dpc_stream_t stream);

/* ---- the whole ConvGRU aggregate / predict recurrence in one launch, and its backward in one more --------------
 * (dpc/model_3d.py:62-72: agg over the first N-P blocks, then P x {network_pred, agg step}; backbone/convrnn.py:24-34,76-79;
 *  network_pred dpc/model_3d.py:36-40).  Rows m = (b, s) of the [B*SQ][D] state are independent sequences (1x1 convolutions):
 *  a workgroup owns 32 rows for all steps.  T = compute dtype (dtype field); all buffers are caller-owned.
 *  P = 0: aggregation only (n_steps = n_agg), the classifier head of eval/model_3d_lc.py reads H_all[n_steps].
 * dpc_gru_pack repacks the five f32 parameters (gate weights [D][2D] = [x half | h half], network_pred [D][D]) into the
 *  fragment-major operand layout both kernels stream (`packed`: 16 * D * D elements of T); call it once per optimizer step.
 * Dropout on the carried state: drop_masks != NULL -> explicit pre-scaled keep masks [n_steps][M][D]; else step_dev != NULL ->
 *  Philox4x32-10 keyed on (seed, step_dev[0]) generated in the kernel (the bits dpc_dropout_mask writes); else none (eval). */
typedef struct dpc_gru_chain_desc {
    int32_t dtype, M, D, SQ, P, n_agg, n_steps, reserved;
    float p_drop;
    uint32_t reserved2;
    uint64_t seed;
    const int32_t* step_dev;
    const float* drop_masks;
    const void* packed;
    const float *bias_u, *bias_r, *bias_o, *bias_1, *bias_2;
    void* X_all;   /* [n_steps][M][D] T   in: relu'd features of the n_agg aggregation steps; out: relu(pred_i) for the others */
    void* H_all;   /* [n_steps+1][M][D] T  [0] = h_0 (read), [s+1] = state after step s (written) */
    void* HR_all;  /* [n_steps][M][D] T    h * r */
    float *U_all, *R_all, *O_all;  /* [n_steps][M][D] f32 gate activations (saved for the backward) */
    void* P1_all;  /* [P][M][D] T          relu(W1 h + b1) */
    void* pred;    /* [B][P][SQ][D] T      predictions, rows in the score's order */
    /* backward only */
    const float* d_pred;  /* [B][P][SQ][D] f32   d loss / d pred */
    void* G_all;          /* [n_steps][M][3D] T  pre-activation gradients [u | r | o] (operands of the batched weight gradients) */
    void *dP1, *dP2;      /* [P][M][D] T         gradients at the pre-activations of network_pred */
    float* d_x;           /* [n_agg][M][D] f32   gradient w.r.t. the aggregation inputs */
    float* ws;            /* [2][M][D] f32       scratch of the backward kernel */
    const float* d_hlast; /* optional [M][D] f32  d loss / d (last state): seeds the backward (P = 0: the LC classifier, eval/model_3d_lc.py:58-60) */
} dpc_gru_chain_desc;
int dpc_gru_pack(const float* w_update, const float* w_reset, const float* w_out, const float* w_pred0, const float* w_pred2,
                 int32_t D, int32_t dtype, void* packed, dpc_stream_t stream);
int dpc_gru_chain_fwd(const dpc_gru_chain_desc* c, dpc_stream_t stream);
int dpc_gru_chain_bwd(const dpc_gru_chain_desc* c, dpc_stream_t stream);

/* ---- downstream classifier LC (eval/model_3d_lc.py:12-65) ------------------------------------------
 * Shares the backbone and ConvGRU kernels; what differs:
 *  - the backbone's BatchNorm3d layers track running statistics (model_3d_lc.py:27-29): dpc_bn_finalize_running is
 *    dpc_bn_finalize + the momentum update of running_mean / running_var (unbiased) in train mode, dpc_bn_eval_coeffs builds
 *    scale / shift from the running buffers in eval mode;
 *  - relu_tpool: feat[n][(b,s)][d] = mean_t relu(x[b*N+n][t][s][d])  (ReLU BEFORE the temporal mean, model_3d_lc.py:52-54)
 *    and its backward dx = (x > 0) d_feat / T;
 *  - the head (model_3d_lc.py:58-64, eval/test.py:244-255): spatial mean of the last ConvGRU state -> BatchNorm1d ->
 *    Dropout(p) -> Linear -> CrossEntropyLoss + top-1, forward and backward.  result[0..1] = mean loss, accuracy. */
int dpc_bn_finalize_running(const float* partials, int32_t rows, int32_t C, double count, const float* gamma, const float* beta,
                            float eps, float* mean, float* invstd, float* scale, float* shift, float* running_mean,
                            float* running_var, int64_t* num_batches_tracked, float momentum, dpc_stream_t stream);
int dpc_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                       int32_t C, float* mean, float* invstd, float* scale, float* shift, dpc_stream_t stream);
int dpc_relu_tpool_fwd(const void* x, int32_t dtype, int32_t B, int32_t N, int32_t T, int32_t SQ, int32_t D, void* feat,
                       dpc_stream_t stream);
int dpc_relu_tpool_bwd(const void* x, const float* d_feat, int32_t dtype, int32_t B, int32_t N, int32_t T, int32_t SQ, int32_t D,
                       void* dx, dpc_stream_t stream);
typedef struct dpc_lc_head_desc {
    int32_t dtype, B, SQ, D, num_class, train;
    float p_drop, momentum, eps;
    uint32_t reserved;
    uint64_t seed;
    const int32_t* step_dev;      /* Philox step counter (train, when drop_mask is NULL) */
    const float* drop_mask;       /* optional explicit pre-scaled keep mask [B][D] */
    const void* h_last;           /* [B*SQ][D] T: last ConvGRU state */
    const float *bn_weight, *bn_bias;
    float *bn_running_mean, *bn_running_var;
    int64_t* bn_num_batches;      /* optional */
    const float *fc_weight, *fc_bias;   /* [num_class][D], [num_class] */
    const int64_t* target;        /* [B] class indices */
    float *ctx, *xhat, *bn_out, *y;     /* [B][D] f32: spatial mean, normalised, BN output (= LC.forward's `context`), after dropout */
    float* stat;                  /* [2][D] mean / invstd used */
    float *logits, *dlogits;      /* [B][num_class]: LC.forward's `output`; d loss / d logits */
    float *row_ws, *result;       /* [B][2], [2] */
    /* backward */
    float *g_fc_weight, *g_fc_bias, *g_bn_weight, *g_bn_bias;
    float* dctx;                  /* [B][D] scratch */
    float* d_hlast;               /* [B*SQ][D] f32: seed of dpc_gru_chain_bwd */
} dpc_lc_head_desc;
int dpc_lc_head_fwd(const dpc_lc_head_desc* c, dpc_stream_t stream);
int dpc_lc_head_bwd(const dpc_lc_head_desc* c, dpc_stream_t stream);

/* ---- GPU-side input pipeline (dpc/dataset_3d.py:85-111, utils/augmentation.py) ---------------------
 * frames u8 [B][F][H0][W0][3] (decoded RGB, HWC) -> block f32 [B][N][3][SL][H][W] (the DPC_RNN / LC input) and / or the stem's
 * space-to-depth operand (what dpc_pack_input_s2d writes), applying per clip: frame sampling frame(n, sl) = start + (n*SL + sl)*ds,
 * the crop box (x1, y1) of RandomCrop / RandomSizedCrop, RandomHorizontalFlip (flip), per-frame RandomGray channel choice
 * (gray [B][N*SL], -1 = keep colour; NULL = none), Scale with NEAREST interpolation (xtab [W] / ytab [H]: device tables output
 * column / row -> column / row inside the crop_w x crop_h crop box, produced by PIL on the host; NULL = no scaling),
 * ToTensor (/255) and Normalize(mean3, std3; HOST arrays of 3 floats).  flip: 0 none, 1 after crop + scale (k400 recipe),
 * 2 of the full frame before the crop (ucf101 recipe, dpc/main.py:114-123).  The random draws stay on the host (a handful of
 * integers per clip; dpc_amd/data.py mirrors the reference's calls of `random` / `np.random`).
 * dpc_frames_to_input_ex adds, bit-exact against PIL / torchvision's PIL path:
 *  - dpc_resample: PIL's separable resampling (Image.resize BILINEAR / NEAREST / ..., 8 bits per channel) as per-clip tables --
 *    output column x reads source columns x1 + xb[x][0] + [0, xb[x][1]) of the crop box with the 22-bit fixed-point coefficients
 *    xk[x][.]; rows likewise; horizontal pass rounded to 8 bits first (RandomSizedCrop's resize, utils/augmentation.py:144-196);
 *  - dpc_frame_jitter: ColorJitter (utils/augmentation.py:253-351), one draw per frame: order[k] = the k-th step of the shuffled
 *    chain (0 brightness, 1 contrast, 2 saturation, 3 hue, 255 none), factor[0..2] = brightness / contrast / saturation factors,
 *    hue_shift = np.uint8(hue_factor * 255).  Needs u8_ws (B*N*SL*H*W*3 bytes) and lsum_ws (B*N*SL uint64). */
typedef struct dpc_clip_aug {
    int32_t start, x1, y1, flip;
} dpc_clip_aug;
int dpc_frames_to_input(const uint8_t* frames, int32_t B, int32_t F, int32_t H0, int32_t W0, const dpc_clip_aug* aug,
                        const int8_t* gray, int32_t N, int32_t SL, int32_t ds, int32_t H, int32_t W, const int32_t* xtab,
                        const int32_t* ytab, int32_t crop_w, int32_t crop_h, const float* mean3, const float* std3, float* block,
                        void* s2d, int32_t dtype_s2d, dpc_stream_t stream);
typedef struct dpc_resample {
    const int32_t *xb, *xk;   /* device: [B][W][2] (first source column, taps), [B][W][ksx] */
    const int32_t *yb, *yk;   /* device: [B][H][2], [B][H][ksy] */
    int32_t ksx, ksy;
} dpc_resample;
typedef struct dpc_frame_jitter {
    float factor[3];
    int32_t hue_shift;
    uint8_t order[4];
} dpc_frame_jitter;
int dpc_frames_to_input_ex(const uint8_t* frames, int32_t B, int32_t F, int32_t H0, int32_t W0, const dpc_clip_aug* aug,
                           const int8_t* gray, int32_t N, int32_t SL, int32_t ds, int32_t H, int32_t W, const dpc_resample* rs,
                           const dpc_frame_jitter* jitter, uint8_t* u8_ws, uint64_t* lsum_ws, const float* mean3, const float* std3,
                           float* block, void* s2d, int32_t dtype_s2d, dpc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DPC_HIP_H */
