// Host-side launch API of the gfx950 kernel library (raw device pointers + hipStream_t).
// Python reaches these through csrc/bind.cpp; nothing here depends on torch.
#pragma once
#include "../common.h"

namespace dpa {

// ------------------------------------------------------------------------------ igemm.hip
// kIgemmRow3: the caller guarantees taps in groups of 3 with equal dy and dx = x0, x0+1, x0+2 (one x0 for all
// groups): 3x3-like stride-1 "same" convolutions may then run the filter-row kernel (one staged A tile per 3 taps)
// kIgemmAxf: A is the RAW input z of a training-mode BatchNorm + ReLU; the kernel applies a = relu(a_scale[c] * z +
// a_shift[c]) while it stages the A tile (the normalised activation is never written to memory).  16-bit dense
// problems with 64-byte aligned reduction channels only (igemm_axf_ok).
// kIgemmRStrided (lean GEMM only): R is compact [NB, OHF, OWF, ldc]; output pixel (y, x) adds R[y / OSY][x / OSX] when
// y % OSY == 0 and x % OSX == 0 (powers of two) -- the data gradient of a strided 1x1 branch, never scattered into a
// zero-filled full-size tensor.
// kIgemmReluMask (lean GEMM only, with a bias and R): C = relu(acc + bias + R), and one byte per 16-byte chunk of C goes
// to relu_mask (bit j = element j > 0): the tail of a bottleneck -- scaled 1x1 convolution + BN shift + identity + ReLU
// -- in the convolution's epilogue (hip_ops._ConvBnResReluFn).
// kIgemmBnFold (with kIgemmBnBwd, bn_relu 0 or 3): the BatchNorm's producer is a 1x1 convolution whose backward folds the
// BatchNorm backward into its own GEMMs (hip_ops._ConvFn, "bn fold"): the gradient is STORED through the ReLU mask, the
// slab receives only sum dz_m (second row zeros) and bn_x is never read -- sum dz_m * xhat follows from the weight gradient.
// kIgemmDense1: the caller guarantees ONE tap (0, 0, 0) -- a 1x1 convolution without padding or a Linear layer --
// so every output pixel (oy, ox) reads input pixel (oy * SY, ox * SX): such launches may run the lean GEMM
// (gemm_lean.hip: no tap bookkeeping, no padding masks).
enum IgemmFlags : int { kIgemmRelu = 1, kIgemmScatter = 2, kIgemmBnBwd = 4, kIgemmRow3 = 8, kIgemmNtStore = 16,
                        kIgemmAxf = 32, kIgemmDense1 = 64, kIgemmRStrided = 128, kIgemmXMask = 256, kIgemmBnFold = 512,
                        kIgemmReluMask = 1024 };

struct IgemmParams {
  const void* A;       // gathered operand: NHWC activations [NB, IH, IW, lda]
  const void* B;       // weights [N][ldb], reduction axis contiguous
  void* C;             // output rows of ldc elements
  const void* R;       // optional tensor added in the epilogue (same dtype / rows as C): C = conv + R
  const uint8_t* r_mask = nullptr;   // optional: one byte per 16-byte chunk of R, bit j = keep element j (the ReLU
                                     // mask of the BN that produced R's gradient: R is added as R * mask)
  const float* bias;   // optional [N]
  float* stats;        // optional per-channel partial (sum, sum^2) slab [tiles_m][2][N]
  // kIgemmBnBwd (data-gradient launches): the output is the gradient dz of z = [relu](bn(bn_x) [+ res]); the
  // slab then receives the BatchNorm-backward partials (sum dz_m, sum dz_m * xhat) instead, dz_m = dz masked by
  // the ReLU (bn_relu 0: none, 1: bn_z > 0, 2: scale*bn_x + shift > 0, 3: bn_z is a bit mask, 6 = 2 | 4: ReLU6, mask
  // 0 < scale*bn_x + shift < 6), so BN backward skips its reduce pass.
  const void* bn_x;    // BN input, same rows / ldc as C
  const void* bn_z;    // BN output (bn_relu == 1)
  const float* bn_coef;  // [4][N]: mean, invstd, scale, shift
  int bn_relu;
  const float* a_scale = nullptr;   // kIgemmAxf: [Cred] each
  const float* a_shift = nullptr;
  // second gathered operand (lean GEMM only, dense 1x1): the reduction continues over Cred2 channels of A2 (rows of lda2
  // elements, same pixels), against columns Cred .. Cred + Cred2 - 1 of B -- C = [A | A2] [B1 | B2]^T in one launch
  // (the "bn fold" data gradient: dy_m (A o W) + a (W^T diag(B) W))
  uint8_t* relu_mask = nullptr;      // kIgemmReluMask: the ReLU bit mask of C (output)
  const void* A2 = nullptr;
  int lda2 = 0, Cred2 = 0;
  const void* B2 = nullptr;         // optional: the second range's weights live in their own matrix [N][ldb2] (else: in B)
  int ldb2 = 0;
  // grouped convolution: grid.y = group; a group's slice starts g * stride ELEMENTS into each operand (groups
  // <= 1: dense).  N, Cred are per group; bias / statistics are indexed by the global channel g * c_gstride + n,
  // stats_ld is the slab's row width (all groups; 0 = N).
  int groups = 1;
  long a_gstride = 0, b_gstride = 0, c_gstride = 0;
  int stats_ld = 0;
  const int* taps;     // device int32 triples (dy, dx, wtap), ntaps entries
  int M, N, Cred, ntaps;
  int lda, ldb, ldc;
  int IH, IW;          // gathered image extent
  int SY, SX;          // gathered-coordinate step per output-grid step
  int OHF, OWF, OSY, OSX, OY0, OX0;  // scatter of the (sub)grid into the full output (kIgemmScatter)
  int flags;
  FastDiv div_ohow, div_ow;          // (sub)grid OH*OW and OW
};
void igemm_launch(const IgemmParams& p, int dtype, int out_dtype, hipStream_t stream);
// whether a launch of this shape can take kIgemmAxf (the caller materialises the activation otherwise)
bool igemm_axf_ok(long ohow, int IH, int IW, int lda, int N, int Cred, int ldb, int dtype, int out_dtype, int groups);
int igemm_tile_m(int dtype);
// rows of the per-tile BatchNorm statistics slab the launch for this problem will write
int igemm_stats_rows(int M, int N, int Cred, int ntaps, int dtype, int out_dtype);
// tuning hook: force tile configuration `cfg` (16-bit in == out problems), -1 = automatic
void igemm_force_config(int cfg);
// tuning hook for the filter-row kernel: -1 automatic, 0 never, 1 64-byte K tiles, 2 128-byte K tiles
void igemm_set_row3(int mode);
void igemm_set_tail_min(int cred);   // K-tile tails for 16-byte aligned reductions of >= cred channels (0: off)
// gemm_lean.hip: takes single-tap 16-bit launches (kIgemmDense1) it covers, false = run the general kernel.
bool gemm_lean_try(const IgemmParams& p, int dtype, int out_dtype, hipStream_t stream);
void igemm_set_lean(int mode);
void igemm_set_lean_ablate(int a);   // debug builds (-DDPA_LEAN_ABLATE) only: 1 no MFMA, 2 no compute, 3 no loop loads, 4 no store      // tuning hook: -1 automatic, 0 never, 1 / 2 force 64- / 128-byte K tiles
void igemm_set_nt_store(int mode);  // streaming (nontemporal) stores of the output tile: 0 by size, 1 always, -1 never
// tensors of at least this many bytes are streamed with nontemporal loads / stores (negative: never)
void set_nt_min_bytes(long bytes);
long nt_min_bytes();
// Workgroup cap of the grid-stride streaming kernels (elementwise, pooling, depthwise).  Not a power of two: see bn.hip.
extern int g_stream_grid;
void set_stream_grid(int blocks);
void set_nt_mask(int mask);   // 1 BN apply, 2 BN backward apply, 4 BN backward reduce, 8 igemm output stores
int igemm_num_configs();
// conv8.hip: the 256 x 256 multi-phase pipeline with an implicit-GEMM front end, fused epilogues and a stream-K
// schedule; takes stride-1 "same" 16-bit launches it covers and expects to win, false = run the other kernels.
bool conv8_covers(const IgemmParams& p, int dtype, int out_dtype);
bool conv8_try(const IgemmParams& p, int dtype, int out_dtype, hipStream_t stream);
void conv8_set_mode(int mode);    // -1 automatic, 0 never, 1 / 2 always one tile per workgroup / equal-range stream-K (when covered)
void conv8_set_tail_split(int on);
void conv8_set_min_k(int k);      // automatic mode: shortest reduction (elements) it takes
int conv8_error(hipStream_t stream);
long conv8_launches();
int conv8_last(int i);       // diagnostics: 0 grid, 1 ranges (> 0) or -pieces of the tail split, 3 CUs

// ------------------------------------------------------------------------------ wgrad.hip
struct WgradParams {
  const void* dY;      // [M][ldy]  output gradient rows (pixels of the OH x OW grid)
  const void* X;       // NHWC input [NB, IH, IW, ldx]
  float* slab;         // fp32 partials [splits][N][Ktot]   (N = out channels, Ktot = ntaps*Cred)
  const int* taps;     // (dy, dx, wtap) triples
  int M, N, Cred, ntaps, Ktot;
  int ldy, ldx;
  int IH, IW, SY, SX;
  int splits, pix_per_split;
  int groups = 1;      // grid.y = group: dY / X slices start g * stride elements in, slab rows g * N .. (g+1) * N
  long y_gstride = 0, x_gstride = 0;
  int slab_rows = 0;   // rows of one split's slab = groups * N
  int taps_zero;       // every tap is (0, 0): a 1x1 convolution without padding (enables the linear fast path)
  // X is the raw input z of a BatchNorm + ReLU: the kernel stages relu(x_scale[c] * z + x_shift[c]) (16-bit, dense)
  const float* x_scale = nullptr;
  const float* x_shift = nullptr;
  // second gradient operand (dense 1x1 problems only): output rows N .. N + N2 - 1 are dY2^T X with dY2 [M][ldy2] -- the
  // Gram matrix X^T X of the "bn fold" (dY2 = X) rides the launch that computes dY^T X, X is staged once for both
  const void* dY2 = nullptr;
  int N2 = 0, ldy2 = 0;
  FastDiv div_ohow, div_ow;
};
void wgrad_launch(const WgradParams& p, int dtype, hipStream_t stream);
// out[i] = beta*out[i] + sum_s slab[s][i]   (out dtype f32/bf16/f16), optional column sums
void slab_reduce_launch(const float* slab, int splits, long n, void* out, int out_dtype, float beta,
                        hipStream_t stream);
int wgrad_choose_splits(long M, int N, int Ktot, int dtype, int* pix_per_split, int groups = 1);
void wgrad_force_config(int cfg);   // tuning hook, -1 = automatic
int wgrad_num_configs();
void wgrad_set_target_blocks(int n);   // tuning hook: workgroups the pixel split aims for

// ------------------------------------------------------------------------------ wgrad8.hip
// Weight gradient of wide layers (dense 1x1 + second gradient operand, or gathered taps with Cred % 128 == 0) on the
// 256 x 256 LDS-DMA pipeline with transposed fragment reads.  Same slab layout as wgrad_launch ([splits][N + N2][C]); the caller asks wgrad8_covers first and takes the
// pixel split from wgrad8_choose_splits (multiples of 64 pixels, one workgroup per CU).
bool wgrad8_covers(long M, int N, int N2, int Cred, int ntaps, bool gather, int dtype);
int wgrad8_choose_splits(long M, int N, int N2, int Ktot, int* pix_per_split);
void wgrad8_launch(const WgradParams& p, int dtype, hipStream_t stream);
void wgrad8_set_mode(int mode);       // -1 automatic, 0 never, 1 whenever covered

// ------------------------------------------------------------------------------ wgrad_row3.hip
// Filter-row weight gradient (R x 3, stride 1, "same"): one staged activation row tile serves the three taps of a filter
// row.  wgrad_row3_splits() == 0: not covered (image width > 64, fp32, channels not a multiple of 64): use wgrad_launch.
// Slab layout [splits][N][R*3*Cred] as wgrad_launch writes it.
int wgrad_row3_splits(int NB, int H, int W, int N, int Cred, int R, int dtype, int* rows_per_split);
void wgrad_row3_launch(const void* dY, const void* X, float* slab, int NB, int H, int W, int N, int Cred, int R, int ph,
                       int ldy, int ldx, int splits, int rows_per_split, int dtype, hipStream_t stream);
void wgrad_row3_set_mode(int mode);
void wgrad_row3_set_variant(int v);   // -1 automatic, 0 never

// ------------------------------------------------------------------------------ bn.hip
// x, y, residual, dy, dx: [M][C] (NHWC flattened), stats fp32.
int bn_partial_rows(long M, int C);
// folds [nparts][2][C] partials into [nout][2][C] (nout*ceil(nparts/nout) >= nparts; empty parts give 0)
void bn_fold_partials_launch(const float* partial, int nparts, int C, float* out, int nout, hipStream_t s);
void bn_stats_launch(const void* x, long M, int C, int dtype, float* partial, int nblk, hipStream_t s);
// Forward statistics slabs ("stats"): [nparts][2][C] floats = (sum, M2 about the part's own mean) per part, followed by
// [nparts] floats = rows of each part (bn_stats_floats()).  These are the partials of Chan's parallel variance; no
// kernel forms E[x^2] - mean^2 in fp32.  Producers: bn_stats_launch, the igemm epilogue (IgemmParams::stats), dw_conv.
inline long bn_stats_floats(long nparts, long C) { return nparts * 2 * C + nparts; }
// local moments of one rank (SyncBatchNorm): mean[C], M2[C] about that mean, count_out[0] = count
void bn_fold_moments_launch(const float* partial, int nparts, int C, double* tmp, int* counter, double count,
                            float* mean, float* m2, float* count_out, hipStream_t s);
// gathered: [W][2C + 1] = (mean_r, M2_r, n_r) of every rank -> Chan merge -> mean, invstd, scale/shift, running update
void bn_sync_merge_launch(const float* gathered, int W, int C, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, long* num_batches_tracked,
                          float* mean, float* invstd, float* scale, float* shift, float* count_out, hipStream_t s);
void bn_eval_coeffs_launch(const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float eps, int C, float* scale, float* shift,
                           hipStream_t s);
// fold + finalize in one launch (last-arriver pattern); tmp: [bn_fold_chunks(nparts)][2][C] doubles, counter: >= 64
// zero-initialised ints that the kernel leaves zeroed (one launch at a time per counter array)
int bn_fold_chunks(int nparts);
void bn_fold_finalize_launch(const float* partial, int nparts, int C, double* tmp, int* counter, double count,
                             const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                             float* running_var, long* num_batches_tracked, float* mean, float* invstd, float* scale,
                             float* shift, hipStream_t s);
void bn_bwd_fold_finalize_launch(const float* partial, int nparts, int C, double* tmp, int* counter, double count,
                                 const float* gamma, const float* mean, const float* invstd, const float* scale,
                                 const float* shift, float* dgamma, float* dbeta, float* coef, hipStream_t s);
void bn_set_tuning(int grid_blocks, int unroll);
void bn_set_reverse(int mask);   // bit 0 / 1: forward / backward apply kernels walk their tensors from the end   // tuning hook for the streaming kernels
void bn_apply_launch(const void* x, const void* residual, void* y, long M, int C, int dtype,
                     const float* scale, const float* shift, int relu, hipStream_t s, uint8_t* mask = nullptr,
                     const float* res_scale = nullptr, const float* res_shift = nullptr);
// backward: dy masked by (y > 0) when relu; partial [nblk][2][C] = (sum dy, sum dy*xhat).  relu & 3 == 3: `y` is the
// bit mask bn_apply_launch wrote (one byte per 16-byte chunk), not the output tensor
void bn_bwd_reduce_launch(const void* dy, const void* x, const void* y, long M, int C, int dtype,
                          const float* mean, const float* invstd, int relu /* 0 | 1: mask from y | 2: from x */,
                          const float* scale, const float* shift, float* partial, int nblk, hipStream_t s);
// sums partials -> dgamma, dbeta (fp32 [C]) and the two per-channel means used by the apply pass
// relu(scale * x + shift) + per-workgroup column sums of the stored result (slab [parts][2][C], row 1 zero)
int bn_apply_colsum_parts(long M, int C, int dtype);
void bn_apply_colsum_launch(const void* x, void* y, long M, int C, int dtype, const float* scale, const float* shift,
                            float* part, int nparts, hipStream_t s);
void bn_bwd_finalize_launch(const float* partial, int nparts, int C, float* dgamma, float* dbeta,
                            hipStream_t s);
// fold partials -> dgamma, dbeta and (coef != nullptr) the apply coefficients [5][C], one launch
void bn_bwd_finalize_fused_launch(const float* partial, int nparts, int C, float* dgamma, float* dbeta,
                                  const float* mean, const float* invstd, const float* gamma, double count,
                                  float* coef, const float* scale, const float* shift, hipStream_t s);
// coefficients from (possibly all-reduced) sums; count_dev (optional) overrides count on the device
void bn_bwd_coeffs_launch(const float* dgamma, const float* dbeta, const float* mean, const float* invstd,
                          const float* gamma, double count, const float* count_dev, int C, float* coef,
                          const float* scale, const float* shift, hipStream_t s);
// dx = A*dy_m + B*x + K (coef rows 0..2), ReLU mask from y (relu 1) or from coef rows 3..4 and x (relu 2)
void bn_bwd_apply_launch(const void* dy, const void* x, const void* y, void* dx, void* dres, long M, int C,
                         int dtype, const float* coef, int relu, hipStream_t s);

// ------------------------------------------------------------------------------ bnfold.hip
// BatchNorm backward folded into the producing 1x1 convolution's backward GEMMs (see the file header).
// raw = dy_m^T a [N][C] fp32, w = the 16-bit weights of the forward pass [N][ldw]; writes dgamma / dbeta (beta_*: accumulate
// factors), coef [3][N] = (A, B, K), wt_scaled [C][ldwt] = (A o W)^T and bw [N][C + 8] = (B o W | K | 0) in the compute dtype
void bnfold_coeffs_launch(const float* raw, const void* w, int ldw, const float* dbeta_sum, const float* mean,
                          const float* invstd, const float* gamma, double count, int N, int C, float* dgamma_out,
                          float beta_g, float* dbeta_out, float beta_b, float* coef, void* wt_scaled, int ldwt, void* bw, int dtype,
                          hipStream_t s);
// out[n][c] = beta * out + A[n] raw[n][c] + B[n] wg[n][c] + K[n] cs[c]
// forward statistics of z = a W^T from t = W (a^T a) and cs = colsum(a) (no pass over z): coef[4][N] = mean, invstd, scale, shift;
// wsc = scale o W in `dtype`; running statistics updated like bn_fold_finalize
void bnfold_fwd_launch(const float* t, const float* w, const float* cs, double count, const float* gamma, const float* beta,
                       float eps, float momentum, float* running_mean, float* running_var, long* nbt, float* coef, void* wsc,
                       int N, int C, int dtype, hipStream_t s);
void bnfold_wgrad_launch(const float* raw, const float* wg, const float* coef, const float* cs, float* out, int N, int C,
                         float beta, hipStream_t s);

// ------------------------------------------------------------------------------ pool.hip
// idx (optional, uint8 per output element): tap code r*kw+q of the first arg-max, consumed by backward
void maxpool_fwd_launch(const void* x, void* y, uint8_t* idx, int NB, int IH, int IW, int C, int OH, int OW,
                        int kh, int kw, int sh, int sw, int ph, int pw, int dtype, hipStream_t s);
void maxpool_bwd_launch(const uint8_t* idx, const void* dy, void* dx, int NB, int IH, int IW, int C,
                        int OH, int OW, int kh, int kw, int sh, int sw, int ph, int pw, int dtype,
                        hipStream_t s);
// stem fusion: y = maxpool(relu(scale*x + shift)) without materialising the normalised tensor; backward gathers the
// pooled gradient through the arg-max codes inside the BatchNorm reduce / apply passes (bn = [mean|invstd|scale|shift])
int bn_relu_maxpool_partial_rows(long in_chunks);
void bn_relu_maxpool_fwd_launch(const void* x, void* y, uint8_t* idx, const float* scale, const float* shift, int NB,
                                int IH, int IW, int C, int OH, int OW, int kh, int kw, int sh, int sw, int ph, int pw,
                                int dtype, hipStream_t s);
void bn_relu_maxpool_bwd_reduce_launch(const uint8_t* idx, const void* dyp, const void* x, const float* bn,
                                       float* partial, int nblk, int NB, int IH, int IW, int C, int OH, int OW, int kh,
                                       int kw, int sh, int sw, int ph, int pw, int dtype, hipStream_t s);
void bn_relu_maxpool_bwd_apply_launch(const uint8_t* idx, const void* dyp, const void* x, const float* bn,
                                      const float* coef, void* dx, int NB, int IH, int IW, int C, int OH, int OW,
                                      int kh, int kw, int sh, int sw, int ph, int pw, int dtype, hipStream_t s);
void avgpool_fwd_launch(const void* x, void* y, int NB, int IH, int IW, int C, int OH, int OW,
                        int dtype, hipStream_t s);
// fixed window with zero padding counted in the divisor; backward != 0: src = dy, dst = dx
void avgpool_win_launch(const void* src, void* dst, int backward, int NB, int IH, int IW, int C, int OH, int OW, int kh,
                        int kw, int sh, int sw, int ph, int pw, int dtype, hipStream_t s);
void avgpool_bwd_launch(const void* dy, void* dx, int NB, int IH, int IW, int C, int OH, int OW,
                        int dtype, hipStream_t s);

// ------------------------------------------------------------------------------ loss.hip
// logits [N][C] (dtype), target int64; loss_sum[0] += mean loss contributions; dlogits optional.
// rowbuf: fp32 workspace [3][N]; metrics (optional): {loss_sum, correct@1, correct@5, rows} accumulated
void xent_fwd_bwd_launch(const void* logits, const long* target, int N, int C, int ldl, int dtype,
                         float* loss_out, void* dlogits, float* rowbuf, float* metrics, hipStream_t s);
void xent_bwd_scale_launch(void* dlogits, long n, int dtype, const float* gout, hipStream_t s);

// ------------------------------------------------------------------------------ elementwise.hip
void normalize_u8_launch(const uint8_t* x, void* y, long npix, int cin, int cout, int hw, int nchw,
                         const float* mean255, const float* inv_std255, int dtype, hipStream_t s);
// images -> space-to-depth [N][H/2][padl + W/2 + padr][16] (in_dtype < 0: uint8 source, normalised with mean255 /
// inv_std255); the pad pixels are zero
void to_s2d_launch(const void* x, void* y, int NB, int H2, int W2, int nchw, int in_dtype, const float* mean255,
                   const float* inv_std255, int out_dtype, hipStream_t s, int padl = 0, int padr = 0);
void nchw_to_nhwc_launch(const void* x, void* y, int NB, int C, int HW, int cpad, int in_dtype,
                         int out_dtype, hipStream_t s);
void relu_fwd_launch(const void* x, void* y, long n, int dtype, hipStream_t s);
void relu_bwd_launch(const void* y, const void* dy, void* dx, long n, int dtype, hipStream_t s);
void add_relu_fwd_launch(const void* a, const void* b, void* y, long n, int dtype, hipStream_t s);
// direct depthwise convolution (groups == channels); w = the parameter's memory [C][KH][KW] (fp32 or compute dtype)
// 256 x 256 multi-phase GEMM (gemm8.hip): C[M][N] = A[M][K] * B[N][K]^T (+ bias) (+ ReLU); 16-bit, K % 64 == 0, N % 8 == 0
void gemm8_launch(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda, long ldb,
                  long ldc, int relu, int dtype, hipStream_t stream);
void dw_set_sliding(int on);    // 0: window-per-pixel depthwise kernels only
int dw_conv_rows(long npix, int C, int dtype);
void dw_conv_launch(const void* in, const void* w, int w_dtype, const float* bias, void* out, float* stats, int backward,
                    int NB, int IH, int IW, int C, int OH, int OW, int KH, int KW, int sh, int sw, int ph, int pw, int relu,
                    int dtype, hipStream_t s);
int dw_wgrad_rows(long npix, int C, int KH, int dtype);
void dw_wgrad_launch(const void* dy, const void* x, float* partial, int NB, int IH, int IW, int C, int OH, int OW, int KH,
                     int KW, int sh, int sw, int ph, int pw, int dtype, hipStream_t s);
// LayerNorm over the last axis of [rows][C]; backward folds dgamma / dbeta from ln_partial_rows() partial rows
int ln_partial_rows(long rows, int C, int dtype);
void ln_fwd_launch(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, long rows, int C,
                   float eps, int dtype, hipStream_t s);
void ln_bwd_launch(const void* dy, const void* x, const float* w, const float* mean, const float* rstd, void* dx,
                   float* partial, float* dgamma, float* dbeta, float beta_w, float beta_b, long rows, int C, int dtype,
                   hipStream_t s);
void add_fwd_launch(const void* a, const void* b, void* y, long n, int dtype, hipStream_t s);
// act: 1 sigmoid, 2 SiLU, 3 hard-sigmoid, 4 hard-swish, 5 GELU (erf); backward recomputes the slope from the input x
void act_fwd_launch(const void* x, void* y, long n, int act, int dtype, hipStream_t s);
void act_bwd_launch(const void* x, const void* dy, void* dx, long n, int act, int dtype, hipStream_t s);
// y[n][p][c] = x[n][p][c] * sc[n][c] (squeeze-excitation gate / stochastic depth); backward: dx and ds = sum_p dy * x
void se_scale_fwd_launch(const void* x, const float* sc, void* y, long NB, int HW, int C, int dtype, hipStream_t s,
                         const float* add = nullptr);   // y = x * sc[n][c] (+ add[n][c]); se_scale_bwd: dx may be null
void se_scale_bwd_launch(const void* dy, const void* x, const float* sc, void* dx, float* ds, long NB, int HW, int C,
                         int dtype, hipStream_t s);
// counter (device, optional): RNG stream position read by the kernel and advanced by n afterwards
void dropout_fwd_launch(const void* x, void* y, uint8_t* mask, long n, float p, uint64_t seed,
                        uint64_t offset, uint64_t* counter, int dtype, hipStream_t s);
void dropout_bwd_launch(const void* dy, const uint8_t* mask, void* dx, long n, float p, int dtype,
                        hipStream_t s);
void cast_launch(const void* x, int in_dtype, void* y, int out_dtype, long n, hipStream_t s);
// [K][T][C] -> [C][T][K]  (conv weight KRSC -> dgrad operand CRSK); cpad zero-pads C of the source view
void transpose_ktc_launch(const void* w, void* wt, int K, int T, int C, int in_dtype, int out_dtype,
                          hipStream_t s);
void axpby_launch(const void* x, int x_dtype, float* y, long n, float alpha, float beta, hipStream_t s);
void fill_launch(void* p, long nbytes, int value, hipStream_t s);
// ShuffleNetV2: (o0, o1) = split(channel_shuffle(cat(a, b), groups = 2)) on [rows][P] tensors with bf real channels
// (the rest zero padding); backward != 0: the inverse permutation (a, b = gradients of o0, o1 -> gradients of a, b)
// NHWC pixels (n, oy * s, ox * s) gathered into a dense tensor
void subsample_launch(const void* src, void* dst, long NB, int IH, int IW, long row_bytes, int s_, hipStream_t s);
void shuffle_pair_launch(const void* a, const void* b, void* o0, void* o1, long rows, int bf, int P, int backward,
                         int dtype, hipStream_t s);
// strided 2-D copy in 16-byte chunks: channel concat / split of NHWC activations
void copy_rows_launch(const void* src, long ld_src_bytes, void* dst, long ld_dst_bytes, long rows, long row_bytes,
                      hipStream_t s);

// ------------------------------------------------------------------------------ attention.hip
// O = softmax(scale[h] * Q K^T + bias[b % nbias][h]) V on the packed projection qkv [B][S][3][H][D] (bf16 / f16);
// out [B][S][H][D], lse [B][H][S] fp32.  bias: optional fp32 [nbias][H][S][S].
int attn_max_tokens(int D);
// x [B][H][W][C] <-> windows [B*nh*nw][wh*ww][C] (zero padding up to nh*wh x nw*ww, cyclic shift (sh, sw) for Swin;
// grid_mode: MaxViT's grid partition, token (i, j) of window (gh, gw) = pixel (i*nh + gh, j*nw + gw))
void window_partition_launch(const void* src, void* dst, int reverse, int B, int H, int W, int C, int wh, int ww, int nh,
                             int nw, int sh, int sw, int grid_mode, int dtype, hipStream_t s);
// Swin V2 cosine attention: q <- mult[h] * q / |q|, k <- k / |k| on the packed [tokens][3][H][D] projection (v copied);
// inv [tokens][2][H] keeps the inverse norms; backward also writes dmult_part [tokens][H] (sum over tokens = d mult)
void qk_normalize_launch(const void* x, const void* dy, void* out, const float* mult, float* inv, float* dmult_part,
                         long tokens, int H, int D, int backward, int dtype, hipStream_t s);
void attn_fwd_launch(const void* qkv, void* out, float* lse, const float* bias, int nbias, const float* scale, int B,
                     int H, int S, int D, int dtype, hipStream_t s);
// dqkv [B][S][3][H][D]; delta: fp32 scratch [B][H][S]; dbias (optional, S <= 64): partial slabs [nsplit][H][S][S] of the
// gradient of the pre-softmax logits summed over the batch (fold with slab_reduce_launch)
int attn_dbias_splits(int B, int H);
void attn_bwd_launch(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                     const float* bias, int nbias, const float* scale, void* dqkv, float* dbias, int nsplit, int B, int H,
                     int S, int D, int dtype, hipStream_t s);

// ------------------------------------------------------------------------------ se_mlp.hip
// squeeze-excitation MLP on pooled [N][C] fp32 rows: s = gate(W2 act(W1 p + b1) + b2); act 0 ReLU / 1 SiLU, gate 0 sigmoid /
// 1 hard-sigmoid.  W1 [S][C], W2 [C][S], w2t = W2^T [S][C].  Backward: per-sample gradients + (dw1 != nullptr) parameter
// gradients summed over the batch in a fixed order.
void se_mlp_fwd_launch(const float* p, const float* w1, const float* b1, const float* w2, const float* b2, float* z1,
                       float* h, float* z2, float* s, int N, int C, int S, int act, int gate, hipStream_t st);
void se_mlp_bwd_launch(const float* ds, const float* s, const float* z2, const float* z1, const float* p, const float* h,
                       const float* w1, const float* w2t, float* dz2, float* dz1, float* dp, float* dw1, float* db1,
                       float* dw2, float* db2, int N, int C, int S, int act, int gate, float inv_hw, hipStream_t st);

// ------------------------------------------------------------------------------ optim.hip
struct SgdTensor {   // one entry per parameter tensor (device table)
  float* p;          // fp32 master / parameter
  const void* g;     // gradient (g_dtype)
  float* buf;        // momentum buffer
  void* lp;          // optional low-precision copy (lp_dtype) or nullptr
  long n;
  long chunk0;       // first chunk index of this tensor in the flattened chunk space
};
void sgd_multi_tensor_launch(const SgdTensor* table, int ntensors, long total_chunks, int g_dtype,
                             int lp_dtype, const float* hyper /*lr, momentum, wd, dampening, nesterov, first*/,
                             const float* inv_scale, const float* found_inf, hipStream_t s);
int sgd_chunk_elems();
void check_finite_flat_launch(const void* g, long n, int dtype, float* found_inf, hipStream_t s);
void scaler_update_launch(float* scale, float* inv_scale, int* good_steps, float* found_inf,
                          int growth_interval, float growth, float backoff, float max_scale,
                          float min_scale, hipStream_t s);

}  // namespace dpa
