// BatchNorm backward folded into the backward GEMMs of the 1x1 convolution that feeds it ("bn fold").
//
// For z = a W^T (1x1 convolution, W: [N][C]) followed by a training-mode BatchNorm, the BatchNorm backward
//     dz = A * dy_m + B * z + K          (A, B, K per channel n; dy_m = dy through the ReLU mask)
// is linear in dy_m and z, and z is linear in a, so the convolution's gradients need neither dz nor z:
//     da = dy_m (diag(A) W) + a (W^T diag(B) W) + K^T W
//     dW = diag(A) (dy_m^T a) + diag(B) W (a^T a) + K (x) colsum(a)
// and the BatchNorm's own reduction  sum_m dy_m z = rowsum(W o (dy_m^T a))  falls out of the weight gradient.  The
// activation-sized passes that disappear: the BatchNorm-backward apply pass (read dy, read z, write dz), the read of z
// in the producer's fused reduction, and z itself no longer has to be kept for backward.  What is added are GEMMs on
// the NARROW tensor a (a quarter of the channels in a bottleneck) and the two small kernels below.
// (hip_ops._ConvFn / _BNFn "bn fold"; replaces cuDNN's batch-norm backward + convolution backward pair of
// loss.backward() in the reference, nd_imagenet.py:302 -- SURVEY.md N-K2, N-K3, N-K5.)
#include "../common.h"
#include "kernels.h"
#include <stdexcept>

namespace dpa {
namespace {

// One workgroup per output channel n.  raw = dy_m^T a [N][C] (fp32), w = the 16-bit weights the forward pass used.
template <typename T>
__global__ __launch_bounds__(256) void bnfold_coeffs_kernel(
    const float* __restrict__ raw, const T* __restrict__ w, int ldw, const float* __restrict__ dbeta_sum,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma, float inv_count,
    int N, int C, float* __restrict__ dgamma_out, float beta_g, float* __restrict__ dbeta_out, float beta_b,
    float* __restrict__ coef, T* __restrict__ wt_scaled, int ldwt, T* __restrict__ bw) {
  __shared__ float red[4];
  __shared__ float abk[3];
  const int n = blockIdx.x, tid = threadIdx.x;
  float q = 0.f;
  for (int c = tid; c < C; c += 256) q += to_float(w[(size_t)n * ldw + c]) * raw[(size_t)n * C + c];
  q = wave_sum(q);
  if ((tid & 63) == 0) red[tid >> 6] = q;
  __syncthreads();
  if (tid == 0) {
    const float Q = (red[0] + red[1]) + (red[2] + red[3]);          // sum_m dy_m z
    const float db = dbeta_sum[n], is = invstd[n], mu = mean[n], g = gamma ? gamma[n] : 1.f;
    const float dg = is * (Q - mu * db);                             // sum_m dy_m xhat
    const float A = g * is;
    const float B = -A * is * dg * inv_count;
    const float K = -A * db * inv_count - B * mu;
    dgamma_out[n] = (beta_g != 0.f ? beta_g * dgamma_out[n] : 0.f) + dg;
    dbeta_out[n] = (beta_b != 0.f ? beta_b * dbeta_out[n] : 0.f) + db;
    coef[n] = A;
    coef[N + n] = B;
    coef[2 * N + n] = K;
    abk[0] = A; abk[1] = B; abk[2] = K;
  }
  __syncthreads();
  const float A = abk[0], B = abk[1], K = abk[2];
  for (int c = tid; c < C + 8; c += 256) {
    if (c < C) {
      const float wv = to_float(w[(size_t)n * ldw + c]);
      wt_scaled[(size_t)c * ldwt + n] = from_float<T>(A * wv);
      bw[(size_t)n * (C + 8) + c] = from_float<T>(B * wv);
    } else {
      bw[(size_t)n * (C + 8) + c] = from_float<T>(c == C ? K : 0.f);
    }
  }
}

// dW[n][c] = beta * dW + A[n] * raw[n][c] + B[n] * wg[n][c] + K[n] * cs[c]     (wg = W (a^T a), cs = colsum(a))
__global__ __launch_bounds__(256) void bnfold_wgrad_kernel(const float* __restrict__ raw, const float* __restrict__ wg,
                                                           const float* __restrict__ coef, const float* __restrict__ cs,
                                                           float* __restrict__ out, int N, int C, float beta) {
  const long total = (long)N * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i / C), c = (int)(i - (long)n * C);
    float v = fmaf(coef[n], raw[i], fmaf(coef[N + n], wg[i], coef[2 * N + n] * cs[c]));
    if (beta != 0.f) v += beta * out[i];
    out[i] = v;
  }
}

// ---- forward: BatchNorm statistics of z = a W^T WITHOUT z ------------------------------------------------------------
// mean_z[n] = w_n . colsum(a) / M and E[z^2][n] = w_n^T (a^T a) w_n / M: with t = W (a^T a) (an fp32 GEMM on the small
// matrices) both are row dot products.  The convolution then runs with the scaled weights scale o W and the shift as its
// bias and applies residual + ReLU in its epilogue: the pre-BatchNorm tensor z is never written (hip_ops._ConvBnResReluFn).
// Sums in double: E[z^2] - mean^2 cancels when |mean| >> std.
template <typename T>
__global__ __launch_bounds__(256) void bnfold_fwd_kernel(const float* __restrict__ t, const float* __restrict__ w,
                                                        const float* __restrict__ cs, double count,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, float momentum, float* running_mean, float* running_var,
                                                        long* nbt, float* __restrict__ coef, T* __restrict__ wsc, int N, int C) {
  __shared__ double red[2][256];
  __shared__ float sc_s;
  const int n = blockIdx.x, tid = threadIdx.x;
  double sm = 0.0, sq = 0.0;
  for (int c = tid; c < C; c += 256) {
    const double wf = (double)w[(size_t)n * C + c];
    sm += wf * (double)cs[c];
    sq += wf * (double)t[(size_t)n * C + c];
  }
  red[0][tid] = sm;
  red[1][tid] = sq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      red[0][tid] += red[0][tid + o];
      red[1][tid] += red[1][tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double mean = red[0][0] / count;
    double var = red[1][0] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[n] : 1.f, b = beta ? beta[n] : 0.f;
    coef[n] = (float)mean;
    coef[N + n] = is;
    coef[2 * N + n] = g * is;
    coef[3 * N + n] = b - (float)mean * g * is;
    sc_s = g * is;
    if (running_mean != nullptr) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * (float)mean;
      running_var[n] = (1.f - momentum) * running_var[n] + momentum * (float)unbiased;
    }
    if (n == 0 && nbt != nullptr) *nbt += 1;
  }
  __syncthreads();
  const float sc = sc_s;
  for (int c = tid; c < C; c += 256) wsc[(size_t)n * C + c] = from_float<T>(sc * w[(size_t)n * C + c]);
}

}  // namespace

void bnfold_coeffs_launch(const float* raw, const void* w, int ldw, const float* dbeta_sum, const float* mean,
                          const float* invstd, const float* gamma, double count, int N, int C, float* dgamma_out,
                          float beta_g, float* dbeta_out, float beta_b, float* coef, void* wt_scaled, int ldwt, void* bw, int dtype,
                          hipStream_t s) {
  if (N <= 0 || C <= 0) return;
  const float inv = (float)(1.0 / count);
  if (dtype == kBF16)
    hipLaunchKernelGGL(bnfold_coeffs_kernel<bf16_t>, dim3(N), dim3(256), 0, s, raw, reinterpret_cast<const bf16_t*>(w), ldw,
                       dbeta_sum, mean, invstd, gamma, inv, N, C, dgamma_out, beta_g, dbeta_out, beta_b, coef,
                       reinterpret_cast<bf16_t*>(wt_scaled), ldwt, reinterpret_cast<bf16_t*>(bw));
  else if (dtype == kF16)
    hipLaunchKernelGGL(bnfold_coeffs_kernel<f16_t>, dim3(N), dim3(256), 0, s, raw, reinterpret_cast<const f16_t*>(w), ldw,
                       dbeta_sum, mean, invstd, gamma, inv, N, C, dgamma_out, beta_g, dbeta_out, beta_b, coef,
                       reinterpret_cast<f16_t*>(wt_scaled), ldwt, reinterpret_cast<f16_t*>(bw));
  else
    throw std::invalid_argument("bnfold: 16-bit compute types only");
  DPA_CHECK_HIP(hipGetLastError());
}

void bnfold_wgrad_launch(const float* raw, const float* wg, const float* coef, const float* cs, float* out, int N, int C,
                         float beta, hipStream_t s) {
  if (N <= 0 || C <= 0) return;
  const long total = (long)N * C;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(bnfold_wgrad_kernel, dim3(blocks), dim3(256), 0, s, raw, wg, coef, cs, out, N, C, beta);
  DPA_CHECK_HIP(hipGetLastError());
}

void bnfold_fwd_launch(const float* t, const float* w, const float* cs, double count, const float* gamma, const float* beta,
                       float eps, float momentum, float* running_mean, float* running_var, long* nbt, float* coef, void* wsc,
                       int N, int C, int dtype, hipStream_t s) {
  if (N <= 0 || C <= 0) return;
  if (dtype == kF32) throw std::invalid_argument("bnfold_fwd: 16-bit activations");
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bnfold_fwd_kernel<T>), dim3(N), dim3(256), 0, s, t, w, cs, count, gamma, beta,
                                               eps, momentum, running_mean, running_var, nbt, coef, (T*)wsc, N, C));
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
