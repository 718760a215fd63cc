// Fused softmax cross-entropy forward + backward + top-1/top-5 bookkeeping (gfx950).
// One workgroup per row: online max / sum-exp in fp32, loss_i = lse - z_t, and in the same pass
// dz = (softmax - onehot) / N.  The target's rank (#{z_j > z_t}) gives top-k correctness without a
// sort.  Per-row results are folded by a one-block kernel (fixed order, deterministic) into the
// mean loss and a packed device-side metrics record {loss_sum, correct@1, correct@5, rows}: the
// training loop reads it every print_freq iterations instead of `.item()` every step
// (reference nd_imagenet.py:296, imagenet_ddp_apex.py:385-406) -- SURVEY.md N-K11, N-K12, N-C8.
#include "../common.h"
#include "kernels.h"

namespace dpa {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void xent_row_kernel(const T* __restrict__ logits, const long* __restrict__ target,
                                                       int N, int C, int ldl, T* __restrict__ dlogits,
                                                       float* __restrict__ rowbuf) {
  __shared__ float sred[2][4];
  __shared__ int sint[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* z = logits + (size_t)row * ldl;
  long tl = target[row];
  const bool tvalid = tl >= 0 && tl < C;
  const int t = tvalid ? (int)tl : 0;
  const float zt = to_float(z[t]);
  float mx = -INFINITY;
  for (int c = tid; c < C; c += 256) mx = fmaxf(mx, to_float(z[c]));
  mx = wave_max(mx);
  if (lane == 0) sred[0][wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sred[0][0], sred[0][1]), fmaxf(sred[0][2], sred[0][3]));
  float se = 0.f;
  int gt = 0;
  for (int c = tid; c < C; c += 256) {
    const float v = to_float(z[c]);
    se += __expf(v - mx);
    gt += (v > zt) ? 1 : 0;
  }
  se = wave_sum(se);
  float gtf = wave_sum((float)gt);
  if (lane == 0) { sred[1][wave] = se; sint[wave] = (int)gtf; }
  __syncthreads();
  se = sred[1][0] + sred[1][1] + sred[1][2] + sred[1][3];
  const int rank = sint[0] + sint[1] + sint[2] + sint[3];
  const float lse = mx + __logf(se);
  if (tid == 0) {
    rowbuf[row] = tvalid ? (lse - zt) : 0.f;
    rowbuf[N + row] = (tvalid && rank < 1) ? 1.f : 0.f;
    rowbuf[2 * N + row] = (tvalid && rank < 5) ? 1.f : 0.f;
  }
  if (dlogits != nullptr) {
    T* dz = dlogits + (size_t)row * ldl;
    const float invn = 1.f / (float)N, inv_se = 1.f / se;
    for (int c = tid; c < C; c += 256) {
      float p = __expf(to_float(z[c]) - mx) * inv_se;
      if (c == t) p -= 1.f;
      dz[c] = from_float<T>(tvalid ? p * invn : 0.f);
    }
    for (int c = C + tid; c < ldl; c += 256) dz[c] = from_float<T>(0.f);
  }
}

__global__ __launch_bounds__(256) void xent_fold_kernel(const float* __restrict__ rowbuf, int N, float* loss_out,
                                                        float* metrics) {
  __shared__ float s[3][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float a[3] = {0.f, 0.f, 0.f};
  for (int i = tid; i < N; i += 256)
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] += rowbuf[k * N + i];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    a[k] = wave_sum(a[k]);
    if (lane == 0) s[k][wave] = a[k];
  }
  __syncthreads();
  if (tid == 0) {
    const float l = s[0][0] + s[0][1] + s[0][2] + s[0][3];
    loss_out[0] = l / (float)N;
    if (metrics != nullptr) {
      metrics[0] += l;
      metrics[1] += s[1][0] + s[1][1] + s[1][2] + s[1][3];
      metrics[2] += s[2][0] + s[2][1] + s[2][2] + s[2][3];
      metrics[3] += (float)N;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(T* x, long n, const float* __restrict__ g) {
  const float s = g[0];
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    x[i] = from_float<T>(to_float(x[i]) * s);
}

}  // namespace

void xent_fwd_bwd_launch(const void* logits, const long* target, int N, int C, int ldl, int dtype, float* loss_out,
                         void* dlogits, float* rowbuf, float* metrics, hipStream_t s) {
  if (N <= 0) return;
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(xent_row_kernel<T>, dim3(N), dim3(256), 0, s, (const T*)logits, target,
                                               N, C, ldl, (T*)dlogits, rowbuf));
  hipLaunchKernelGGL(xent_fold_kernel, dim3(1), dim3(256), 0, s, rowbuf, N, loss_out, metrics);
  DPA_CHECK_HIP(hipGetLastError());
}

void xent_bwd_scale_launch(void* dlogits, long n, int dtype, const float* gout, hipStream_t s) {
  if (n <= 0) return;
  long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(scale_kernel<T>, dim3((unsigned)b), dim3(256), 0, s, (T*)dlogits, n, gout));
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
