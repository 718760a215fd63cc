// Fused multi-tensor SGD (momentum, weight decay, nesterov) + AMP loss-scaler kernels (gfx950).
//
// ONE launch walks a device-resident tensor table: each workgroup binary-searches its tensor from a
// flat chunk index, reads the gradient (fp32 / bf16 / fp16) times 1/loss_scale, applies
// torch.optim.SGD's exact update (g += wd*p; first step buf = g, else buf = mu*buf + (1-damp)*g;
// p -= lr*(nesterov ? g + mu*buf : buf)) on the fp32 master and momentum, and writes the
// low-precision compute copy the next forward consumes.  lr / momentum / wd live in a device
// array (per-iteration LR schedules cost no re-capture under hipGraph replay) and the whole update
// is predicated on the loss scaler's found_inf flag without a host sync.
//
// Replaces torch.optim.SGD.step + Apex amp_C multi_tensor_{scale,axpby} + LossScaler
// (reference nd_imagenet.py:303, imagenet_ddp_apex.py:380-383) -- SURVEY.md N-K13, N-K14.
#include "../common.h"
#include "kernels.h"

namespace dpa {
namespace {

constexpr int kChunk = 2048;  // elements per workgroup chunk (256 threads x 8)

template <typename TG, typename TL>
__global__ __launch_bounds__(256) void sgd_kernel(const SgdTensor* __restrict__ table, int ntensors,
                                                  const float* __restrict__ hyper, const float* __restrict__ inv_scale,
                                                  const float* __restrict__ found_inf) {
  if (found_inf != nullptr && found_inf[0] > 0.f) return;   // overflow: skip the whole step
  const long chunk = blockIdx.x;
  int lo = 0, hi = ntensors - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].chunk0 <= chunk) lo = mid; else hi = mid - 1;
  }
  const SgdTensor t = table[lo];
  const long base = (chunk - t.chunk0) * kChunk;
  const float lr = hyper[0], mu = hyper[1], wd = hyper[2], damp = hyper[3];
  const bool nesterov = hyper[4] != 0.f, first = hyper[5] != 0.f;
  const float gs = inv_scale != nullptr ? inv_scale[0] : 1.f;
  const TG* __restrict__ g = reinterpret_cast<const TG*>(t.g);
  TL* __restrict__ lp = reinterpret_cast<TL*>(t.lp);
#pragma unroll
  for (int k = 0; k < kChunk / 256; ++k) {
    const long i = base + k * 256 + threadIdx.x;
    if (i >= t.n) break;
    float p = t.p[i];
    float gr = to_float(g[i]) * gs + wd * p;
    if (mu != 0.f) {
      float b = first ? gr : mu * t.buf[i] + (1.f - damp) * gr;
      t.buf[i] = b;
      gr = nesterov ? gr + mu * b : b;
    }
    p -= lr * gr;
    t.p[i] = p;
    if (lp != nullptr) lp[i] = from_float<TL>(p);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void check_finite_kernel(const T* __restrict__ g, long n, float* found_inf) {
  const long stride = (long)gridDim.x * blockDim.x;
  bool bad = false;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = to_float(g[i]);
    bad |= !(fabsf(v) <= 3.402823466e38f);   // inf or nan
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) found_inf[0] = 1.f;  // idempotent plain store
}

__global__ void scaler_update_kernel(float* scale, float* inv_scale, int* good_steps, float* found_inf,
                                     int growth_interval, float growth, float backoff, float max_scale,
                                     float min_scale) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = scale[0];
  int good = good_steps[0];
  if (found_inf[0] > 0.f) {
    s *= backoff;
    good = 0;
  } else {
    good += 1;
    if (good >= growth_interval) {
      s *= growth;
      good = 0;
    }
  }
  s = fminf(fmaxf(s, min_scale), max_scale);
  scale[0] = s;
  inv_scale[0] = 1.f / s;
  good_steps[0] = good;
  found_inf[0] = 0.f;
}

}  // namespace

void sgd_multi_tensor_launch(const SgdTensor* table, int ntensors, long total_chunks, int g_dtype, int lp_dtype,
                             const float* hyper, const float* inv_scale, const float* found_inf, hipStream_t s) {
  if (ntensors <= 0 || total_chunks <= 0) return;
  dim3 grid((unsigned)total_chunks), block(256);
#define DPA_SGD(TG, TL) hipLaunchKernelGGL((sgd_kernel<TG, TL>), grid, block, 0, s, table, ntensors, hyper, inv_scale, found_inf)
  if (g_dtype == kF32 && lp_dtype == kBF16) DPA_SGD(float, bf16_t);
  else if (g_dtype == kF32 && lp_dtype == kF16) DPA_SGD(float, f16_t);
  else if (g_dtype == kF32) DPA_SGD(float, float);
  else if (g_dtype == kBF16 && lp_dtype == kBF16) DPA_SGD(bf16_t, bf16_t);
  else if (g_dtype == kF16 && lp_dtype == kF16) DPA_SGD(f16_t, f16_t);
  else throw std::invalid_argument("sgd: unsupported gradient / copy dtype combination");
#undef DPA_SGD
  DPA_CHECK_HIP(hipGetLastError());
}

int sgd_chunk_elems() { return kChunk; }

void check_finite_flat_launch(const void* g, long n, int dtype, float* found_inf, hipStream_t s) {
  if (n <= 0) return;
  long b = (n + 255) / 256;
  if (b > 1024) b = 1024;
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(check_finite_kernel<T>, dim3((unsigned)b), dim3(256), 0, s, (const T*)g,
                                               n, found_inf));
  DPA_CHECK_HIP(hipGetLastError());
}

void scaler_update_launch(float* scale, float* inv_scale, int* good_steps, float* found_inf, int growth_interval,
                          float growth, float backoff, float max_scale, float min_scale, hipStream_t s) {
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, s, scale, inv_scale, good_steps, found_inf,
                     growth_interval, growth, backoff, max_scale, min_scale);
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
