// Max / adaptive-average pooling for NHWC activations (gfx950), 16-byte channel vectors.
// Backward passes are gathers (one thread per INPUT chunk): deterministic, no atomics, and the
// max-pool backward recomputes each covering window's first arg-max instead of storing indices.
// Replaces ATen/cuDNN pooling in the reference models (ResNet stem 3x3 s2 p1, AlexNet 3x3 s2,
// VGG 2x2 s2, AdaptiveAvgPool2d) -- SURVEY.md N-K8, N-K9.
#include "../common.h"
#include "kernels.h"

namespace dpa {
namespace {

struct PoolGeom {
  int NB, IH, IW, C, OH, OW, kh, kw, sh, sw, ph, pw;
};

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ idx, PoolGeom g, long total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int cc = g.C / VEC;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cc);
    long pix = i / cc;
    const int ox = (int)(pix % g.OW); pix /= g.OW;
    const int oy = (int)(pix % g.OH);
    const int n = (int)(pix / g.OH);
    float m[VEC];
    int am[VEC];   // first arg-max in scan order (torch semantics), as the tap code r*kw + q
#pragma unroll
    for (int j = 0; j < VEC; ++j) { m[j] = -INFINITY; am[j] = 255; }
    for (int r = 0; r < g.kh; ++r) {
      const int iy = oy * g.sh - g.ph + r;
      if ((unsigned)iy >= (unsigned)g.IH) continue;
      for (int q = 0; q < g.kw; ++q) {
        const int ix = ox * g.sw - g.pw + q;
        if ((unsigned)ix >= (unsigned)g.IW) continue;
        Vec16<T> v;
        v.load(x + (((size_t)n * g.IH + iy) * g.IW + ix) * g.C + c * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float f = v.get(j);
          if (f > m[j] || f != f || am[j] == 255) { m[j] = f; am[j] = r * g.kw + q; }
        }
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.set(j, m[j]);
    o.store(y + i * VEC);
    if (idx != nullptr) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) idx[i * VEC + j] = (uint8_t)am[j];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const uint8_t* __restrict__ idx, const T* __restrict__ dy,
                                                          T* __restrict__ dx, PoolGeom g, long total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int cc = g.C / VEC;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cc);
    long pix = i / cc;
    const int ix = (int)(pix % g.IW); pix /= g.IW;
    const int iy = (int)(pix % g.IH);
    const int n = (int)(pix / g.IH);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    // output windows that contain (iy, ix)
    int oy_lo = (iy + g.ph - g.kh + g.sh) / g.sh; if (iy + g.ph - g.kh + 1 <= 0) oy_lo = 0;
    int ox_lo = (ix + g.pw - g.kw + g.sw) / g.sw; if (ix + g.pw - g.kw + 1 <= 0) ox_lo = 0;
    const int oy_hi = min(g.OH - 1, (iy + g.ph) / g.sh);
    const int ox_hi = min(g.OW - 1, (ix + g.pw) / g.sw);
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const int code = (iy - (oy * g.sh - g.ph)) * g.kw + (ix - (ox * g.sw - g.pw));
        const size_t o = (((size_t)n * g.OH + oy) * g.OW + ox) * g.C + c * VEC;
        Vec16<T> d;
        d.load(dy + o);
        uint8_t am[VEC];
        if (VEC == 8) *reinterpret_cast<uint2*>(am) = *reinterpret_cast<const uint2*>(idx + o);
        else *reinterpret_cast<uint32_t*>(am) = *reinterpret_cast<const uint32_t*>(idx + o);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (am[j] == code) acc[j] += d.get(j);
      }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.set(j, acc[j]);
    o.store(dx + i * VEC);
  }
}

__device__ __forceinline__ int win_start(int o, int I, int O) { return (o * I) / O; }
__device__ __forceinline__ int win_end(int o, int I, int O) { return ((o + 1) * I + O - 1) / O; }

template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, PoolGeom g,
                                                          long total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int cc = g.C / VEC;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cc);
    long pix = i / cc;
    const int ox = (int)(pix % g.OW); pix /= g.OW;
    const int oy = (int)(pix % g.OH);
    const int n = (int)(pix / g.OH);
    const int y0 = win_start(oy, g.IH, g.OH), y1 = win_end(oy, g.IH, g.OH);
    const int x0 = win_start(ox, g.IW, g.OW), x1 = win_end(ox, g.IW, g.OW);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        Vec16<T> v;
        v.load(x + (((size_t)n * g.IH + yy) * g.IW + xx) * g.C + c * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += v.get(j);
      }
    const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.set(j, acc[j] * inv);
    o.store(y + i * VEC);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, PoolGeom g,
                                                          long total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int cc = g.C / VEC;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cc);
    long pix = i / cc;
    const int ix = (int)(pix % g.IW); pix /= g.IW;
    const int iy = (int)(pix % g.IH);
    const int n = (int)(pix / g.IH);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int oy = 0; oy < g.OH; ++oy) {
      const int y0 = win_start(oy, g.IH, g.OH), y1 = win_end(oy, g.IH, g.OH);
      if (iy < y0 || iy >= y1) continue;
      for (int ox = 0; ox < g.OW; ++ox) {
        const int x0 = win_start(ox, g.IW, g.OW), x1 = win_end(ox, g.IW, g.OW);
        if (ix < x0 || ix >= x1) continue;
        Vec16<T> d;
        d.load(dy + (((size_t)n * g.OH + oy) * g.OW + ox) * g.C + c * VEC);
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += d.get(j) * inv;
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.set(j, acc[j]);
    o.store(dx + i * VEC);
  }
}

inline int grid_for(long n) {
  long b = (n + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  return (int)(b < 1 ? 1 : b);
}
inline void check_c(int C, int dtype) {
  if (C % (dtype == kF32 ? 4 : 8) != 0) throw std::invalid_argument("pool: C must be a multiple of 16 bytes");
}

}  // namespace

void maxpool_fwd_launch(const void* x, void* y, uint8_t* idx, int NB, int IH, int IW, int C, int OH, int OW, int kh, int kw,
                        int sh, int sw, int ph, int pw, int dtype, hipStream_t s) {
  check_c(C, dtype);
  PoolGeom g{NB, IH, IW, C, OH, OW, kh, kw, sh, sw, ph, pw};
  const long total = (long)NB * OH * OW * (C / (dtype == kF32 ? 4 : 8));
  if (total <= 0) return;
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s,
                                               (const T*)x, (T*)y, idx, g, total));
  DPA_CHECK_HIP(hipGetLastError());
}
void maxpool_bwd_launch(const uint8_t* idx, const void* dy, void* dx, int NB, int IH, int IW, int C, int OH, int OW,
                        int kh, int kw, int sh, int sw, int ph, int pw, int dtype, hipStream_t s) {
  check_c(C, dtype);
  PoolGeom g{NB, IH, IW, C, OH, OW, kh, kw, sh, sw, ph, pw};
  const long total = (long)NB * IH * IW * (C / (dtype == kF32 ? 4 : 8));
  if (total <= 0) return;
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s,
                                               idx, (const T*)dy, (T*)dx, g, total));
  DPA_CHECK_HIP(hipGetLastError());
}
void avgpool_fwd_launch(const void* x, void* y, int NB, int IH, int IW, int C, int OH, int OW, int dtype,
                        hipStream_t s) {
  check_c(C, dtype);
  PoolGeom g{NB, IH, IW, C, OH, OW, 0, 0, 0, 0, 0, 0};
  const long total = (long)NB * OH * OW * (C / (dtype == kF32 ? 4 : 8));
  if (total <= 0) return;
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(avgpool_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s,
                                               (const T*)x, (T*)y, g, total));
  DPA_CHECK_HIP(hipGetLastError());
}
void avgpool_bwd_launch(const void* dy, void* dx, int NB, int IH, int IW, int C, int OH, int OW, int dtype,
                        hipStream_t s) {
  check_c(C, dtype);
  PoolGeom g{NB, IH, IW, C, OH, OW, 0, 0, 0, 0, 0, 0};
  const long total = (long)NB * IH * IW * (C / (dtype == kF32 ? 4 : 8));
  if (total <= 0) return;
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(avgpool_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s,
                                               (const T*)dy, (T*)dx, g, total));
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
