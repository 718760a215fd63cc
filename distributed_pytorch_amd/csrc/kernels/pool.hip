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

// Window scan of output (n, oy, ox), channels c*VEC..: max and FIRST arg-max in scan order (torch semantics) as
// the tap code r*kw + q.  KH, KW > 0: compile-time window, every tap loaded up front from a clamped address
// (all loads in flight at once; a runtime-bounded loop serialises one memory latency per tap).  BN: the values
// are relu(scale*x + shift) rounded to T, i.e. what a stand-alone BatchNorm pass would have stored.
template <typename T, int VEC, int KH, int KW, bool BN>
__device__ __forceinline__ void window_max(const T* __restrict__ x, const PoolGeom& g, uint32_t n, int oy, int ox, int c,
                                           const float (&sc)[VEC], const float (&sh)[VEC], float (&m)[VEC],
                                           int (&am)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; ++j) { m[j] = -INFINITY; am[j] = 255; }
  auto visit = [&](const Vec16<T>& v, int code) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float f = v.get(j);
      if (BN) f = to_float(from_float<T>(fmaxf(f * sc[j] + sh[j], 0.f)));
      if (f > m[j] || f != f || am[j] == 255) { m[j] = f; am[j] = code; }
    }
  };
  if constexpr (KH > 0) {
    Vec16<T> v[KH * KW];
    bool ok[KH * KW];
#pragma unroll
    for (int r = 0; r < KH; ++r)
#pragma unroll
      for (int q = 0; q < KW; ++q) {
        const int iy = oy * g.sh - g.ph + r, ix = ox * g.sw - g.pw + q;
        ok[r * KW + q] = (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
        const int cy = min(max(iy, 0), g.IH - 1), cx = min(max(ix, 0), g.IW - 1);
        v[r * KW + q].load(x + (((size_t)n * g.IH + cy) * g.IW + cx) * g.C + c * VEC);
      }
#pragma unroll
    for (int t = 0; t < KH * KW; ++t)
      if (ok[t]) visit(v[t], t);
  } else {
    for (int r = 0; r < g.kh; ++r) {
      const int iy = oy * g.sh - g.ph + r;
      if ((unsigned)iy >= (unsigned)g.IH) continue;
      for (int q = 0; q < g.kw; ++q) {
        const int ix = ox * g.sw - g.pw + q;
        if ((unsigned)ix >= (unsigned)g.IW) continue;
        Vec16<T> v;
        v.load(x + (((size_t)n * g.IH + iy) * g.IW + ix) * g.C + c * VEC);
        visit(v, r * g.kw + q);
      }
    }
  }
}

template <typename T, int KH, int KW>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ idx, PoolGeom g, long total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int cc = g.C / VEC;
  const long stride = (long)gridDim.x * blockDim.x;
  const float none[VEC] = {};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cc);
    long pix = i / cc;
    const int ox = (int)(pix % g.OW); pix /= g.OW;
    const int oy = (int)(pix % g.OH);
    const int n = (int)(pix / g.OH);
    float m[VEC];
    int am[VEC];
    window_max<T, VEC, KH, KW, false>(x, g, (uint32_t)n, oy, ox, c, none, none, m, am);
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

// Gradient arriving at input position (n, iy, ix), channels c*VEC..: sum of dy over the windows whose arg-max is
// here, in ascending (oy, ox) order.  NW > 0: at most NW windows per axis contain a pixel (kernel <= NW * stride);
// all NW*NW candidates are loaded up front from clamped addresses.  NW == 0: runtime-bounded loops.
template <typename T, int VEC, int NW>
__device__ __forceinline__ void pooled_grad(const uint8_t* __restrict__ idx, const T* __restrict__ dy,
                                            const PoolGeom& g, uint32_t n, int iy, int ix, int c, float (&acc)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  auto add = [&](const Vec16<T>& d, const uint8_t (&am)[VEC], int code) {
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      if (am[j] == code) acc[j] += d.get(j);
  };
  auto load_codes = [&](size_t o, uint8_t (&am)[VEC]) {
    if (VEC == 8) *reinterpret_cast<uint2*>(am) = *reinterpret_cast<const uint2*>(idx + o);
    else *reinterpret_cast<uint32_t*>(am) = *reinterpret_cast<const uint32_t*>(idx + o);
  };
  if constexpr (NW > 0) {
    const int oyh = (iy + g.ph) / g.sh, oxh = (ix + g.pw) / g.sw;      // last window that can contain the pixel
    Vec16<T> d[NW * NW];
    uint8_t am[NW * NW][VEC];
    int code[NW * NW];
#pragma unroll
    for (int a = 0; a < NW; ++a)
#pragma unroll
      for (int b = 0; b < NW; ++b) {
        const int oy = oyh - (NW - 1 - a), ox = oxh - (NW - 1 - b);
        const int r = iy + g.ph - oy * g.sh, q = ix + g.pw - ox * g.sw;
        const bool ok = oy >= 0 && oy < g.OH && ox >= 0 && ox < g.OW && r < g.kh && q < g.kw;
        code[a * NW + b] = ok ? r * g.kw + q : 254;                     // 254: no stored code (< 250) matches
        const int cy = min(max(oy, 0), g.OH - 1), cx = min(max(ox, 0), g.OW - 1);
        const size_t o = (((size_t)n * g.OH + cy) * g.OW + cx) * g.C + c * VEC;
        d[a * NW + b].load(dy + o);
        load_codes(o, am[a * NW + b]);
      }
#pragma unroll
    for (int t = 0; t < NW * NW; ++t) add(d[t], am[t], code[t]);
  } else {
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
        load_codes(o, am);
        add(d, am, code);
      }
  }
}

template <typename T, int NW>
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
    pooled_grad<T, VEC, NW>(idx, dy, g, (uint32_t)n, iy, ix, c, acc);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.set(j, acc[j]);
    o.store(dx + i * VEC);
  }
}

// ------------------------------------------------------------------------------------------------
// Stem fusion: y = maxpool(relu(scale * x + shift)) where x is the stem convolution's output.  The
// normalised full-resolution tensor (the largest activation of a ResNet: N x 112 x 112 x 64) is never
// written: forward reads x once and writes the pooled tensor + arg-max codes; backward gathers the pooled
// gradient through the codes on the fly, both for the BatchNorm reduction and for dx = A*g + B*x + K.
// Every thread keeps ONE channel chunk (the launch makes gridDim.x * 256 a multiple of C / VEC).
struct FusedGeom {
  PoolGeom g;
  FastDiv div_ohow, div_ow, div_ihiw, div_iw;
};

template <int VEC> __device__ __forceinline__ void load_vec(const float* __restrict__ p, int c0, float (&o)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; j += 4) {
    const float4 v = *reinterpret_cast<const float4*>(p + c0 + j);
    o[j] = v.x; o[j + 1] = v.y; o[j + 2] = v.z; o[j + 3] = v.w;
  }
}

template <typename T, int KH, int KW>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                  uint8_t* __restrict__ idx,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, FusedGeom fg,
                                                                  long total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const PoolGeom& g = fg.g;
  const int cc = g.C / VEC;
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cc);
  float sc[VEC], sh[VEC];
  load_vec<VEC>(scale, c * VEC, sc);
  load_vec<VEC>(shift, c * VEC, sh);
  for (; i < total; i += stride) {
    const uint32_t pix = (uint32_t)(i / cc);
    const uint32_t n = fdiv(pix, fg.div_ohow);
    const uint32_t rem = pix - n * fg.div_ohow.d;
    const int oy = (int)fdiv(rem, fg.div_ow);
    const int ox = (int)(rem - (uint32_t)oy * fg.div_ow.d);
    float m[VEC];
    int am[VEC];
    // values rounded exactly like the stand-alone BN pass, so fused and unfused runs pick the same arg-max
    window_max<T, VEC, KH, KW, true>(x, g, n, oy, ox, c, sc, sh, m, am);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.set(j, m[j]);
    o.store(y + i * VEC);
    uint8_t code[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) code[j] = (uint8_t)am[j];
    if (VEC == 8) *reinterpret_cast<uint2*>(idx + i * VEC) = *reinterpret_cast<const uint2*>(code);
    else *reinterpret_cast<uint32_t*>(idx + i * VEC) = *reinterpret_cast<const uint32_t*>(code);
  }
}

// MODE 0: partial[block][2][C] = (sum g_m, sum g_m * xhat) over this block's pixels (g_m: pooled gradient, ReLU
// masked).  MODE 1: dx = A*g_m + B*x + K with coef = [A | B | K] (bn_bwd_fold_finalize's work array).
template <typename T, int MODE, int NW>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_kernel(const uint8_t* __restrict__ idx,
                                                                  const T* __restrict__ dyp, const T* __restrict__ x,
                                                                  const float* __restrict__ bn /* mean,invstd,scale,shift */,
                                                                  const float* __restrict__ coef, T* __restrict__ dx,
                                                                  float* __restrict__ partial, FusedGeom fg, long total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  __shared__ float red[MODE == 0 ? 2 : 1][MODE == 0 ? 256 * VEC : 1];
  const PoolGeom& g = fg.g;
  const int C = g.C, cc = C / VEC;
  const long stride = (long)gridDim.x * blockDim.x;
  const int tid = threadIdx.x;
  long i = (long)blockIdx.x * blockDim.x + tid;
  const int c = (int)(i % cc);           // fixed for the thread: stride % cc == 0
  float mu[VEC], is[VEC], sc[VEC], sh[VEC], ca[VEC], cb[VEC], ck[VEC], s1[VEC], s2[VEC];
  load_vec<VEC>(bn + 2 * C, c * VEC, sc);
  load_vec<VEC>(bn + 3 * C, c * VEC, sh);
  if (MODE == 0) {
    load_vec<VEC>(bn, c * VEC, mu);
    load_vec<VEC>(bn + C, c * VEC, is);
#pragma unroll
    for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.f;
  } else {
    load_vec<VEC>(coef, c * VEC, ca);
    load_vec<VEC>(coef + C, c * VEC, cb);
    load_vec<VEC>(coef + 2 * C, c * VEC, ck);
  }
  for (; i < total; i += stride) {
    const uint32_t pix = (uint32_t)(i / cc);
    const uint32_t n = fdiv(pix, fg.div_ihiw);
    const uint32_t rem = pix - n * fg.div_ihiw.d;
    const int iy = (int)fdiv(rem, fg.div_iw);
    const int ix = (int)(rem - (uint32_t)iy * fg.div_iw.d);
    Vec16<T> vx;
    vx.load(x + i * VEC);
    float gr[VEC];
    pooled_grad<T, VEC, NW>(idx, dyp, g, n, iy, ix, c, gr);
    Vec16<T> vo;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float xv = vx.get(j);
      float d = gr[j];
      if (!(to_float(from_float<T>(sc[j] * xv + sh[j])) > 0.f)) d = 0.f;
      else d = to_float(from_float<T>(d));      // the unfused path stores this gradient in T before BN reads it
      if (MODE == 0) {
        s1[j] += d;
        s2[j] += d * ((xv - mu[j]) * is[j]);
      } else {
        vo.set(j, fmaf(ca[j], d, fmaf(cb[j], xv, ck[j])));      // explicit: every instantiation contracts alike
      }
    }
    if (MODE == 1) vo.store(dx + i * VEC);
  }
  if (MODE == 0) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      red[0][tid * VEC + j] = s1[j];
      red[1][tid * VEC + j] = s2[j];
    }
    __syncthreads();
    const int rp = 256 / cc;                 // threads per chunk column inside the block (tid % cc == column)
    for (int t = tid; t < C; t += 256) {
      const int cchunk = t / VEC, j = t - cchunk * VEC;
      // the block's first thread sits at chunk column (blockIdx.x * 256) % cc == 0
      float a1 = 0.f, a2 = 0.f;
      for (int q = 0; q < rp; ++q) {
        a1 += red[0][(q * cc + cchunk) * VEC + j];
        a2 += red[1][(q * cc + cchunk) * VEC + j];
      }
      partial[((size_t)blockIdx.x * 2 + 0) * C + t] = a1;
      partial[((size_t)blockIdx.x * 2 + 1) * C + t] = a2;
    }
  }
}

// The ResNet stem case (3x3 windows, stride 2, padding 1, even input extents): a lane owns a 2 x 2 block of input pixels.
// Rows 2i / 2i + 1 lie in windows {i} / {i, i + 1} (columns alike), so the block meets exactly the four windows
// (i..i+1, j..j+1): 4 gradient chunks + 4 code words are loaded for 4 input pixels, where the per-pixel kernel above
// asked for 4 of each PER pixel (36 load instructions per block instead of 12; 0.97 + 1.01 ms per step at batch 1024).
// Same values, same order of additions (ascending window) as pooled_grad.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_b2_kernel(const uint8_t* __restrict__ idx,
                                                                     const T* __restrict__ dyp, const T* __restrict__ x,
                                                                     const float* __restrict__ bn,
                                                                     const float* __restrict__ coef, T* __restrict__ dx,
                                                                     float* __restrict__ partial, FusedGeom fg,
                                                                     FastDiv div_bhbw, FastDiv div_bw, long total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  __shared__ float red[MODE == 0 ? 2 : 1][MODE == 0 ? 256 * VEC : 1];
  const PoolGeom& g = fg.g;
  const int C = g.C, cc = C / VEC;
  const long stride = (long)gridDim.x * blockDim.x;
  const int tid = threadIdx.x;
  long i = (long)blockIdx.x * blockDim.x + tid;
  const int c = (int)(i % cc);           // fixed for the thread: stride % cc == 0
  float mu[VEC], is[VEC], sc[VEC], sh[VEC], ca[VEC], cb[VEC], ck[VEC], s1[VEC], s2[VEC];
  load_vec<VEC>(bn + 2 * C, c * VEC, sc);
  load_vec<VEC>(bn + 3 * C, c * VEC, sh);
  if (MODE == 0) {
    load_vec<VEC>(bn, c * VEC, mu);
    load_vec<VEC>(bn + C, c * VEC, is);
#pragma unroll
    for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.f;
  } else {
    load_vec<VEC>(coef, c * VEC, ca);
    load_vec<VEC>(coef + C, c * VEC, cb);
    load_vec<VEC>(coef + 2 * C, c * VEC, ck);
  }
  const int BW = g.IW / 2;
  for (; i < total; i += stride) {
    const uint32_t blk = (uint32_t)(i / cc);
    const uint32_t n = fdiv(blk, div_bhbw);
    const uint32_t rem = blk - n * div_bhbw.d;
    const int bi = (int)fdiv(rem, div_bw);
    const int bj = (int)(rem - (uint32_t)bi * BW);
    // the four windows (bi + a, bj + b); the lower / right ones may not exist at the image edge
    Vec16<T> d[2][2];
    uint8_t am[2][2][VEC];
    bool okw[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int oy = bi + a, ox = bj + b;
        okw[a][b] = oy < g.OH && ox < g.OW;
        const size_t o = (((size_t)n * g.OH + min(oy, g.OH - 1)) * g.OW + min(ox, g.OW - 1)) * C + c * VEC;
        d[a][b].load(dyp + o);
        if (VEC == 8) *reinterpret_cast<uint2*>(am[a][b]) = *reinterpret_cast<const uint2*>(idx + o);
        else *reinterpret_cast<uint32_t*>(am[a][b]) = *reinterpret_cast<const uint32_t*>(idx + o);
      }
    Vec16<T> vx[2][2];
    const size_t xo = (((size_t)n * g.IH + 2 * bi) * g.IW + 2 * bj) * C + c * VEC;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) vx[a][b].load(x + xo + ((size_t)a * g.IW + b) * C);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        // pixel (2bi + a, 2bj + b): window (bi + wa, bj + wb) holds it at code (a + 1 - 2wa) * 3 + (b + 1 - 2wb),
        // wa in {0} for a = 0, {0, 1} for a = 1 (columns alike); ascending (window row, window column)
        Vec16<T> vo;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float gsum = 0.f;
#pragma unroll
          for (int wa = 0; wa <= a; ++wa)
#pragma unroll
            for (int wb = 0; wb <= b; ++wb) {
              const int code = (a + 1 - 2 * wa) * 3 + (b + 1 - 2 * wb);
              if (okw[wa][wb] && am[wa][wb][j] == code) gsum += d[wa][wb].get(j);
            }
          const float xv = vx[a][b].get(j);
          float dd = gsum;
          if (!(to_float(from_float<T>(sc[j] * xv + sh[j])) > 0.f)) dd = 0.f;
          else dd = to_float(from_float<T>(dd));      // the unfused path stores this gradient in T before BN reads it
          if (MODE == 0) {
            s1[j] += dd;
            s2[j] += dd * ((xv - mu[j]) * is[j]);
          } else {
            vo.set(j, fmaf(ca[j], dd, fmaf(cb[j], xv, ck[j])));
          }
        }
        if (MODE == 1) vo.store(dx + xo + ((size_t)a * g.IW + b) * C);
      }
  }
  if (MODE == 0) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      red[0][tid * VEC + j] = s1[j];
      red[1][tid * VEC + j] = s2[j];
    }
    __syncthreads();
    const int rp = 256 / cc;
    for (int t = tid; t < C; t += 256) {
      const int cchunk = t / VEC, j = t - cchunk * VEC;
      float a1 = 0.f, a2 = 0.f;
      for (int q = 0; q < rp; ++q) {
        a1 += red[0][(q * cc + cchunk) * VEC + j];
        a2 += red[1][(q * cc + cchunk) * VEC + j];
      }
      partial[((size_t)blockIdx.x * 2 + 0) * C + t] = a1;
      partial[((size_t)blockIdx.x * 2 + 1) * C + t] = a2;
    }
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

// Fixed-window average pooling with zero padding counted in the divisor (torch's count_include_pad=True; Inception's
// 3x3 stride-1 pad-1 branch pool).  Windows overlap, so backward gathers: input pixel (iy, ix) sums dy over the output
// positions whose window covers it -- no atomics, fixed order.
template <typename T>
__global__ __launch_bounds__(256) void avgpool_win_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, PoolGeom g,
                                                              long total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int cc = g.C / VEC;
  const float inv = 1.f / (float)(g.kh * g.kw);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cc);
    long pix = i / cc;
    const int ox = (int)(pix % g.OW); pix /= g.OW;
    const int oy = (int)(pix % g.OH);
    const int n = (int)(pix / g.OH);
    const int y0 = max(oy * g.sh - g.ph, 0), y1 = min(oy * g.sh - g.ph + g.kh, g.IH);
    const int x0 = max(ox * g.sw - g.pw, 0), x1 = min(ox * g.sw - g.pw + g.kw, g.IW);
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
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.set(j, acc[j] * inv);
    o.store(y + i * VEC);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_win_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, PoolGeom g,
                                                              long total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int cc = g.C / VEC;
  const float inv = 1.f / (float)(g.kh * g.kw);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cc);
    long pix = i / cc;
    const int ix = (int)(pix % g.IW); pix /= g.IW;
    const int iy = (int)(pix % g.IH);
    const int n = (int)(pix / g.IH);
    // outputs o with o*s - p <= i < o*s - p + k
    const int ty = iy + g.ph, tx = ix + g.pw;
    const int oy0 = max((ty - g.kh + g.sh) / g.sh, 0), oy1 = min(ty / g.sh, g.OH - 1);
    const int ox0 = max((tx - g.kw + g.sw) / g.sw, 0), ox1 = min(tx / g.sw, g.OW - 1);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy)
      for (int ox = ox0; ox <= ox1; ++ox) {
        Vec16<T> d;
        d.load(dy + (((size_t)n * g.OH + oy) * g.OW + ox) * g.C + c * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += d.get(j);
      }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.set(j, acc[j] * inv);
    o.store(dx + i * VEC);
  }
}

// Global average pool (output 1x1: classifier heads, every squeeze-excitation block).  The adaptive kernel above gives
// one lane a whole H*W column -- 12,544 serial loads at 112x112.  Here a workgroup owns sample n and CL channel chunks;
// its 256 / CL pixel rows stride over H*W and meet in LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void global_avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, int HW, int chunks,
                                                             int CL) {
  constexpr int VEC = 16 / (int)sizeof(T);
  __shared__ float red[256][VEC + 1];
  const int cl = threadIdx.x % CL, pr = threadIdx.x / CL, PR = 256 / CL;
  const int ch = blockIdx.x * CL + cl;
  const long n = blockIdx.y;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (ch < chunks) {
    const T* base = x + (n * HW * (long)chunks + ch) * VEC;
    int p = pr;
    for (; p + 3 * PR < HW; p += 4 * PR) {           // four loads in flight per lane
      Vec16<T> v0, v1, v2, v3;
      v0.load(base + (long)p * chunks * VEC);
      v1.load(base + (long)(p + PR) * chunks * VEC);
      v2.load(base + (long)(p + 2 * PR) * chunks * VEC);
      v3.load(base + (long)(p + 3 * PR) * chunks * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += (v0.get(j) + v1.get(j)) + (v2.get(j) + v3.get(j));
    }
    for (; p < HW; p += PR) {
      Vec16<T> v;
      v.load(base + (long)p * chunks * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += v.get(j);
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[threadIdx.x][j] = acc[j];
  __syncthreads();
  if (pr == 0 && ch < chunks) {
    const float inv = 1.f / (float)HW;
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float t = 0.f;
      for (int r = 0; r < PR; ++r) t += red[r * CL + cl][j];
      o.set(j, t * inv);
    }
    o.store(y + (n * (long)chunks + ch) * VEC);
  }
}

inline int grid_for(long n) {
  long b = (n + 255) / 256;
  if (b > g_stream_grid) b = g_stream_grid;
  return (int)(b < 1 ? 1 : b);
}
// a pixel lies in at most 2 windows per axis (3x3 s2, 2x2 s2, 3x3 s2 ceil ...): the gather's fast path
inline bool two_windows(int kh, int kw, int sh, int sw) { return kh <= 2 * sh && kw <= 2 * sw; }
int g_pool_b2 = 1;       // 0: never the 2x2-block stem kernel (A/B hook)
// the 2x2-block backward kernel: 3x3 windows, stride 2, padding 1, even extents (the ResNet / DenseNet / GoogLeNet stems)
inline bool stem_blocks(int kh, int kw, int sh, int sw, int ph, int pw, int IH, int IW, int OH, int OW) {
  return g_pool_b2 && kh == 3 && kw == 3 && sh == 2 && sw == 2 && ph == 1 && pw == 1 && IH % 2 == 0 && IW % 2 == 0 &&
         OH == IH / 2 && OW == IW / 2;
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
  if (kh == 3 && kw == 3) {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((maxpool_fwd_kernel<T, 3, 3>), dim3(grid_for(total)), dim3(256), 0, s,
                                                 (const T*)x, (T*)y, idx, g, total));
  } else if (kh == 2 && kw == 2) {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((maxpool_fwd_kernel<T, 2, 2>), dim3(grid_for(total)), dim3(256), 0, s,
                                                 (const T*)x, (T*)y, idx, g, total));
  } else {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((maxpool_fwd_kernel<T, 0, 0>), dim3(grid_for(total)), dim3(256), 0, s,
                                                 (const T*)x, (T*)y, idx, g, total));
  }
  DPA_CHECK_HIP(hipGetLastError());
}
void maxpool_bwd_launch(const uint8_t* idx, const void* dy, void* dx, int NB, int IH, int IW, int C, int OH, int OW,
                        int kh, int kw, int sh, int sw, int ph, int pw, int dtype, hipStream_t s) {
  check_c(C, dtype);
  PoolGeom g{NB, IH, IW, C, OH, OW, kh, kw, sh, sw, ph, pw};
  const long total = (long)NB * IH * IW * (C / (dtype == kF32 ? 4 : 8));
  if (total <= 0) return;
  if (kh <= sh && kw <= sw) {
    // non-overlapping windows (VGG's 2x2 stride 2): one candidate window per pixel instead of four
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((maxpool_bwd_kernel<T, 1>), dim3(grid_for(total)), dim3(256), 0, s,
                                                 idx, (const T*)dy, (T*)dx, g, total));
  } else if (two_windows(kh, kw, sh, sw)) {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((maxpool_bwd_kernel<T, 2>), dim3(grid_for(total)), dim3(256), 0, s,
                                                 idx, (const T*)dy, (T*)dx, g, total));
  } else if (kh <= 3 * sh && kw <= 3 * sw) {
    // 3x3 stride 1 (the pooling branch of every Inception block): nine candidate windows, all loaded up front
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((maxpool_bwd_kernel<T, 3>), dim3(grid_for(total)), dim3(256), 0, s,
                                                 idx, (const T*)dy, (T*)dx, g, total));
  } else {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((maxpool_bwd_kernel<T, 0>), dim3(grid_for(total)), dim3(256), 0, s,
                                                 idx, (const T*)dy, (T*)dx, g, total));
  }
  DPA_CHECK_HIP(hipGetLastError());
}
namespace {
FusedGeom fused_geom(int NB, int IH, int IW, int C, int OH, int OW, int kh, int kw, int sh, int sw, int ph, int pw,
                     int dtype) {
  check_c(C, dtype);
  const int cc = C / (dtype == kF32 ? 4 : 8);
  if (256 % cc != 0) throw std::invalid_argument("fused BN+ReLU+maxpool: C / vector width must divide 256");
  if ((long)NB * IH * IW >= (1l << 31) || kh * kw > 250)
    throw std::invalid_argument("fused BN+ReLU+maxpool: problem too large");
  FusedGeom fg;
  fg.g = PoolGeom{NB, IH, IW, C, OH, OW, kh, kw, sh, sw, ph, pw};
  fg.div_ohow = make_fastdiv((uint32_t)(OH * OW));
  fg.div_ow = make_fastdiv((uint32_t)OW);
  fg.div_ihiw = make_fastdiv((uint32_t)(IH * IW));
  fg.div_iw = make_fastdiv((uint32_t)IW);
  return fg;
}
}  // namespace

int bn_relu_maxpool_partial_rows(long in_chunks) {
  long b = (in_chunks + 256 * 8 - 1) / (256 * 8);     // >= 8 chunks per thread
  if (b > g_stream_grid / 2) b = g_stream_grid / 2;
  return (int)(b < 1 ? 1 : b);
}

void bn_relu_maxpool_fwd_launch(const void* x, void* y, uint8_t* idx, const float* scale, const float* shift, int NB,
                                int IH, int IW, int C, int OH, int OW, int kh, int kw, int sh, int sw, int ph, int pw,
                                int dtype, hipStream_t s) {
  const FusedGeom fg = fused_geom(NB, IH, IW, C, OH, OW, kh, kw, sh, sw, ph, pw, dtype);
  const long total = (long)NB * OH * OW * (C / (dtype == kF32 ? 4 : 8));
  if (total <= 0) return;
  if (kh == 3 && kw == 3) {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_relu_maxpool_fwd_kernel<T, 3, 3>), dim3(grid_for(total)), dim3(256),
                                                 0, s, (const T*)x, (T*)y, idx, scale, shift, fg, total));
  } else if (kh == 2 && kw == 2) {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_relu_maxpool_fwd_kernel<T, 2, 2>), dim3(grid_for(total)), dim3(256),
                                                 0, s, (const T*)x, (T*)y, idx, scale, shift, fg, total));
  } else {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_relu_maxpool_fwd_kernel<T, 0, 0>), dim3(grid_for(total)), dim3(256),
                                                 0, s, (const T*)x, (T*)y, idx, scale, shift, fg, total));
  }
  DPA_CHECK_HIP(hipGetLastError());
}

void bn_relu_maxpool_bwd_reduce_launch(const uint8_t* idx, const void* dyp, const void* x, const float* bn,
                                       float* partial, int nblk, int NB, int IH, int IW, int C, int OH, int OW, int kh,
                                       int kw, int sh, int sw, int ph, int pw, int dtype, hipStream_t s) {
  const FusedGeom fg = fused_geom(NB, IH, IW, C, OH, OW, kh, kw, sh, sw, ph, pw, dtype);
  const long total = (long)NB * IH * IW * (C / (dtype == kF32 ? 4 : 8));
  if (total <= 0 || nblk <= 0) return;
  if (stem_blocks(kh, kw, sh, sw, ph, pw, IH, IW, OH, OW)) {
    const FastDiv dbb = make_fastdiv((uint32_t)((IH / 2) * (IW / 2))), dbw = make_fastdiv((uint32_t)(IW / 2));
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_relu_maxpool_bwd_b2_kernel<T, 0>), dim3(nblk), dim3(256), 0, s, idx,
                                                 (const T*)dyp, (const T*)x, bn, nullptr, (T*)nullptr, partial, fg, dbb,
                                                 dbw, total / 4));
    DPA_CHECK_HIP(hipGetLastError());
    return;
  }
  if (two_windows(kh, kw, sh, sw)) {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_relu_maxpool_bwd_kernel<T, 0, 2>), dim3(nblk), dim3(256), 0, s, idx,
                                                 (const T*)dyp, (const T*)x, bn, nullptr, (T*)nullptr, partial, fg,
                                                 total));
  } else {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_relu_maxpool_bwd_kernel<T, 0, 0>), dim3(nblk), dim3(256), 0, s, idx,
                                                 (const T*)dyp, (const T*)x, bn, nullptr, (T*)nullptr, partial, fg,
                                                 total));
  }
  DPA_CHECK_HIP(hipGetLastError());
}

void bn_relu_maxpool_bwd_apply_launch(const uint8_t* idx, const void* dyp, const void* x, const float* bn,
                                      const float* coef, void* dx, int NB, int IH, int IW, int C, int OH, int OW,
                                      int kh, int kw, int sh, int sw, int ph, int pw, int dtype, hipStream_t s) {
  const FusedGeom fg = fused_geom(NB, IH, IW, C, OH, OW, kh, kw, sh, sw, ph, pw, dtype);
  const long total = (long)NB * IH * IW * (C / (dtype == kF32 ? 4 : 8));
  if (total <= 0) return;
  if (stem_blocks(kh, kw, sh, sw, ph, pw, IH, IW, OH, OW)) {
    const FastDiv dbb = make_fastdiv((uint32_t)((IH / 2) * (IW / 2))), dbw = make_fastdiv((uint32_t)(IW / 2));
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_relu_maxpool_bwd_b2_kernel<T, 1>), dim3(grid_for(total / 4)),
                                                 dim3(256), 0, s, idx, (const T*)dyp, (const T*)x, bn, coef, (T*)dx,
                                                 nullptr, fg, dbb, dbw, total / 4));
    DPA_CHECK_HIP(hipGetLastError());
    return;
  }
  if (two_windows(kh, kw, sh, sw)) {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_relu_maxpool_bwd_kernel<T, 1, 2>), dim3(grid_for(total)), dim3(256),
                                                 0, s, idx, (const T*)dyp, (const T*)x, bn, coef, (T*)dx, nullptr, fg,
                                                 total));
  } else {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_relu_maxpool_bwd_kernel<T, 1, 0>), dim3(grid_for(total)), dim3(256),
                                                 0, s, idx, (const T*)dyp, (const T*)x, bn, coef, (T*)dx, nullptr, fg,
                                                 total));
  }
  DPA_CHECK_HIP(hipGetLastError());
}

void avgpool_fwd_launch(const void* x, void* y, int NB, int IH, int IW, int C, int OH, int OW, int dtype,
                        hipStream_t s) {
  check_c(C, dtype);
  PoolGeom g{NB, IH, IW, C, OH, OW, 0, 0, 0, 0, 0, 0};
  const long total = (long)NB * OH * OW * (C / (dtype == kF32 ? 4 : 8));
  if (total <= 0) return;
  if (OH == 1 && OW == 1 && IH * IW >= 16 && NB <= 65535) {
    const int chunks = C / (dtype == kF32 ? 4 : 8);
    int CL = 1;
    while (CL < chunks && CL < 16) CL <<= 1;
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(global_avgpool_kernel<T>, dim3(ceil_div(chunks, CL), NB), dim3(256), 0, s,
                                                 (const T*)x, (T*)y, IH * IW, chunks, CL));
    DPA_CHECK_HIP(hipGetLastError());
    return;
  }
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

void avgpool_win_launch(const void* src, void* dst, int backward, int NB, int IH, int IW, int C, int OH, int OW, int kh,
                        int kw, int sh, int sw, int ph, int pw, int dtype, hipStream_t s) {
  check_c(C, dtype);
  if (kh < 1 || kw < 1 || sh < 1 || sw < 1 || ph < 0 || pw < 0 || 2 * ph > kh || 2 * pw > kw)
    throw std::invalid_argument("avgpool_win: bad window");
  if (OH != (IH + 2 * ph - kh) / sh + 1 || OW != (IW + 2 * pw - kw) / sw + 1)
    throw std::invalid_argument("avgpool_win: output extent does not match the window");
  PoolGeom g{NB, IH, IW, C, OH, OW, kh, kw, sh, sw, ph, pw};
  const long total = (long)NB * (backward ? IH * IW : OH * OW) * (C / (dtype == kF32 ? 4 : 8));
  if (total <= 0) return;
  if (backward) {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(avgpool_win_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s,
                                                 (const T*)src, (T*)dst, g, total));
  } else {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(avgpool_win_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s,
                                                 (const T*)src, (T*)dst, g, total));
  }
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
