// LayerNorm over the channel (last) axis of NHWC activations (gfx950) -- ConvNeXt's LayerNorm / LayerNorm2d; in NHWC both
// are the same row-wise op, so torchvision's NCHW<->NHWC permutes around them disappear.
//
// A row (C channels, 16-byte chunks) is owned by G lanes of one wave, G = the power of two >= chunks (4..64): C = 96 bf16
// packs four rows into a wave instead of idling 52 lanes.  Rows wider than 64 chunks give every lane up to MAXCH chunks.
// The row stays in registers between the statistics and the normalise pass (one read, one write), statistics are a
// true two-pass mean / variance in fp32.  Backward keeps the per-channel dgamma / dbeta sums lane-local across all rows
// a lane visits (its chunk positions never change), meets across row groups with shuffles, across waves in LDS, and
// writes ONE partial row per workgroup; a small second kernel folds the partial rows in a fixed order (no atomics).
#include "../common.h"
#include "kernels.h"

namespace dpa {
namespace {

constexpr int kLnMaxBlocks = 512;

__device__ __forceinline__ float group_sum(float v, int G) {
  for (int off = G >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <typename T, int MAXCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     long rows, int C, int G, float eps) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int chunks = C / VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane & (G - 1);                 // lane inside its row group
  const int rpw = 64 / G;                        // rows per wave
  const long rstride = (long)gridDim.x * 4 * rpw;
  const float invC = 1.f / (float)C;
  // uniform trip count per wave (the shuffles need all 64 lanes): iterate on the wave's first row and mask the
  // lanes whose row is past the end
  for (long base = ((long)blockIdx.x * 4 + wave) * rpw; base < rows; base += rstride) {
    const long row = base + lane / G;
    const bool live = row < rows;
    Vec16<T> v[MAXCH];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
      const int ch = gl + k * G;
      if (live && ch < chunks) {
        v[k].load(x + row * C + ch * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) s += v[k].get(j);
      }
    }
    const float mean = group_sum(s, G) * invC;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
      const int ch = gl + k * G;
      if (live && ch < chunks) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float d = v[k].get(j) - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(group_sum(q, G) * invC + eps);
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
      const int ch = gl + k * G;
      if (live && ch < chunks) {
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          o.set(j, (v[k].get(j) - mean) * rstd * w[ch * VEC + j] + b[ch * VEC + j]);
        o.store(y + row * C + ch * VEC);
      }
    }
    if (live && gl == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
  }
}

template <typename T, int MAXCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ w, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, T* __restrict__ dx,
                                                     float* __restrict__ partial, long rows, int C, int G) {
  constexpr int VEC = 16 / (int)sizeof(T);
  extern __shared__ float red[];                 // [2][C]
  const int chunks = C / VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane & (G - 1);
  const int rpw = 64 / G;
  const long rstride = (long)gridDim.x * 4 * rpw;
  const float invC = 1.f / (float)C;
  float dw[MAXCH][VEC], db[MAXCH][VEC];
#pragma unroll
  for (int k = 0; k < MAXCH; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) dw[k][j] = db[k][j] = 0.f;

  const long first = ((long)blockIdx.x * 4 + wave) * rpw;
  // uniform trip count per wave: iterate on the wave's first row, mask the lanes whose row is past the end
  for (long base = first; base < rows; base += rstride) {
    const long row = base + lane / G;
    const bool live = row < rows;
    const float mean = live ? mean_in[row] : 0.f, rstd = live ? rstd_in[row] : 0.f;
    Vec16<T> g[MAXCH], v[MAXCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
      const int ch = gl + k * G;
      if (live && ch < chunks) {
        g[k].load(dy + row * C + ch * VEC);
        v[k].load(x + row * C + ch * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float gj = g[k].get(j), xh = (v[k].get(j) - mean) * rstd;
          const float gw = gj * w[ch * VEC + j];
          s1 += gw;
          s2 += gw * xh;
          dw[k][j] += gj * xh;
          db[k][j] += gj;
        }
      }
    }
    s1 = group_sum(s1, G) * invC;
    s2 = group_sum(s2, G) * invC;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
      const int ch = gl + k * G;
      if (live && ch < chunks) {
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float xh = (v[k].get(j) - mean) * rstd;
          o.set(j, rstd * (g[k].get(j) * w[ch * VEC + j] - s1 - xh * s2));
        }
        o.store(dx + row * C + ch * VEC);
      }
    }
  }
  // row groups of one wave hold sums for the same channels: meet with shuffles (lane bits >= log2 G)
#pragma unroll
  for (int k = 0; k < MAXCH; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      for (int off = 32; off >= G; off >>= 1) {
        dw[k][j] += __shfl_xor(dw[k][j], off, 64);
        db[k][j] += __shfl_xor(db[k][j], off, 64);
      }
    }
  // waves meet in LDS, one after the other (fixed order)
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv && lane < G) {
#pragma unroll
      for (int k = 0; k < MAXCH; ++k) {
        const int ch = gl + k * G;
        if (ch < chunks) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const int c = ch * VEC + j;
            red[c] = (wv == 0 ? 0.f : red[c]) + dw[k][j];
            red[C + c] = (wv == 0 ? 0.f : red[C + c]) + db[k][j];
          }
        }
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < 2 * C; c += 256) partial[(size_t)blockIdx.x * 2 * C + c] = red[c];
}

// out[which][c] = sum over partial rows (which 0 -> dgamma, 1 -> dbeta).  16 column lanes x 16 part lanes per workgroup,
// four rows in flight per lane, fixed order.
__global__ __launch_bounds__(256) void ln_fold_kernel(const float* __restrict__ partial, int nparts, int C,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, float beta_w,
                                                      float beta_b) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;          // column of the [2*C] row
  float a = 0.f;
  if (c < 2 * C) {
    int p = pl;
    for (; p + 48 < nparts; p += 64) {
      const float v0 = partial[(size_t)p * 2 * C + c], v1 = partial[(size_t)(p + 16) * 2 * C + c];
      const float v2 = partial[(size_t)(p + 32) * 2 * C + c], v3 = partial[(size_t)(p + 48) * 2 * C + c];
      a += (v0 + v1) + (v2 + v3);
    }
    for (; p < nparts; p += 16) a += partial[(size_t)p * 2 * C + c];
  }
  red[pl][cl] = a;
  __syncthreads();
  if (pl != 0 || c >= 2 * C) return;
  float t = 0.f;
  for (int q = 0; q < 16; ++q) t += red[q][cl];
  if (c < C) dgamma[c] = beta_w == 0.f ? t : t + beta_w * dgamma[c];
  else dbeta[c - C] = beta_b == 0.f ? t : t + beta_b * dbeta[c - C];
}

struct LnPlan {
  int G, maxch, blocks;
};
LnPlan ln_plan(long rows, int C, int dtype) {
  const int vec = dtype == kF32 ? 4 : 8;
  if (C <= 0 || C % vec != 0) throw std::invalid_argument("layer_norm: C must fill 16-byte vectors");
  const int chunks = C / vec;
  if (chunks > 512) throw std::invalid_argument("layer_norm: C too large (max 512 16-byte chunks per row)");
  LnPlan p;
  if (chunks <= 64) {
    p.G = 4;
    while (p.G < chunks) p.G <<= 1;
    p.maxch = 1;
  } else {
    p.G = 64;
    p.maxch = chunks <= 128 ? 2 : (chunks <= 256 ? 4 : 8);
  }
  const long rows_per_block = 4 * (64 / p.G);
  long b = (rows + rows_per_block - 1) / rows_per_block;
  if (b > kLnMaxBlocks) b = kLnMaxBlocks;
  p.blocks = (int)(b < 1 ? 1 : b);
  return p;
}

}  // namespace

int ln_partial_rows(long rows, int C, int dtype) { return ln_plan(rows, C, dtype).blocks; }

void ln_fwd_launch(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, long rows, int C,
                   float eps, int dtype, hipStream_t s) {
  if (rows <= 0) return;
  const LnPlan p = ln_plan(rows, C, dtype);
#define DPA_LN_FWD(MC)                                                                                              \
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_fwd_kernel<T, MC>), dim3(p.blocks), dim3(256), 0, s, (const T*)x, w, \
                                               b, (T*)y, mean, rstd, rows, C, p.G, eps))
  if (p.maxch == 1) { DPA_LN_FWD(1); }
  else if (p.maxch == 2) { DPA_LN_FWD(2); }
  else if (p.maxch == 4) { DPA_LN_FWD(4); }
  else { DPA_LN_FWD(8); }
#undef DPA_LN_FWD
  DPA_CHECK_HIP(hipGetLastError());
}

void ln_bwd_launch(const void* dy, const void* x, const float* w, const float* mean, const float* rstd, void* dx,
                   float* partial, float* dgamma, float* dbeta, float beta_w, float beta_b, long rows, int C, int dtype,
                   hipStream_t s) {
  if (rows <= 0) return;
  const LnPlan p = ln_plan(rows, C, dtype);
  const size_t lds = (size_t)2 * C * sizeof(float);
#define DPA_LN_BWD(MC)                                                                                                \
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, MC>), dim3(p.blocks), dim3(256), lds, s, (const T*)dy, \
                                               (const T*)x, w, mean, rstd, (T*)dx, partial, rows, C, p.G))
  if (p.maxch == 1) { DPA_LN_BWD(1); }
  else if (p.maxch == 2) { DPA_LN_BWD(2); }
  else if (p.maxch == 4) { DPA_LN_BWD(4); }
  else { DPA_LN_BWD(8); }
#undef DPA_LN_BWD
  hipLaunchKernelGGL(ln_fold_kernel, dim3(ceil_div(2 * C, 16)), dim3(256), 0, s, partial, p.blocks, C, dgamma, dbeta,
                     beta_w, beta_b);
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
