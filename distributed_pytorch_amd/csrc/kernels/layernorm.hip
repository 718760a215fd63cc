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

#ifndef DPA_LN_BLOCKS
#define DPA_LN_BLOCKS 1280
#endif
constexpr int kLnMaxBlocks = DPA_LN_BLOCKS;     // 5 workgroups per CU
#ifndef DPA_LN_BWD_BLOCKS
#define DPA_LN_BWD_BLOCKS 768
#endif
// backward, rows wider than 64 chunks: every workgroup ends with a serial 4-wave merge and a 2C-float partial row;
// measured on [25216, 768]: 512 workgroups 47.9 us, 768 (one resident round at 3 waves per SIMD) 40.7, 1024 52.8;
// rows over 128 chunks (one row per trip, 1-2 waves per SIMD) keep 512
constexpr int kLnMaxBlocksWide = DPA_LN_BWD_BLOCKS;
#ifndef DPA_LN_BWD_R2
#define DPA_LN_BWD_R2 2
#endif
constexpr int kLnRows = 2;                     // rows per wave and trip for rows of up to 128 chunks

__device__ __forceinline__ float group_sum(float v, int G) {
  for (int off = G >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// R rows per wave and trip, every load of the trip requested before the first use; for rows of up to 128 chunks
// (MAXCH <= 2) the lane's gamma / beta values live in registers for the whole kernel (its chunk positions never change).
// The first version kept one row per wave in flight on 512 workgroups with 16 scalar parameter loads per chunk:
// 2.2-2.4 TB/s on ViT-B's [25216, 768] activations, which fit the Infinity Cache.
template <typename T, int MAXCH>
__device__ __forceinline__ void ln_load_params(const float* __restrict__ p, int gl, int G, int chunks,
                                               float (&r)[MAXCH][16 / sizeof(T)]) {
  constexpr int VEC = 16 / (int)sizeof(T);
#pragma unroll
  for (int k = 0; k < MAXCH; ++k) {
    const int ch = gl + k * G;
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      const float4 a = ch < chunks ? reinterpret_cast<const float4*>(p + ch * VEC)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      r[k][4 * q] = a.x; r[k][4 * q + 1] = a.y; r[k][4 * q + 2] = a.z; r[k][4 * q + 3] = a.w;
    }
  }
}

template <typename T, int MAXCH, int R>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     long rows, int C, int G, float eps) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr bool HOIST = MAXCH <= 2;
  const int chunks = C / VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane & (G - 1);                 // lane inside its row group
  const int rpw = 64 / G;                        // rows per wave and sub-trip
  const long rstride = (long)gridDim.x * 4 * rpw * R;
  const float invC = 1.f / (float)C;
  float wr[HOIST ? MAXCH : 1][VEC], br[HOIST ? MAXCH : 1][VEC];
  if constexpr (HOIST) {
    ln_load_params<T, MAXCH>(w, gl, G, chunks, wr);
    ln_load_params<T, MAXCH>(b, gl, G, chunks, br);
  }
  // uniform trip count per wave (the shuffles need all 64 lanes): iterate on the wave's first row and mask the
  // lanes whose row is past the end
  for (long base = ((long)blockIdx.x * 4 + wave) * rpw * R; base < rows; base += rstride) {
    Vec16<T> v[R][MAXCH];
    long row[R];
    bool live[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = base + r * rpw + lane / G;
      live[r] = row[r] < rows;
#pragma unroll
      for (int k = 0; k < MAXCH; ++k) {
        const int ch = gl + k * G;
        if (live[r] && ch < chunks) v[r][k].load(x + row[r] * C + ch * VEC);
        else v[r][k].raw = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    float s[R], q[R], mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s[r] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXCH; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[r] += v[r][k].get(j);      // absent chunks hold zeros
    }
#pragma unroll
    for (int r = 0; r < R; ++r) mean[r] = group_sum(s[r], G) * invC;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      q[r] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXCH; ++k) {
        const int ch = gl + k * G;
        if (ch < chunks) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float d = v[r][k].get(j) - mean[r];
            q[r] += d * d;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) rstd[r] = rsqrtf(group_sum(q[r], G) * invC + eps);
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int k = 0; k < MAXCH; ++k) {
        const int ch = gl + k * G;
        if (live[r] && ch < chunks) {
          Vec16<T> o;
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float wj = HOIST ? wr[HOIST ? k : 0][j] : w[ch * VEC + j];
            const float bj = HOIST ? br[HOIST ? k : 0][j] : b[ch * VEC + j];
            o.set(j, (v[r][k].get(j) - mean[r]) * rstd[r] * wj + bj);
          }
          o.store(y + row[r] * C + ch * VEC);
        }
      }
      if (live[r] && gl == 0) {
        mean_out[row[r]] = mean[r];
        rstd_out[row[r]] = rstd[r];
      }
    }
  }
}

template <typename T, int MAXCH, int R>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ w, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, T* __restrict__ dx,
                                                     float* __restrict__ partial, long rows, int C, int G) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr bool HOIST = MAXCH <= 2;
  extern __shared__ float red[];                 // [2][C]
  const int chunks = C / VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane & (G - 1);
  const int rpw = 64 / G;
  const long rstride = (long)gridDim.x * 4 * rpw * R;
  const float invC = 1.f / (float)C;
  float dw[MAXCH][VEC], db[MAXCH][VEC], wr[HOIST ? MAXCH : 1][VEC];
#pragma unroll
  for (int k = 0; k < MAXCH; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) dw[k][j] = db[k][j] = 0.f;
  if constexpr (HOIST) ln_load_params<T, MAXCH>(w, gl, G, chunks, wr);

  const long first = ((long)blockIdx.x * 4 + wave) * rpw * R;
  // uniform trip count per wave: iterate on the wave's first row, mask the lanes whose row is past the end
  for (long base = first; base < rows; base += rstride) {
    Vec16<T> g[R][MAXCH], v[R][MAXCH];
    long row[R];
    bool live[R];
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      row[r] = base + r * rpw + lane / G;
      live[r] = row[r] < rows;
      mean[r] = live[r] ? mean_in[row[r]] : 0.f;
      rstd[r] = live[r] ? rstd_in[row[r]] : 0.f;
#pragma unroll
      for (int k = 0; k < MAXCH; ++k) {
        const int ch = gl + k * G;
        if (live[r] && ch < chunks) {
          g[r][k].load(dy + row[r] * C + ch * VEC);
          v[r][k].load(x + row[r] * C + ch * VEC);
        } else {
          g[r][k].raw = make_uint4(0u, 0u, 0u, 0u);
          v[r][k].raw = make_uint4(0u, 0u, 0u, 0u);
        }
      }
    }
    float s1[R], s2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s1[r] = s2[r] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXCH; ++k) {
        const int ch = gl + k * G;
        if (ch < chunks) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float gj = g[r][k].get(j), xh = (v[r][k].get(j) - mean[r]) * rstd[r];   // dead rows: g = 0, rstd = 0
            const float gw = gj * (HOIST ? wr[HOIST ? k : 0][j] : w[ch * VEC + j]);
            s1[r] += gw;
            s2[r] += gw * xh;
            dw[k][j] += gj * xh;
            db[k][j] += gj;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s1[r] = group_sum(s1[r], G) * invC;
      s2[r] = group_sum(s2[r], G) * invC;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int k = 0; k < MAXCH; ++k) {
        const int ch = gl + k * G;
        if (live[r] && ch < chunks) {
          Vec16<T> o;
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float xh = (v[r][k].get(j) - mean[r]) * rstd[r];
            const float wj = HOIST ? wr[HOIST ? k : 0][j] : w[ch * VEC + j];
            o.set(j, rstd[r] * (g[r][k].get(j) * wj - s1[r] - xh * s2[r]));
          }
          o.store(dx + row[r] * C + ch * VEC);
        }
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

// out[which][c] = sum over partial rows (which 0 -> dgamma, 1 -> dbeta).  A workgroup owns 16 columns of the [2C] row:
// 4 column lanes (one float4 each) x 64 part lanes, eight rows in flight per lane (the first version walked 32 rows
// per lane four at a time on 96 workgroups: 10+ us of dependent latency behind every LayerNorm backward), then a
// fixed-order tree over the part lanes in LDS.  2C is a multiple of 8, a float4 never straddles the dgamma / dbeta seam.
__global__ __launch_bounds__(256) void ln_fold_kernel(const float* __restrict__ partial, int nparts, int C,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, float beta_w,
                                                      float beta_b) {
  __shared__ float4 red[64][4];
  const int cl = threadIdx.x & 3, pl = threadIdx.x >> 2;
  const int c = (blockIdx.x * 4 + cl) * 4;     // first of this lane's 4 columns of the [2*C] row
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < 2 * C) {
    int p = pl;
    for (; p + 7 * 64 < nparts; p += 8 * 64) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(partial + (size_t)(p + u * 64) * 2 * C + c);
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        a.x += v[u].x + v[u + 1].x; a.y += v[u].y + v[u + 1].y;
        a.z += v[u].z + v[u + 1].z; a.w += v[u].w + v[u + 1].w;
      }
    }
    for (; p < nparts; p += 64) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)p * 2 * C + c);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[pl][cl] = a;
  __syncthreads();
  for (int h = 32; h > 0; h >>= 1) {
    if (pl < h) {
      const float4 o = red[pl + h][cl];
      float4 m = red[pl][cl];
      m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
      red[pl][cl] = m;
    }
    __syncthreads();
  }
  if (pl != 0 || c >= 2 * C) return;
  const float4 t4 = red[0][cl];
  const float t[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cj = c + j;
    if (cj < C) dgamma[cj] = beta_w == 0.f ? t[j] : t[j] + beta_w * dgamma[cj];
    else dbeta[cj - C] = beta_b == 0.f ? t[j] : t[j] + beta_b * dbeta[cj - C];
  }
}

struct LnPlan {
  int G, maxch, blocks, blocks_bwd;
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
  const long rows_per_block = 4 * (64 / p.G) * (p.maxch <= 2 ? kLnRows : 1);
  long b = (rows + rows_per_block - 1) / rows_per_block;
  if (b > kLnMaxBlocks) b = kLnMaxBlocks;
  p.blocks = (int)(b < 1 ? 1 : b);
  const int wide = p.maxch > 2 ? 512 : kLnMaxBlocksWide;
  p.blocks_bwd = p.maxch > 1 && p.blocks > wide ? wide : p.blocks;
  return p;
}

}  // namespace

int ln_partial_rows(long rows, int C, int dtype) { return ln_plan(rows, C, dtype).blocks_bwd; }

void ln_fwd_launch(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, long rows, int C,
                   float eps, int dtype, hipStream_t s) {
  if (rows <= 0) return;
  const LnPlan p = ln_plan(rows, C, dtype);
#define DPA_LN_FWD(MC)                                                                                              \
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_fwd_kernel<T, MC, (MC <= 2 ? kLnRows : 1)>), dim3(p.blocks), dim3(256), 0, s, (const T*)x, w, \
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
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, MC, (MC == 1 ? kLnRows : (MC == 2 ? DPA_LN_BWD_R2 : 1))>), dim3(p.blocks_bwd), dim3(256), lds, s, (const T*)dy, \
                                               (const T*)x, w, mean, rstd, (T*)dx, partial, rows, C, p.G))
  if (p.maxch == 1) { DPA_LN_BWD(1); }
  else if (p.maxch == 2) { DPA_LN_BWD(2); }
  else if (p.maxch == 4) { DPA_LN_BWD(4); }
  else { DPA_LN_BWD(8); }
#undef DPA_LN_BWD
  hipLaunchKernelGGL(ln_fold_kernel, dim3(ceil_div(2 * C, 16)), dim3(256), 0, s, partial, p.blocks_bwd, C, dgamma, dbeta,
                     beta_w, beta_b);
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
