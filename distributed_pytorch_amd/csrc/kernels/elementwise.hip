// Memory-bound elementwise / layout kernels (gfx950): input normalisation, ReLU, residual add,
// dropout, casts, weight transposes.  All activation kernels move 16 bytes per lane.
// Replaces the ATen elementwise kernels and the DataPrefetcher's `.float().sub_().div_()`
// (reference imagenet_ddp_apex.py:339-340) -- SURVEY.md N-K15..N-K19.
#include "../common.h"
#include "kernels.h"

namespace dpa {
namespace {

inline int grid_for(long n, int per_block = 256) {
  long b = (n + per_block - 1) / per_block;
  if (b > g_stream_grid) b = g_stream_grid;
  return (int)(b < 1 ? 1 : b);
}

// uint8 [npix][cin] (NHWC) or [N][cin][hw] (NCHW) -> T [npix][cout], channels >= cin zero padded
template <typename T>
__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t* __restrict__ x, T* __restrict__ y, long npix,
                                                           int cin, int cout, int hw, int nchw,
                                                           const float* __restrict__ mean255,
                                                           const float* __restrict__ inv_std255) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const long stride = (long)gridDim.x * blockDim.x;
  const int chunks = cout / VEC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix * chunks; i += stride) {
    const long pix = i / chunks;
    const int c0 = (int)(i - pix * chunks) * VEC;
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = c0 + j;
      float v = 0.f;
      if (c < cin) {
        const uint8_t b = nchw ? x[((pix / hw) * cin + c) * (long)hw + pix % hw] : x[pix * cin + c];
        v = ((float)b - mean255[c]) * inv_std255[c];
      }
      o.set(j, v);
    }
    o.store(y + i * VEC);
  }
}

// Space-to-depth image layout for 7x7 stride-2 stems: [N][H][W][3] (or NCHW) -> [N][H/2][W/2][16] where channel
// (py*2 + px)*3 + c holds pixel (2Y+py, 2X+px), channels 12..15 are zero.  The stem then is a 4x4 stride-1
// convolution over 16 channels (K = 256 instead of 7*7*8 = 392 with channel padding) on a tensor half the size.
// U8: (x - 255*mean_c) / (255*std_c) (the prefetcher's normalisation); otherwise a plain conversion.
// The output rows may carry `padl` / `padr` zero pixels left / right of the W2 data pixels: with two on the left and
// (at least) one on the right the four horizontal taps of an output pixel are 64 CONTIGUOUS in-range channels, and
// the stem runs as a 4-tap x 64-channel problem on the aligned kernels (hip_ops._StemS2DFn).
template <typename TI, typename TO, bool U8>
__global__ __launch_bounds__(256) void to_s2d_kernel(const TI* __restrict__ x, TO* __restrict__ y, int NB, int H2, int W2,
                                                     int nchw, const float* __restrict__ mean255,
                                                     const float* __restrict__ inv_std255, int padl, int padr) {
  constexpr int VEC = 16 / (int)sizeof(TO);
  constexpr int CH = 16 / VEC;                       // chunks per s2d pixel
  const int WP = W2 + padl + padr;
  const long total = (long)NB * H2 * WP * CH;
  const long stride = (long)gridDim.x * blockDim.x;
  const int H = 2 * H2, W = 2 * W2;
  float mu[3] = {0.f, 0.f, 0.f}, is[3] = {1.f, 1.f, 1.f};
  if (U8) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { mu[c] = mean255[c]; is[c] = inv_std255[c]; }
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int h = (int)(i % CH);
    long pix = i / CH;
    const int X = (int)(pix % WP) - padl; pix /= WP;
    const int Y = (int)(pix % H2);
    const long n = pix / H2;
    const bool inside = X >= 0 && X < W2;
    Vec16<TO> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int ch = h * VEC + j;
      float v = 0.f;
      if (ch < 12 && inside) {
        const int q = ch / 3, c = ch - q * 3;
        const int iy = 2 * Y + (q >> 1), ix = 2 * X + (q & 1);
        const long src = nchw ? ((n * 3 + c) * H + iy) * W + ix : ((n * H + iy) * W + ix) * 3 + c;
        if constexpr (U8) v = ((float)x[src] - mu[c]) * is[c];
        else v = to_float(x[src]);
      }
      o.set(j, v);
    }
    o.store(y + i * VEC);
  }
}

// The prefetcher's case (uint8 NHWC source): one thread per s2d pixel, one workgroup per output row (no divisions).  The
// 12 source bytes of a pixel are two runs of 6 contiguous bytes (rows 2Y and 2Y+1, columns 2X and 2X+1), 2-byte aligned
// because W is even: three 16-bit loads each instead of twelve byte loads, and both 16-byte chunks leave in one go
// (0.43 -> 0.2x ms on 1024 x 224 x 224 images: the per-chunk kernel was bound by its byte loads and index arithmetic).
template <typename TO>
__global__ __launch_bounds__(128) void to_s2d_u8_rows_kernel(const uint8_t* __restrict__ x, TO* __restrict__ y, int H2,
                                                             int W2, const float* __restrict__ mean255,
                                                             const float* __restrict__ inv_std255, int padl, int padr) {
  static_assert(sizeof(TO) == 2, "16-bit outputs");
  const int WP = W2 + padl + padr;
  const long rowid = blockIdx.x;                       // n * H2 + Y
  const long n = rowid / H2;
  const int Y = (int)(rowid - n * H2);
  const int W = 2 * W2, H = 2 * H2;
  float mu[3], is[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { mu[c] = mean255[c]; is[c] = inv_std255[c]; }
  const uint8_t* r0 = x + ((n * H + 2 * Y) * W) * 3;
  const uint8_t* r1 = r0 + (long)W * 3;
  for (int xp = threadIdx.x; xp < WP; xp += blockDim.x) {
    const int X = xp - padl;
    Vec16<TO> o0, o1;
    o0.raw = make_uint4(0u, 0u, 0u, 0u);
    o1.raw = make_uint4(0u, 0u, 0u, 0u);
    if (X >= 0 && X < W2) {
      const uint16_t* p0 = reinterpret_cast<const uint16_t*>(r0 + 6 * X);
      const uint16_t* p1 = reinterpret_cast<const uint16_t*>(r1 + 6 * X);
      const uint32_t a0 = p0[0], a1 = p0[1], a2 = p0[2], b0 = p1[0], b1 = p1[1], b2 = p1[2];
      const uint32_t v[12] = {a0 & 255u, a0 >> 8, a1 & 255u, a1 >> 8, a2 & 255u, a2 >> 8,
                              b0 & 255u, b0 >> 8, b1 & 255u, b1 >> 8, b2 & 255u, b2 >> 8};
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) o0.set(ch, ((float)v[ch] - mu[ch % 3]) * is[ch % 3]);
#pragma unroll
      for (int ch = 8; ch < 12; ++ch) o1.set(ch - 8, ((float)v[ch] - mu[ch % 3]) * is[ch % 3]);
    }
    TO* dst = y + (rowid * WP + xp) * 16;
    o0.store(dst);
    o1.store(dst + 8);
  }
}

// float-ish NCHW -> NHWC with channel padding (LDS-free: C is tiny (3) for images)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const TI* __restrict__ x, TO* __restrict__ y, int NB, int C,
                                                           int HW, int cpad) {
  const long total = (long)NB * HW * cpad;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cpad);
    const long pix = i / cpad;
    const long n = pix / HW, p = pix % HW;
    y[i] = from_float<TO>(c < C ? to_float(x[(n * C + c) * HW + p]) : 0.f);
  }
}

template <typename T, int OP>  // 0: relu, 1: relu_bwd(y, dy), 2: add_relu(a, b), 3: a + b
__global__ __launch_bounds__(256) void ew_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y,
                                                 long nchunks) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
    Vec16<T> va, vb, vo;
    va.load(a + i * VEC);
    if (OP != 0) vb.load(b + i * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float r;
      if (OP == 0) r = fmaxf(va.get(j), 0.f);
      else if (OP == 1) r = va.get(j) > 0.f ? vb.get(j) : 0.f;
      else if (OP == 2) r = fmaxf(va.get(j) + vb.get(j), 0.f);
      else r = va.get(j) + vb.get(j);
      vo.set(j, r);
    }
    vo.store(y + i * VEC);
  }
}
template <typename T, int OP>
__global__ void ew_tail_kernel(const T* a, const T* b, T* y, long begin, long n) {
  const long i = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r;
  if (OP == 0) r = fmaxf(to_float(a[i]), 0.f);
  else if (OP == 1) r = to_float(a[i]) > 0.f ? to_float(b[i]) : 0.f;
  else if (OP == 2) r = fmaxf(to_float(a[i]) + to_float(b[i]), 0.f);
  else r = to_float(a[i]) + to_float(b[i]);
  y[i] = from_float<T>(r);
}

template <typename T, int OP>
void ew_launch(const void* a, const void* b, void* y, long n, hipStream_t s) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const long nchunks = n / VEC;
  if (nchunks > 0)
    hipLaunchKernelGGL((ew_kernel<T, OP>), dim3(grid_for(nchunks)), dim3(256), 0, s, (const T*)a, (const T*)b, (T*)y,
                       nchunks);
  const long rem = n - nchunks * VEC;
  if (rem > 0)
    hipLaunchKernelGGL((ew_tail_kernel<T, OP>), dim3(1), dim3(64), 0, s, (const T*)a, (const T*)b, (T*)y,
                       nchunks * VEC, n);
  DPA_CHECK_HIP(hipGetLastError());
}


// ---- smooth activations (EfficientNet / MobileNetV3 / squeeze-excitation gates) -------------------------------
// act: 1 sigmoid, 2 SiLU (x * sigmoid(x)), 3 hard-sigmoid (relu6(x + 3) / 6), 4 hard-swish (x * hard-sigmoid(x)),
// 5 GELU (exact, erf).
// Backward recomputes the derivative from the saved INPUT, so no activation-sized mask / output is kept.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void act_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out,
                                                  long nchunks, int act) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
    Vec16<T> vx, vg, vo;
    vx.load(x + i * VEC);
    if (BWD) vg.load(dy + i * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float v = vx.get(j);
      vo.set(j, BWD ? vg.get(j) * act_slope(act, v) : act_value(act, v));
    }
    vo.store(out + i * VEC);
  }
}
template <typename T, bool BWD>
__global__ void act_tail_kernel(const T* x, const T* dy, T* out, long begin, long n, int act) {
  const long i = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = to_float(x[i]);
  out[i] = from_float<T>(BWD ? to_float(dy[i]) * act_slope(act, v) : act_value(act, v));
}

template <typename T, bool BWD>
void act_launch_t(const void* x, const void* dy, void* out, long n, int act, hipStream_t s) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const long nchunks = n / VEC;
  if (nchunks > 0)
    hipLaunchKernelGGL((act_kernel<T, BWD>), dim3(grid_for(nchunks)), dim3(256), 0, s, (const T*)x, (const T*)dy, (T*)out,
                       nchunks, act);
  const long rem = n - nchunks * VEC;
  if (rem > 0)
    hipLaunchKernelGGL((act_tail_kernel<T, BWD>), dim3(1), dim3(64), 0, s, (const T*)x, (const T*)dy, (T*)out,
                       nchunks * VEC, n, act);
  DPA_CHECK_HIP(hipGetLastError());
}

// ---- per-(sample, channel) scale: squeeze-excitation gate, stochastic depth ("row" mode: s constant over c) -------
// y[n][p][c] = x[n][p][c] * s[n][c]                                         (s fp32 [N][C])
template <typename T>
__global__ __launch_bounds__(256) void se_scale_kernel(const T* __restrict__ x, const float* __restrict__ s,
                                                       T* __restrict__ y, long total, int HW, int chunks,
                                                       const float* __restrict__ add) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long pix = i / chunks;
    const int ch = (int)(i - pix * chunks);
    const long n = pix / HW;
    const float* sp = s + n * (long)chunks * VEC + ch * VEC;
    Vec16<T> v, o;
    v.load(x + i * VEC);
    if (add != nullptr) {            // y = x * s[n][c] + add[n][c] (the SE block's dx: dy * gate + d(mean) / HW)
      const float* ap = add + n * (long)chunks * VEC + ch * VEC;
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.set(j, v.get(j) * sp[j] + ap[j]);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.set(j, v.get(j) * sp[j]);
    }
    o.store(y + i * VEC);
  }
}

// backward of the gate: dx = dy * s,  ds[n][c] = sum_p dy[n][p][c] * x[n][p][c].  One workgroup owns sample n and
// 16 chunks of channels; 16 pixel rows of the workgroup stride over H*W and meet in LDS (fixed order: deterministic).
template <typename T>
__global__ __launch_bounds__(256) void se_scale_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const float* __restrict__ s, T* __restrict__ dx,
                                                           float* __restrict__ ds, int HW, int chunks) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int CL = 16, PR = 16;
  __shared__ float red[PR][CL][VEC + 1];
  const int cl = threadIdx.x % CL, pr = threadIdx.x / CL;
  const int ch = blockIdx.x * CL + cl;
  const long n = blockIdx.y;
  const int C = chunks * VEC;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (ch < chunks) {
    float sv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) sv[j] = s[n * C + ch * VEC + j];
    for (int p = pr; p < HW; p += PR) {
      const long off = ((n * HW + p) * (long)chunks + ch) * VEC;
      Vec16<T> g, v, o;
      g.load(dy + off);
      v.load(x + off);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float gj = g.get(j);
        acc[j] += gj * v.get(j);
        o.set(j, gj * sv[j]);
      }
      if (dx != nullptr) o.store(dx + off);
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[pr][cl][j] = acc[j];
  __syncthreads();
  if (ds != nullptr && pr == 0 && ch < chunks) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float t = 0.f;
      for (int r = 0; r < PR; ++r) t += red[r][cl][j];
      ds[n * C + ch * VEC + j] = t;
    }
  }
}

// counter-based RNG (splitmix64 finaliser over (seed, offset + index)): stateless, graph-replayable
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ mask, long n, float p, uint64_t seed,
                                                          uint64_t offset, const uint64_t* __restrict__ counter) {
  // the stream position may live on the device (counter) so a captured hipGraph draws fresh masks per replay
  if (counter != nullptr) offset += *counter;
  const float scale = 1.f / (1.f - p);
  const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool keep = hash_u32(seed, offset + (uint64_t)i) >= thr;
    mask[i] = keep ? 1 : 0;
    y[i] = from_float<T>(keep ? to_float(x[i]) * scale : 0.f);
  }
}
__global__ void counter_add_kernel(uint64_t* counter, uint64_t n) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *counter += n;
}
template <typename T>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                          T* __restrict__ dx, long n, float p) {
  const float scale = 1.f / (1.f - p);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dx[i] = from_float<T>(mask[i] ? to_float(dy[i]) * scale : 0.f);
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, long n) {
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i + j < n) y[i + j] = from_float<TO>(to_float(x[i + j]));
  }
}

// [K][T][C] -> [C][T][K]; 32x32 LDS tile transpose over (K, C) per tap
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_ktc_kernel(const TI* __restrict__ w, TO* __restrict__ wt, int K, int T,
                                                            int C) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, c = c0 + tx;
    tile[r][tx] = (k < K && c < C) ? to_float(w[((size_t)k * T + t) * C + c]) : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, k = k0 + tx;
    if (c < C && k < K) wt[((size_t)c * T + t) * K + k] = from_float<TO>(tile[tx][r]);
  }
}

template <typename TX>
__global__ __launch_bounds__(256) void axpby_kernel(const TX* __restrict__ x, float* __restrict__ y, long n, float alpha,
                                                    float beta) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = alpha * to_float(x[i]);
    y[i] = beta == 0.f ? v : v + beta * y[i];
  }
}

// dst[r][0..cols) = src[r][0..cols) for row-strided 2-D views, 16 bytes per lane (channel concat / split in NHWC)
__global__ __launch_bounds__(256) void copy_rows_kernel(const uint4* __restrict__ src, long ld_src, uint4* __restrict__ dst,
                                                        long ld_dst, long rows, int chunks) {
  const long total = rows * chunks;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long r = i / chunks;
    const int c = (int)(i - r * chunks);
    dst[r * ld_dst + c] = src[r * ld_src + c];
  }
}

// dst[n][oy][ox][:] = src[n][oy * s][ox * s][:] for NHWC tensors, 16 bytes per lane: the pixels a stride-s 1x1 convolution reads,
// gathered once for the dense GEMMs of the folded downsample branch (ATen's strided copy ran at a third of the bandwidth)
__global__ __launch_bounds__(256) void subsample_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long total,
                                                        int OH, int OW, int IH, int IW, int s, int chunks) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long pix = i / chunks;
    const int c = (int)(i - pix * chunks);
    const long t = pix / OW;
    const int ox = (int)(pix - t * OW);
    const long n = t / OH;
    const int oy = (int)(t - n * OH);
    dst[i] = src[((n * IH + (long)oy * s) * IW + (long)ox * s) * chunks + c];
  }
}

// ShuffleNetV2 channel shuffle (groups = 2) fused with the channel split that always follows it.  A stage's activation
// lives as a PAIR of tensors (left, right), each [rows][P] with `bf` real channels and P - bf zero pad channels (P = bf
// rounded up to 16-byte vectors: ShuffleNetV2 x1.0 / x2.0 have 58 / 122-channel halves).
//   forward : out = shuffle(cat(A, B)), out[2i] = A[i], out[2i+1] = B[i]; (Lo, Ro) = (out[:bf], out[bf:])
//   backward: the inverse permutation of (dLo, dRo) back to (dA, dB)
// One thread per 16-byte output chunk; the scattered 2/4-byte reads hit rows that are resident in L2.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void shuffle_pair_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                           T* __restrict__ o0, T* __restrict__ o1, long rows, int bf,
                                                           int P) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int chunks = P / VEC;
  const long total = rows * chunks * 2;
  const int h = bf / 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long r2 = i / chunks;
    const int ck = (int)(i - r2 * chunks);
    const int which = (int)(r2 & 1);          // output tensor 0 / 1
    const long r = r2 >> 1;
    const T* ra = a + r * P;
    const T* rb = b + r * P;
    T v[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = ck * VEC + j;
      T t = from_float<T>(0.f);
      if (c < bf) {
        if (!BWD) {            // out channel (which * bf + c) of the shuffled tensor
          const int src = (c >> 1) + which * h;
          t = (c & 1) ? rb[src] : ra[src];
        } else {               // which 0: dA[c], which 1: dB[c]; shuffled index 2c + which lives in (a | b) = (dLo | dRo)
          const int s = 2 * c + which;
          t = s < bf ? ra[s] : rb[s - bf];
        }
      }
      v[j] = t;
    }
    T* dst = (which ? o1 : o0) + r * P + ck * VEC;
    uint4 w;
    __builtin_memcpy(&w, v, 16);
    *reinterpret_cast<uint4*>(dst) = w;
  }
}

#define DPA_DISPATCH_2(di, dout, ...)                                                        \
  DPA_DISPATCH_DTYPE(di, { using TI = T; DPA_DISPATCH_DTYPE(dout, { using TO = T; __VA_ARGS__; }) })

}  // namespace

void normalize_u8_launch(const uint8_t* x, void* y, long npix, int cin, int cout, int hw, int nchw,
                         const float* mean255, const float* inv_std255, int dtype, hipStream_t s) {
  if (cout % (dtype == kF32 ? 4 : 8) != 0) throw std::invalid_argument("normalize_u8: cout must fill 16-byte vectors");
  if (npix <= 0) return;
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(normalize_u8_kernel<T>, dim3(grid_for(npix * (cout / (16 / (int)sizeof(T))))),
                                               dim3(256), 0, s, x, (T*)y, npix, cin, cout, hw, nchw, mean255,
                                               inv_std255));
  DPA_CHECK_HIP(hipGetLastError());
}

void to_s2d_launch(const void* x, void* y, int NB, int H2, int W2, int nchw, int in_dtype, const float* mean255,
                   const float* inv_std255, int out_dtype, hipStream_t s, int padl, int padr) {
  const long total = (long)NB * H2 * (W2 + padl + padr) * (out_dtype == kF32 ? 4 : 2);
  if (total <= 0) return;
  if (in_dtype < 0) {      // uint8 source
    if (mean255 == nullptr || inv_std255 == nullptr) throw std::invalid_argument("to_s2d: uint8 input needs mean / std");
    if (!nchw && out_dtype != kF32 && (long)NB * H2 < 2147483647l && ((uintptr_t)x & 1) == 0) {
      if (out_dtype == kBF16)
        hipLaunchKernelGGL(to_s2d_u8_rows_kernel<bf16_t>, dim3((unsigned)((long)NB * H2)), dim3(128), 0, s,
                           (const uint8_t*)x, (bf16_t*)y, H2, W2, mean255, inv_std255, padl, padr);
      else
        hipLaunchKernelGGL(to_s2d_u8_rows_kernel<f16_t>, dim3((unsigned)((long)NB * H2)), dim3(128), 0, s,
                           (const uint8_t*)x, (f16_t*)y, H2, W2, mean255, inv_std255, padl, padr);
      DPA_CHECK_HIP(hipGetLastError());
      return;
    }
    DPA_DISPATCH_DTYPE(out_dtype, hipLaunchKernelGGL((to_s2d_kernel<uint8_t, T, true>), dim3(grid_for(total)), dim3(256),
                                                     0, s, (const uint8_t*)x, (T*)y, NB, H2, W2, nchw, mean255,
                                                     inv_std255, padl, padr));
  } else {
    DPA_DISPATCH_2(in_dtype, out_dtype,
                   hipLaunchKernelGGL((to_s2d_kernel<TI, TO, false>), dim3(grid_for(total)), dim3(256), 0, s,
                                      (const TI*)x, (TO*)y, NB, H2, W2, nchw, nullptr, nullptr, padl, padr));
  }
  DPA_CHECK_HIP(hipGetLastError());
}

void nchw_to_nhwc_launch(const void* x, void* y, int NB, int C, int HW, int cpad, int in_dtype, int out_dtype,
                         hipStream_t s) {
  const long total = (long)NB * HW * cpad;
  if (total <= 0) return;
  DPA_DISPATCH_2(in_dtype, out_dtype,
                 hipLaunchKernelGGL((nchw_to_nhwc_kernel<TI, TO>), dim3(grid_for(total)), dim3(256), 0, s,
                                    (const TI*)x, (TO*)y, NB, C, HW, cpad));
  DPA_CHECK_HIP(hipGetLastError());
}

void relu_fwd_launch(const void* x, void* y, long n, int dtype, hipStream_t s) {
  DPA_DISPATCH_DTYPE(dtype, (ew_launch<T, 0>(x, nullptr, y, n, s)));
}
void relu_bwd_launch(const void* yv, const void* dy, void* dx, long n, int dtype, hipStream_t s) {
  DPA_DISPATCH_DTYPE(dtype, (ew_launch<T, 1>(yv, dy, dx, n, s)));
}
void add_relu_fwd_launch(const void* a, const void* b, void* y, long n, int dtype, hipStream_t s) {
  DPA_DISPATCH_DTYPE(dtype, (ew_launch<T, 2>(a, b, y, n, s)));
}

void add_fwd_launch(const void* a, const void* b, void* y, long n, int dtype, hipStream_t s) {
  DPA_DISPATCH_DTYPE(dtype, (ew_launch<T, 3>(a, b, y, n, s)));
}

void act_fwd_launch(const void* x, void* y, long n, int act, int dtype, hipStream_t s) {
  if (act < 1 || act > 5) throw std::invalid_argument("act: unknown activation code");
  if (n <= 0) return;
  DPA_DISPATCH_DTYPE(dtype, (act_launch_t<T, false>(x, nullptr, y, n, act, s)));
}
void act_bwd_launch(const void* x, const void* dy, void* dx, long n, int act, int dtype, hipStream_t s) {
  if (act < 1 || act > 5) throw std::invalid_argument("act: unknown activation code");
  if (n <= 0) return;
  DPA_DISPATCH_DTYPE(dtype, (act_launch_t<T, true>(x, dy, dx, n, act, s)));
}

void se_scale_fwd_launch(const void* x, const float* sc, void* y, long NB, int HW, int C, int dtype, hipStream_t s,
                         const float* add) {
  const int vec = dtype == kF32 ? 4 : 8;
  if (C % vec != 0) throw std::invalid_argument("se_scale: C must fill 16-byte vectors");
  const long total = NB * HW * (C / vec);
  if (total <= 0) return;
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(se_scale_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)x, sc,
                                               (T*)y, total, HW, C / vec, add));
  DPA_CHECK_HIP(hipGetLastError());
}
void se_scale_bwd_launch(const void* dy, const void* x, const float* sc, void* dx, float* ds, long NB, int HW, int C,
                         int dtype, hipStream_t s) {
  const int vec = dtype == kF32 ? 4 : 8;
  if (C % vec != 0) throw std::invalid_argument("se_scale: C must fill 16-byte vectors");
  if (NB <= 0 || HW <= 0) return;
  if (NB > 65535) throw std::invalid_argument("se_scale: batch exceeds grid.y");
  const int chunks = C / vec;
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(se_scale_bwd_kernel<T>, dim3(ceil_div(chunks, 16), (unsigned)NB), dim3(256),
                                               0, s, (const T*)dy, (const T*)x, sc, (T*)dx, ds, HW, chunks));
  DPA_CHECK_HIP(hipGetLastError());
}

void dropout_fwd_launch(const void* x, void* y, uint8_t* mask, long n, float p, uint64_t seed, uint64_t offset,
                        uint64_t* counter, int dtype, hipStream_t s) {
  if (n <= 0) return;
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(dropout_fwd_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, (const T*)x,
                                               (T*)y, mask, n, p, seed, offset, (const uint64_t*)counter));
  if (counter != nullptr)
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, s, counter, (uint64_t)n);
  DPA_CHECK_HIP(hipGetLastError());
}
void dropout_bwd_launch(const void* dy, const uint8_t* mask, void* dx, long n, float p, int dtype, hipStream_t s) {
  if (n <= 0) return;
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(dropout_bwd_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, (const T*)dy,
                                               mask, (T*)dx, n, p));
  DPA_CHECK_HIP(hipGetLastError());
}

void cast_launch(const void* x, int in_dtype, void* y, int out_dtype, long n, hipStream_t s) {
  if (n <= 0) return;
  DPA_DISPATCH_2(in_dtype, out_dtype,
                 hipLaunchKernelGGL((cast_kernel<TI, TO>), dim3(grid_for(n, 1024)), dim3(256), 0, s, (const TI*)x,
                                    (TO*)y, n));
  DPA_CHECK_HIP(hipGetLastError());
}

void transpose_ktc_launch(const void* w, void* wt, int K, int ntaps, int C, int in_dtype, int out_dtype,
                          hipStream_t s) {
  if (K <= 0 || ntaps <= 0 || C <= 0) return;
  if (ntaps > 65535) throw std::invalid_argument("transpose_ktc: too many taps");
  dim3 grid(ceil_div(C, 32), ceil_div(K, 32), ntaps);
  DPA_DISPATCH_2(in_dtype, out_dtype,
                 hipLaunchKernelGGL((transpose_ktc_kernel<TI, TO>), grid, dim3(256), 0, s, (const TI*)w, (TO*)wt, K,
                                    ntaps, C));
  DPA_CHECK_HIP(hipGetLastError());
}

void axpby_launch(const void* x, int x_dtype, float* y, long n, float alpha, float beta, hipStream_t s) {
  if (n <= 0) return;
  DPA_DISPATCH_DTYPE(x_dtype, hipLaunchKernelGGL(axpby_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, (const T*)x, y, n,
                                                 alpha, beta));
  DPA_CHECK_HIP(hipGetLastError());
}

void copy_rows_launch(const void* src, long ld_src_bytes, void* dst, long ld_dst_bytes, long rows, long row_bytes,
                      hipStream_t s) {
  if (rows <= 0 || row_bytes <= 0) return;
  if (row_bytes % 16 || ld_src_bytes % 16 || ld_dst_bytes % 16 || ((uintptr_t)src % 16) || ((uintptr_t)dst % 16))
    throw std::invalid_argument("copy_rows: rows, strides and bases must be multiples of 16 bytes");
  const int chunks = (int)(row_bytes / 16);
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(rows * chunks)), dim3(256), 0, s, (const uint4*)src,
                     ld_src_bytes / 16, (uint4*)dst, ld_dst_bytes / 16, rows, chunks);
  DPA_CHECK_HIP(hipGetLastError());
}

void subsample_launch(const void* src, void* dst, long NB, int IH, int IW, long row_bytes, int s_, hipStream_t s) {
  if (row_bytes % 16 || ((uintptr_t)src % 16) || ((uintptr_t)dst % 16) || s_ < 1)
    throw std::invalid_argument("subsample: pixels must be multiples of 16 bytes");
  const int OH = (IH + s_ - 1) / s_, OW = (IW + s_ - 1) / s_, chunks = (int)(row_bytes / 16);
  const long total = NB * OH * OW * chunks;
  if (total <= 0) return;
  hipLaunchKernelGGL(subsample_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, total, OH, OW,
                     IH, IW, s_, chunks);
  DPA_CHECK_HIP(hipGetLastError());
}

void shuffle_pair_launch(const void* a, const void* b, void* o0, void* o1, long rows, int bf, int P, int backward,
                         int dtype, hipStream_t s) {
  const int vec = dtype == kF32 ? 4 : 8;
  if (rows <= 0) return;
  if (P % vec || bf > P || bf <= 0 || (bf & 1)) throw std::invalid_argument("shuffle_pair: bad channel counts");
  const long total = rows * (P / vec) * 2;
  if (backward) {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((shuffle_pair_kernel<T, true>), dim3(grid_for(total)), dim3(256), 0, s,
                                                 (const T*)a, (const T*)b, (T*)o0, (T*)o1, rows, bf, P));
  } else {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((shuffle_pair_kernel<T, false>), dim3(grid_for(total)), dim3(256), 0, s,
                                                 (const T*)a, (const T*)b, (T*)o0, (T*)o1, rows, bf, P));
  }
  DPA_CHECK_HIP(hipGetLastError());
}

void fill_launch(void* p, long nbytes, int value, hipStream_t s) {
  if (nbytes > 0) DPA_CHECK_HIP(hipMemsetAsync(p, value, (size_t)nbytes, s));
}

int g_stream_grid = 256 * 16;     // measured 1280 / 2560 / 4096 / 5120 on seven families: 4096 best or equal (profiles/r2/stream_grid_ab.txt)
void set_stream_grid(int blocks) { g_stream_grid = blocks > 0 ? blocks : 256 * 16; }

}  // namespace dpa
