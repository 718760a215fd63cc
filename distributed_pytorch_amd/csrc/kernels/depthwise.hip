// Direct depthwise convolution (groups == channels) for NHWC activations on gfx950: MobileNet / MNASNet / EfficientNet
// 3x3 and 5x5, ConvNeXt 7x7.  A depthwise filter has no reduction over channels, so there is nothing for MFMA to do:
// the block-diagonal tiles of the grouped implicit-GEMM path multiply 63/64 structural zeros.  These kernels are plain
// bandwidth kernels instead -- 16 bytes of channels per lane, filter taps from LDS, fp32 accumulation:
//
//   dw_conv_kernel<T, BWD=false>  y[n,oy,ox,c]  = sum_{r,s} x[n, oy*sh-ph+r, ox*sw-pw+s, c] * w[c,r,s] (+bias)(+ReLU),
//                                  optionally with the per-channel (sum y, sum y^2) partial rows BatchNorm consumes;
//   dw_conv_kernel<T, BWD=true>   dx[n,iy,ix,c] = sum_{r,s} dy[n,(iy+ph-r)/sh,(ix+pw-s)/sw,c] * w[c,r,s]  (gather form:
//                                  no zero insertion, no atomics; taps whose source is off the stride lattice are masked);
//   dw_wgrad_kernel<T, KW>        dw[c,r,s]     = sum_{n,oy,ox} dy[n,oy,ox,c] * x[n, oy*sh-ph+r, ox*sw-pw+s, c]
//                                  one filter ROW per lane (KW x 16-byte accumulators), lanes = chunks x rows x pixels,
//                                  pixel lanes meet in LDS, one partial row per workgroup, folded by slab_reduce.
//
// The filter is read straight from the parameter (fp32 master or compute dtype, memory [C][KH][KW]) and rounded to the
// compute dtype while it is staged in LDS, so results equal the implicit-GEMM path's up to accumulation order.
#include "../common.h"
#include "kernels.h"

namespace dpa {
namespace {

struct DwGeom {
  int NB, IH, IW, C, OH, OW, KH, KW, sh, sw, ph, pw;
};

constexpr int kDwMaxTaps = 49;
constexpr int kDwSlab = 8;          // channel chunks per workgroup

// source coordinate of tap `t` for written coordinate `p`: fprop p*st - pad + t; backward-data (p + pad - t) / st when
// that lands on the stride lattice
template <bool BWD>
__device__ __forceinline__ bool dw_src(int p, int t, int st, int pad, int extent, int& src) {
  if (BWD) {
    const int u = p + pad - t;
    if (st == 1) src = u;
    else if (st == 2) src = u >> 1;
    else src = u / st;
    return u >= 0 && src * st == u && src < extent;
  }
  src = p * st - pad + t;
  return src >= 0 && src < extent;
}

// KHT > 0: the whole KHT x KWT window is loaded up front (3x3, 5x5); KHT == 0: one filter row at a time (7x7, odd shapes)
template <typename T, typename TW, bool BWD, int KWT, int KHT>
__global__ __launch_bounds__(256) void dw_conv_kernel(const T* __restrict__ in, const TW* __restrict__ w,
                                                      const float* __restrict__ bias, T* __restrict__ out,
                                                      float* __restrict__ stats, DwGeom g, int CSL, int relu,
                                                      FastDiv div_w, FastDiv div_h) {
  constexpr int VEC = 16 / (int)sizeof(T);
  __shared__ float wl[kDwMaxTaps][kDwSlab * VEC];
  __shared__ float red[3][256][VEC + 1];      // shifted sum, shifted sum of squares, shift (+ pixel count in slot VEC)
  const int chunks = g.C / VEC;
  const int taps = g.KH * g.KW;
  const int cl = threadIdx.x % CSL, pl = threadIdx.x / CSL, PXL = 256 / CSL;
  const int ch = blockIdx.y * CSL + cl;
  const int c_slab = blockIdx.y * CSL * VEC;
  for (int i = threadIdx.x; i < taps * CSL * VEC; i += 256) {
    const int t = i / (CSL * VEC), cc = i - t * (CSL * VEC);
    const int c = c_slab + cc;
    wl[t][cc] = c < g.C ? to_float(from_float<T>(to_float(w[(size_t)c * taps + t]))) : 0.f;
  }
  __syncthreads();
  // out pixels: fprop -> (n, oy, ox) over OH x OW; backward-data -> (n, iy, ix) over IH x IW
  const int PH = BWD ? g.IH : g.OH, PW = BWD ? g.IW : g.OW;     // extents of the tensor being written
  const int SH = BWD ? g.OH : g.IH, SW = BWD ? g.OW : g.IW;     // extents of the tensor being read
  const uint32_t npix = (uint32_t)g.NB * PH * PW;              // < 2^31, checked by the launcher
  // statistics: per thread sums of (v - k), (v - k)^2 with k = the thread's first output (see bn.hip: the slab holds
  // Chan partials (sum, M2, rows), never a raw sum of squares)
  float s1[VEC], s2[VEC], kk[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = kk[j] = 0.f;
  float npx = 0.f;
  const bool active = ch < chunks;
  const int st_y = g.sh, st_x = g.sw;
  for (uint32_t pix = blockIdx.x * PXL + pl; pix < npix; pix += gridDim.x * PXL) {
    if (!active) continue;
    const uint32_t q = fdiv(pix, div_w);
    const int px = (int)(pix - q * PW);
    const uint32_t n = fdiv(q, div_h);
    const int py = (int)(q - n * PH);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = (!BWD && bias != nullptr) ? bias[ch * VEC + j] : 0.f;
    const T* base = in + (long)n * SH * SW * g.C + ch * VEC;
    if constexpr (KHT > 0) {
      // every tap of the window in flight at once: loads from clamped addresses, masks applied at the multiply
      Vec16<T> v[KHT][KWT];
      bool ok[KHT][KWT];
      int sx[KWT];
      bool okx[KWT];
#pragma unroll
      for (int s = 0; s < KWT; ++s) okx[s] = dw_src<BWD>(px, s, st_x, g.pw, SW, sx[s]);
#pragma unroll
      for (int r = 0; r < KHT; ++r) {
        int sy;
        const bool oky = dw_src<BWD>(py, r, st_y, g.ph, SH, sy);
        const T* row = base + (long)(oky ? sy : 0) * SW * g.C;
#pragma unroll
        for (int s = 0; s < KWT; ++s) {
          ok[r][s] = oky && okx[s];
          v[r][s].load(row + (long)(okx[s] ? sx[s] : 0) * g.C);
        }
      }
#pragma unroll
      for (int r = 0; r < KHT; ++r)
#pragma unroll
        for (int s = 0; s < KWT; ++s) {
          if (ok[r][s]) {
            const float* wt = &wl[r * KWT + s][cl * VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] += v[r][s].get(j) * wt[j];
          }
        }
    } else {
      for (int r = 0; r < g.KH; ++r) {
        int sy;
        if (!dw_src<BWD>(py, r, st_y, g.ph, SH, sy)) continue;
        const T* row = base + (long)sy * SW * g.C;
        // the whole filter row in flight at once (a tap loop with data-dependent skips would serialise one
        // memory latency per tap)
        Vec16<T> v[KWT];
        bool okx[KWT];
#pragma unroll
        for (int s = 0; s < KWT; ++s) {
          int sx;
          okx[s] = s < g.KW && dw_src<BWD>(px, s, st_x, g.pw, SW, sx);
          v[s].load(row + (long)(okx[s] ? sx : 0) * g.C);
        }
#pragma unroll
        for (int s = 0; s < KWT; ++s) {
          if (okx[s]) {
            const float* wt = &wl[r * g.KW + s][cl * VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] += v[s].get(j) * wt[j];
          }
        }
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float v = acc[j];
      if (relu) v = fmaxf(v, 0.f);
      o.set(j, v);
      if (!BWD && stats != nullptr) {
        const float rv = o.get(j);           // what BatchNorm will read back
        if (npx == 0.f) kk[j] = rv;
        const float f = rv - kk[j];
        s1[j] += f;
        s2[j] += f * f;
      }
    }
    npx += 1.f;
    o.store(out + (long)pix * g.C + ch * VEC);
  }
  if (!BWD && stats != nullptr) {
    // per thread: sums of (v - k_t), (v - k_t)^2 and its shift k_t; the pixel lanes are re-based on the shift of
    // lane 0 (a sample as well, so every term stays at the scale of the spread) and added in fp32
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      red[0][threadIdx.x][j] = s1[j];
      red[1][threadIdx.x][j] = s2[j];
      red[2][threadIdx.x][j] = kk[j];
    }
    red[0][threadIdx.x][VEC] = npx;
    __syncthreads();
    if (pl == 0 && active) {
      float N = 0.f;
      for (int p = 0; p < PXL; ++p) N += red[0][p * CSL + cl][VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float K = red[2][cl][j];          // lane 0 always has a pixel when the block has any
        float S = 0.f, Q = 0.f;
        for (int p = 0; p < PXL; ++p) {
          const int t = p * CSL + cl;
          const float n = red[0][t][VEC], a = red[0][t][j], d = red[2][t][j] - K;
          S += a + n * d;
          Q += red[1][t][j] + 2.f * d * a + n * d * d;
        }
        const float m2 = N > 0.f ? Q - S * S / N : 0.f;
        stats[((size_t)blockIdx.x * 2 + 0) * g.C + ch * VEC + j] = S + N * K;
        stats[((size_t)blockIdx.x * 2 + 1) * g.C + ch * VEC + j] = fmaxf(m2, 0.f);
      }
      if (blockIdx.y == 0 && cl == 0) stats[(size_t)gridDim.x * 2 * g.C + blockIdx.x] = (float)N;
    }
  }
}

// lanes: cl (channel chunk of the slab) fastest, then filter row r, then pixel lane pl
template <typename T, int KW>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                       float* __restrict__ partial, DwGeom g, int CSL, int PL,
                                                       FastDiv div_w, FastDiv div_h) {
  constexpr int VEC = 16 / (int)sizeof(T);
  __shared__ float red[256][VEC + 1];
  const int chunks = g.C / VEC;
  const int taps = g.KH * g.KW;
  const int lanes = CSL * g.KH;
  const int cl = threadIdx.x % CSL, r = (threadIdx.x / CSL) % g.KH, pl = threadIdx.x / lanes;
  const int ch = blockIdx.y * CSL + cl;
  const bool active = pl < PL && ch < chunks;
  const uint32_t npix = (uint32_t)g.NB * g.OH * g.OW;          // < 2^31, checked by the launcher
  float acc[KW][VEC];
#pragma unroll
  for (int s = 0; s < KW; ++s)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[s][j] = 0.f;
  if (active) {
    // two pixels per iteration: 2 * (1 + KW) independent 16-byte loads in flight per lane
    const uint32_t step = gridDim.x * PL;
    for (uint32_t pix0 = blockIdx.x * PL + pl; pix0 < npix; pix0 += 2 * step) {
      Vec16<T> gv[2], xv[2][KW];
      bool ok[2][KW];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t pix = pix0 + u * step;
        const bool live = pix < npix;
        const uint32_t pc = live ? pix : pix0;
        const uint32_t q = fdiv(pc, div_w);
        const int ox = (int)(pc - q * g.OW);
        const uint32_t n = fdiv(q, div_h);
        const int oy = (int)(q - n * g.OH);
        const int iy = oy * g.sh - g.ph + r;
        const bool oky = live && iy >= 0 && iy < g.IH;
        gv[u].load(dy + (long)pc * g.C + ch * VEC);
        const T* row = x + (((long)n * g.IH + (oky ? iy : 0)) * g.IW) * g.C + ch * VEC;
        const int ix0 = ox * g.sw - g.pw;
#pragma unroll
        for (int s = 0; s < KW; ++s) {           // clamped addresses, masks at the multiply
          const int ix = ix0 + s;
          ok[u][s] = oky && s < g.KW && ix >= 0 && ix < g.IW;
          xv[u][s].load(row + (long)(ok[u][s] ? ix : 0) * g.C);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < KW; ++s) {
          if (ok[u][s]) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[s][j] += gv[u].get(j) * xv[u][s].get(j);
          }
        }
    }
  }
  // pixel lanes meet in LDS, one tap column at a time (fixed order)
  float* prow = partial + (size_t)blockIdx.x * g.C * taps;
#pragma unroll
  for (int s = 0; s < KW; ++s) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[threadIdx.x][j] = acc[s][j];
    __syncthreads();
    if (pl == 0 && ch < chunks && s < g.KW) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float t = 0.f;
        for (int p = 0; p < PL; ++p) t += red[p * lanes + r * CSL + cl][j];
        prow[(size_t)(ch * VEC + j) * taps + r * g.KW + s] = t;
      }
    }
    __syncthreads();
  }
}

int slab_lanes(int chunks) {
  int c = 1;
  while (c < chunks && c < kDwSlab) c <<= 1;
  return c;
}

void check_geom(const DwGeom& g, int dtype) {
  const int vec = dtype == kF32 ? 4 : 8;
  if (g.C <= 0 || g.C % vec != 0) throw std::invalid_argument("depthwise: C must fill 16-byte vectors");
  if (g.KH < 1 || g.KW < 1 || g.KH * g.KW > kDwMaxTaps || g.KH > 7 || g.KW > 7)
    throw std::invalid_argument("depthwise: filter larger than 7x7");
  if (g.sh < 1 || g.sw < 1 || g.ph < 0 || g.pw < 0) throw std::invalid_argument("depthwise: bad stride / padding");
  if (g.OH != (g.IH + 2 * g.ph - g.KH) / g.sh + 1 || g.OW != (g.IW + 2 * g.pw - g.KW) / g.sw + 1 || g.OH < 1 || g.OW < 1)
    throw std::invalid_argument("depthwise: output extent does not match the geometry");
}

}  // namespace

int dw_conv_rows(long npix, int C, int dtype) {
  const int chunks = C / (dtype == kF32 ? 4 : 8);
  const int CSL = slab_lanes(chunks);
  const int gy = ceil_div(chunks, CSL);
  long bx = ceil_div(npix, (long)(256 / CSL));
  const long cap = std::max(1, (g_stream_grid / 2) / gy);
  if (bx > cap) bx = cap;
  return (int)(bx < 1 ? 1 : bx);
}

void dw_conv_launch(const void* in, const void* w, int w_dtype, const float* bias, void* out, float* stats, int backward,
                    int NB, int IH, int IW, int C, int OH, int OW, int KH, int KW, int sh, int sw, int ph, int pw, int relu,
                    int dtype, hipStream_t s) {
  DwGeom g{NB, IH, IW, C, OH, OW, KH, KW, sh, sw, ph, pw};
  check_geom(g, dtype);
  if (w_dtype != kF32 && w_dtype != dtype) throw std::invalid_argument("depthwise: weight must be fp32 or the compute dtype");
  if (backward && (stats != nullptr || bias != nullptr || relu)) throw std::invalid_argument("depthwise: epilogue is fprop only");
  const int chunks = C / (dtype == kF32 ? 4 : 8);
  const int CSL = slab_lanes(chunks);
  const long npix = (long)NB * (backward ? IH * IW : OH * OW);
  if (npix <= 0) return;
  dim3 grid(dw_conv_rows(npix, C, dtype), ceil_div(chunks, CSL));
  if (npix >= (1L << 31)) throw std::invalid_argument("depthwise: more than 2^31 pixels");
  const FastDiv div_w = make_fastdiv((uint32_t)(backward ? IW : OW)), div_h = make_fastdiv((uint32_t)(backward ? IH : OH));
#define DPA_DW(BW, KWT, KHT)                                                                                          \
  DPA_DISPATCH_DTYPE(dtype, {                                                                                         \
    if (w_dtype == kF32)                                                                                              \
      hipLaunchKernelGGL((dw_conv_kernel<T, float, BW, KWT, KHT>), grid, dim3(256), 0, s, (const T*)in, (const float*)w,    \
                         bias, (T*)out, stats, g, CSL, relu, div_w, div_h);                                           \
    else                                                                                                              \
      hipLaunchKernelGGL((dw_conv_kernel<T, T, BW, KWT, KHT>), grid, dim3(256), 0, s, (const T*)in, (const T*)w, bias,      \
                         (T*)out, stats, g, CSL, relu, div_w, div_h);                                                 \
  })
#define DPA_DW_KW(BW)                                     \
  if (KH == 3 && KW == 3) { DPA_DW(BW, 3, 3); }           \
  else if (KH == 5 && KW == 5) { DPA_DW(BW, 5, 5); }      \
  else if (KW <= 1) { DPA_DW(BW, 1, 0); }                 \
  else if (KW <= 3) { DPA_DW(BW, 3, 0); }                 \
  else if (KW <= 5) { DPA_DW(BW, 5, 0); }                 \
  else { DPA_DW(BW, 7, 0); }
  if (backward) { DPA_DW_KW(true) } else { DPA_DW_KW(false) }
#undef DPA_DW_KW
#undef DPA_DW
  DPA_CHECK_HIP(hipGetLastError());
}

int dw_wgrad_rows(long npix, int C, int KH, int dtype) {
  const int chunks = C / (dtype == kF32 ? 4 : 8);
  const int CSL = slab_lanes(chunks);
  const int PL = 256 / (CSL * KH);
  const int gy = ceil_div(chunks, CSL);
  long bx = ceil_div(npix, (long)PL * 64);             // >= 64 pixels per lane before another partial row pays
  const long cap = std::max(1, 1024 / gy);
  if (bx > cap) bx = cap;
  return (int)(bx < 1 ? 1 : bx);
}

void dw_wgrad_launch(const void* dy, const void* x, float* partial, int NB, int IH, int IW, int C, int OH, int OW, int KH,
                     int KW, int sh, int sw, int ph, int pw, int dtype, hipStream_t s) {
  DwGeom g{NB, IH, IW, C, OH, OW, KH, KW, sh, sw, ph, pw};
  check_geom(g, dtype);
  const int chunks = C / (dtype == kF32 ? 4 : 8);
  const int CSL = slab_lanes(chunks);
  const int PL = 256 / (CSL * KH);
  const long npix = (long)NB * OH * OW;
  if (npix <= 0 || PL < 1) throw std::invalid_argument("depthwise wgrad: empty problem");
  if (npix >= (1L << 31)) throw std::invalid_argument("depthwise: more than 2^31 pixels");
  const FastDiv div_w = make_fastdiv((uint32_t)OW), div_h = make_fastdiv((uint32_t)OH);
  dim3 grid(dw_wgrad_rows(npix, C, KH, dtype), ceil_div(chunks, CSL));
#define DPA_DWW(KWT)                                                                                              \
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dw_wgrad_kernel<T, KWT>), grid, dim3(256), 0, s, (const T*)dy, (const T*)x, \
                                               partial, g, CSL, PL, div_w, div_h))
  if (KW <= 1) { DPA_DWW(1); }
  else if (KW <= 3) { DPA_DWW(3); }
  else if (KW <= 5) { DPA_DWW(5); }
  else { DPA_DWW(7); }
#undef DPA_DWW
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
