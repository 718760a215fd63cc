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

template <typename T, typename TW, bool BWD, int KWT>
__global__ __launch_bounds__(256) void dw_conv_kernel(const T* __restrict__ in, const TW* __restrict__ w,
                                                      const float* __restrict__ bias, T* __restrict__ out,
                                                      float* __restrict__ stats, DwGeom g, int CSL, int relu) {
  constexpr int VEC = 16 / (int)sizeof(T);
  __shared__ float wl[kDwMaxTaps][kDwSlab * VEC];
  __shared__ float red[2][256][VEC + 1];
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
  const long npix = (long)g.NB * PH * PW;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.f;
  const bool active = ch < chunks;
  for (long pix = (long)blockIdx.x * PXL + pl; pix < npix; pix += (long)gridDim.x * PXL) {
    if (!active) continue;
    const int px = (int)(pix % PW);
    const long q = pix / PW;
    const int py = (int)(q % PH);
    const long n = q / PH;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = (!BWD && bias != nullptr) ? bias[ch * VEC + j] : 0.f;
    const T* base = in + n * SH * SW * (long)g.C + ch * VEC;
    for (int r = 0; r < g.KH; ++r) {
      int sy;
      bool oky;
      if (BWD) {
        const int ty = py + g.ph - r;
        sy = ty / g.sh;
        oky = ty >= 0 && sy * g.sh == ty && sy < SH;
      } else {
        sy = py * g.sh - g.ph + r;
        oky = sy >= 0 && sy < SH;
      }
      if (!oky) continue;
      const T* row = base + (long)sy * SW * g.C;
      const float* wr = &wl[r * g.KW][cl * VEC];
      // the whole filter row in flight at once: loads from clamped addresses, masks applied at the multiply
      // (a tap loop with data-dependent skips would serialise one memory latency per tap)
      Vec16<T> v[KWT];
      bool okx[KWT];
#pragma unroll
      for (int s = 0; s < KWT; ++s) {
        int sx;
        if (BWD) {
          const int tx = px + g.pw - s;
          sx = tx / g.sw;
          okx[s] = s < g.KW && tx >= 0 && sx * g.sw == tx && sx < SW;
        } else {
          sx = px * g.sw - g.pw + s;
          okx[s] = s < g.KW && sx >= 0 && sx < SW;
        }
        v[s].load(row + (long)(okx[s] ? sx : 0) * g.C);
      }
#pragma unroll
      for (int s = 0; s < KWT; ++s) {
        if (okx[s]) {
          const float* wt = wr + s * (kDwSlab * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[j] += v[s].get(j) * wt[j];
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
        s1[j] += rv;
        s2[j] += rv * rv;
      }
    }
    o.store(out + pix * g.C + ch * VEC);
  }
  if (!BWD && stats != nullptr) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      red[0][threadIdx.x][j] = s1[j];
      red[1][threadIdx.x][j] = s2[j];
    }
    __syncthreads();
    if (pl == 0 && active) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float a = 0.f, b = 0.f;
        for (int p = 0; p < PXL; ++p) {
          a += red[0][p * CSL + cl][j];
          b += red[1][p * CSL + cl][j];
        }
        stats[((size_t)blockIdx.x * 2 + 0) * g.C + ch * VEC + j] = a;
        stats[((size_t)blockIdx.x * 2 + 1) * g.C + ch * VEC + j] = b;
      }
    }
  }
}

// lanes: cl (channel chunk of the slab) fastest, then filter row r, then pixel lane pl
template <typename T, int KW>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                       float* __restrict__ partial, DwGeom g, int CSL, int PL) {
  constexpr int VEC = 16 / (int)sizeof(T);
  __shared__ float red[256][VEC + 1];
  const int chunks = g.C / VEC;
  const int taps = g.KH * g.KW;
  const int lanes = CSL * g.KH;
  const int cl = threadIdx.x % CSL, r = (threadIdx.x / CSL) % g.KH, pl = threadIdx.x / lanes;
  const int ch = blockIdx.y * CSL + cl;
  const bool active = pl < PL && ch < chunks;
  const long npix = (long)g.NB * g.OH * g.OW;
  float acc[KW][VEC];
#pragma unroll
  for (int s = 0; s < KW; ++s)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[s][j] = 0.f;
  if (active) {
    for (long pix = (long)blockIdx.x * PL + pl; pix < npix; pix += (long)gridDim.x * PL) {
      const int ox = (int)(pix % g.OW);
      const long q = pix / g.OW;
      const int oy = (int)(q % g.OH);
      const long n = q / g.OH;
      const int iy = oy * g.sh - g.ph + r;
      if (iy < 0 || iy >= g.IH) continue;
      Vec16<T> gv;
      gv.load(dy + pix * g.C + ch * VEC);
      const T* row = x + ((n * g.IH + iy) * (long)g.IW) * g.C + ch * VEC;
      const int ix0 = ox * g.sw - g.pw;
      Vec16<T> xv[KW];
      bool ok[KW];
#pragma unroll
      for (int s = 0; s < KW; ++s) {           // all loads of the filter row in flight, clamped addresses
        const int ix = ix0 + s;
        ok[s] = s < g.KW && ix >= 0 && ix < g.IW;
        xv[s].load(row + (long)(ok[s] ? ix : 0) * g.C);
      }
#pragma unroll
      for (int s = 0; s < KW; ++s) {
        if (ok[s]) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[s][j] += gv.get(j) * xv[s].get(j);
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
  const long cap = std::max(1, 2048 / gy);
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
#define DPA_DW(BW, KWT)                                                                                               \
  DPA_DISPATCH_DTYPE(dtype, {                                                                                         \
    if (w_dtype == kF32)                                                                                              \
      hipLaunchKernelGGL((dw_conv_kernel<T, float, BW, KWT>), grid, dim3(256), 0, s, (const T*)in, (const float*)w, bias,   \
                         (T*)out, stats, g, CSL, relu);                                                               \
    else                                                                                                              \
      hipLaunchKernelGGL((dw_conv_kernel<T, T, BW, KWT>), grid, dim3(256), 0, s, (const T*)in, (const T*)w, bias, (T*)out,  \
                         stats, g, CSL, relu);                                                                        \
  })
#define DPA_DW_KW(BW)                       \
  if (KW <= 1) { DPA_DW(BW, 1); }           \
  else if (KW <= 3) { DPA_DW(BW, 3); }      \
  else if (KW <= 5) { DPA_DW(BW, 5); }      \
  else { DPA_DW(BW, 7); }
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
  dim3 grid(dw_wgrad_rows(npix, C, KH, dtype), ceil_div(chunks, CSL));
#define DPA_DWW(KWT)                                                                                              \
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dw_wgrad_kernel<T, KWT>), grid, dim3(256), 0, s, (const T*)dy, (const T*)x, \
                                               partial, g, CSL, PL))
  if (KW <= 1) { DPA_DWW(1); }
  else if (KW <= 3) { DPA_DWW(3); }
  else if (KW <= 5) { DPA_DWW(5); }
  else { DPA_DWW(7); }
#undef DPA_DWW
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
