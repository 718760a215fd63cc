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
constexpr int kDwSlab = 16;         // channel chunks per workgroup (at most)
constexpr int kDwMaxPairs = 28;     // tap pairs: 25 for a flattened 7x7 window, 7 rows x 4 pairs in row mode

// A chunk of zeros in global memory: a tap that falls into the padding (or off the stride lattice) loads THIS instead
// of being skipped, so the inner loops have no data-dependent branches and no masks on the data.
__device__ __attribute__((aligned(16))) const uint32_t g_dw_zero[4] = {0u, 0u, 0u, 0u};

// 16-bit types: v_dot2c_f32_{bf16,f16} multiplies TWO taps of one channel per instruction, D += a.lo * b.lo + a.hi * b.hi
// in fp32, straight from the packed operands -- no unpack to fp32, half the FMAs.  The loaded chunks hold channel pairs,
// so two chunks (taps t, t+1) are re-paired per channel with v_perm_b32; the filter sits in LDS already as
// {w_t[c], w_t+1[c]} words.  8 perms + 8 dot2 per tap PAIR and chunk against 2 x (8 unpacks + 8 FMAs): the 5x5 and 7x7
// kernels were VALU bound (49 taps x 16 instructions per 16 bytes written).
template <typename T>
__device__ __forceinline__ float dw_dot2(uint32_t a, uint32_t b, float c) {
  if constexpr (TypeInfo<T>::kCode == kBF16) {
    bf16x2_t p, q;
    __builtin_memcpy(&p, &a, 4);
    __builtin_memcpy(&q, &b, 4);
    return __builtin_amdgcn_fdot2_f32_bf16(p, q, c, false);
  } else {
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    f16x2_t p, q;
    __builtin_memcpy(&p, &a, 4);
    __builtin_memcpy(&q, &b, 4);
    return __builtin_amdgcn_fdot2(p, q, c, false);
  }
}
// acc[j] += xa[j] * wa[j] + xb[j] * wb[j] for the 8 channels of a chunk; w[j] = {wa[j] (low half), wb[j] (high half)}
template <typename T>
__device__ __forceinline__ void dw_pair(const uint4& xa, const uint4& xb, const uint4& w0, const uint4& w1,
                                        float (&acc)[8]) {
  const uint32_t a[4] = {xa.x, xa.y, xa.z, xa.w}, b[4] = {xb.x, xb.y, xb.z, xb.w};
  const uint32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t lo = __builtin_amdgcn_perm(b[q], a[q], 0x05040100u);     // {a.low16, b.low16}: channel 2q
    const uint32_t hi = __builtin_amdgcn_perm(b[q], a[q], 0x07060302u);     // {a.high16, b.high16}: channel 2q + 1
    acc[2 * q] = dw_dot2<T>(lo, w[2 * q], acc[2 * q]);
    acc[2 * q + 1] = dw_dot2<T>(hi, w[2 * q + 1], acc[2 * q + 1]);
  }
}
template <typename T> __device__ __forceinline__ uint32_t dw_bits16(float f) {
  const T t = from_float<T>(f);
  uint16_t u;
  __builtin_memcpy(&u, &t, 2);
  return (uint32_t)u;
}

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
  constexpr bool PK = sizeof(T) == 2;                        // tap pairs through dot2
  constexpr int PR = (KWT + 1) / 2;                          // row mode: pairs per filter row
  __shared__ float wl[PK ? 1 : kDwMaxTaps][PK ? 1 : kDwSlab * VEC];
  __shared__ __attribute__((aligned(16))) uint32_t wl2[PK ? kDwMaxPairs : 1][PK ? kDwSlab * VEC : 1];
  __shared__ float red[2][256][VEC + 1];      // re-based sum and sum of squares per lane (+ pixel count in slot VEC)
  __shared__ float kbuf[kDwSlab][VEC];        // the shift every lane of a channel is re-based on
  const int chunks = g.C / VEC;
  const int taps = g.KH * g.KW;
  const int cl = threadIdx.x % CSL, pl = threadIdx.x / CSL, PXL = 256 / CSL;
  const int ch = blockIdx.y * CSL + cl;
  const int c_slab = blockIdx.y * CSL * VEC;
  if constexpr (PK) {
    // pair (a, b) of taps: flattened window order when the window is a template constant, (row, column pair) in row mode
    const int npairs = KHT > 0 ? (KHT * KWT + 1) / 2 : g.KH * PR;
    for (int i = threadIdx.x; i < npairs * CSL * VEC; i += 256) {
      const int tp = i / (CSL * VEC), cc = i - tp * (CSL * VEC);
      const int c = c_slab + cc;
      int ta, tb;
      if (KHT > 0) {
        ta = 2 * tp;
        tb = 2 * tp + 1 < taps ? 2 * tp + 1 : -1;
      } else {
        const int r = tp / PR, sp = tp - r * PR;
        ta = 2 * sp < g.KW ? r * g.KW + 2 * sp : -1;
        tb = 2 * sp + 1 < g.KW ? r * g.KW + 2 * sp + 1 : -1;
      }
      const uint32_t lo = (c < g.C && ta >= 0) ? dw_bits16<T>(to_float(w[(size_t)c * taps + ta])) : 0u;
      const uint32_t hi = (c < g.C && tb >= 0) ? dw_bits16<T>(to_float(w[(size_t)c * taps + tb])) : 0u;
      wl2[tp][cc] = lo | (hi << 16);
    }
  } else {
    for (int i = threadIdx.x; i < taps * CSL * VEC; i += 256) {
      const int t = i / (CSL * VEC), cc = i - t * (CSL * VEC);
      const int c = c_slab + cc;
      wl[t][cc] = c < g.C ? to_float(from_float<T>(to_float(w[(size_t)c * taps + t]))) : 0.f;
    }
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
  const bool active = ch < chunks && pl < PXL;
  const int st_y = g.sh, st_x = g.sw;
  const T* zero = reinterpret_cast<const T*>(g_dw_zero);
  // A workgroup walks ONE contiguous span of pixels (several image rows), PXL at a time: the rows above and below a
  // pixel are then read by the same CU a few iterations apart and hit in L1 / L2.  (Grid-stride assignment sent the
  // three rows of a 3x3 window to three workgroups on three XCDs: every input line crossed the fabric 3+ times and the
  // kernels ran at 1.7 TB/s of useful bytes.)
  const uint32_t span = ((npix + gridDim.x - 1) / gridDim.x + PXL - 1) / PXL * PXL;
  const uint32_t p_begin = blockIdx.x * span, p_end = min(npix, p_begin + span);
  for (uint32_t pix = p_begin + pl; pix < p_end; pix += PXL) {
    if (!active) continue;
    const uint32_t q = fdiv(pix, div_w);
    const int px = (int)(pix - q * PW);
    const uint32_t n = fdiv(q, div_h);
    const int py = (int)(q - n * PH);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = (!BWD && bias != nullptr) ? bias[ch * VEC + j] : 0.f;
    const T* base = in + (long)n * SH * SW * g.C + ch * VEC;
    if constexpr (PK && KHT > 0) {
      // every tap of the window in flight at once; a tap outside the image reads the zero chunk
      constexpr int NT = KHT * KWT;
      Vec16<T> v[NT + 1];
      int sx[KWT];
      bool okx[KWT];
#pragma unroll
      for (int s = 0; s < KWT; ++s) okx[s] = dw_src<BWD>(px, s, st_x, g.pw, SW, sx[s]);
#pragma unroll
      for (int r = 0; r < KHT; ++r) {
        int sy;
        const bool oky = dw_src<BWD>(py, r, st_y, g.ph, SH, sy);
        const T* row = base + (long)sy * SW * g.C;
#pragma unroll
        for (int s = 0; s < KWT; ++s) v[r * KWT + s].load(oky && okx[s] ? row + (long)sx[s] * g.C : zero);
      }
      v[NT].raw = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int tp = 0; tp < (NT + 1) / 2; ++tp) {
        const uint4* wp = reinterpret_cast<const uint4*>(&wl2[tp][cl * VEC]);
        dw_pair<T>(v[2 * tp].raw, v[2 * tp + 1].raw, wp[0], wp[1], acc);
      }
    } else if constexpr (PK) {
      for (int r = 0; r < g.KH; ++r) {
        int sy;
        if (!dw_src<BWD>(py, r, st_y, g.ph, SH, sy)) continue;
        const T* row = base + (long)sy * SW * g.C;
        // the whole filter row in flight at once
        Vec16<T> v[2 * PR];
#pragma unroll
        for (int s = 0; s < 2 * PR; ++s) {
          int sx = 0;
          const bool ok = s < g.KW && dw_src<BWD>(px, s, st_x, g.pw, SW, sx);
          v[s].load(ok ? row + (long)sx * g.C : zero);
        }
#pragma unroll
        for (int sp = 0; sp < PR; ++sp) {
          const uint4* wp = reinterpret_cast<const uint4*>(&wl2[r * PR + sp][cl * VEC]);
          dw_pair<T>(v[2 * sp].raw, v[2 * sp + 1].raw, wp[0], wp[1], acc);
        }
      }
    } else if constexpr (KHT > 0) {
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
    // per thread: sums of (v - k_t), (v - k_t)^2 and its shift k_t.  Every lane re-bases its sums on the shift of pixel
    // lane 0 of its chunk (a sample as well, so every term stays at the scale of the spread), then one thread per
    // CHANNEL of the slab adds the pixel lanes in fp32 (the first version gave a chunk's 8 channels to one thread).
    if (pl == 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) kbuf[cl][j] = kk[j];       // pixel lane 0 always has a pixel when the block has any
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float d = kk[j] - kbuf[cl][j];
      red[0][threadIdx.x][j] = s1[j] + npx * d;
      red[1][threadIdx.x][j] = s2[j] + 2.f * d * s1[j] + npx * d * d;
    }
    red[0][threadIdx.x][VEC] = npx;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < CSL * VEC) {
      const int c2 = t / VEC, j = t - c2 * VEC;
      const int chn = blockIdx.y * CSL + c2;
      if (chn < chunks) {
        const float K = kbuf[c2][j];
        float N = 0.f, S = 0.f, Q = 0.f;
        for (int p = 0; p < PXL; ++p) {
          const int u = p * CSL + c2;
          N += red[0][u][VEC];
          S += red[0][u][j];
          Q += red[1][u][j];
        }
        const float m2 = N > 0.f ? Q - S * S / N : 0.f;
        stats[((size_t)blockIdx.x * 2 + 0) * g.C + chn * VEC + j] = S + N * K;
        stats[((size_t)blockIdx.x * 2 + 1) * g.C + chn * VEC + j] = fmaxf(m2, 0.f);
        if (blockIdx.y == 0 && t == 0) stats[(size_t)gridDim.x * 2 * g.C + blockIdx.x] = N;
      }
    }
  }
}

// Stride-1 depthwise convolution (forward and backward-data), 16-bit types, KH x KW a template constant: a lane owns one
// (column x, channel chunk) and walks DOWN a strip of R output rows with the window's rows in registers -- every new
// output row loads KW chunks instead of KH * KW, and the coordinate / address work is done once per strip.  Counters on
// the window-per-pixel kernel (56x56x144, 3x3): 297 VALU + 87 SALU instructions per 64 outputs (the arithmetic is ~90),
// 3.2x the input's bytes requested from L2 at a 30 % hit rate.  Both directions run the same loop: the filter is staged
// in LDS in WINDOW order (flipped for backward-data), out(y, x) = sum_{k,c} src(y - pad_y + k, x - pad_x + c) * wwin[k][c]
// with pad = p (forward) or K - 1 - p (backward), taps outside the source reading the zero chunk.
// PF: the next output row's chunks are requested before the current row's arithmetic (one more window row in registers).
template <typename T, typename TW, bool BWD, int KH, int KW, bool PF>
__global__ __launch_bounds__(256, KH == 3 ? (BWD ? 4 : 3) : 2) void dw_conv_s1_kernel(const T* __restrict__ in, const TW* __restrict__ w,
                                                         const float* __restrict__ bias, T* __restrict__ out,
                                                         float* __restrict__ stats, DwGeom g, int CSL, int relu, int R,
                                                         FastDiv div_w, FastDiv div_s) {
  static_assert(sizeof(T) == 2, "16-bit types");
  constexpr int VEC = 8;
  constexpr int PR = (KW + 1) / 2;            // tap pairs per window row (the last one pairs with zeros when KW is odd)
  constexpr int NS = PF ? KH + 1 : KH;        // window row slots
  __shared__ __attribute__((aligned(16))) uint32_t wl2[KH * PR][kDwSlab * VEC];
  __shared__ float red[2][256][VEC + 1];
  __shared__ float kbuf[kDwSlab][VEC];
  const int chunks = g.C / VEC;
  constexpr int taps = KH * KW;
  const int cl = threadIdx.x % CSL, pl = threadIdx.x / CSL, PXL = 256 / CSL;
  const int ch = blockIdx.y * CSL + cl;
  const int c_slab = blockIdx.y * CSL * VEC;
  for (int i = threadIdx.x; i < KH * PR * CSL * VEC; i += 256) {
    const int tp = i / (CSL * VEC), cc = i - tp * (CSL * VEC);
    const int c = c_slab + cc;
    const int k = tp / PR, cp = tp - k * PR;
    const int ca = 2 * cp, cb = 2 * cp + 1;
    // window (k, c) <-> filter tap (k, c) forward, (KH-1-k, KW-1-c) backward
    const int ta = BWD ? (KH - 1 - k) * KW + (KW - 1 - ca) : k * KW + ca;
    const int tb = BWD ? (KH - 1 - k) * KW + (KW - 1 - cb) : k * KW + cb;
    const uint32_t lo = c < g.C ? dw_bits16<T>(to_float(w[(size_t)c * taps + ta])) : 0u;
    const uint32_t hi = (c < g.C && cb < KW) ? dw_bits16<T>(to_float(w[(size_t)c * taps + tb])) : 0u;
    wl2[tp][cc] = lo | (hi << 16);
  }
  __syncthreads();
  const int PH = BWD ? g.IH : g.OH, PW = BWD ? g.IW : g.OW;     // written tensor
  const int SH = BWD ? g.OH : g.IH, SW = BWD ? g.OW : g.IW;     // read tensor
  const int pad_y = BWD ? KH - 1 - g.ph : g.ph, pad_x = BWD ? KW - 1 - g.pw : g.pw;
  const uint32_t nstrips = div_s.d;
  const uint32_t nitems = (uint32_t)g.NB * nstrips * PW;       // < 2^31 (launcher)
  float s1[VEC], s2[VEC], kk[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = kk[j] = 0.f;
  float npx = 0.f;
  const bool active = ch < chunks && pl < PXL;
  const T* zero = reinterpret_cast<const T*>(g_dw_zero);
  const long rowC = (long)SW * g.C;
  // contiguous span of items per workgroup: x fastest, then the strips of one image (halo rows shared in L2)
  const uint32_t span = ((nitems + gridDim.x - 1) / gridDim.x + PXL - 1) / PXL * PXL;
  const uint32_t i_begin = blockIdx.x * span, i_end = min(nitems, i_begin + span);
  for (uint32_t item = i_begin + pl; item < i_end; item += PXL) {
    if (!active) continue;
    const uint32_t q = fdiv(item, div_w);
    const int x = (int)(item - q * PW);
    const uint32_t n = fdiv(q, div_s);
    const int y0 = (int)(q - n * nstrips) * R;
    const int y1 = min(PH, y0 + R);
    const T* base = in + (long)n * SH * rowC + ch * VEC;
    const T* colp[KW];                         // source column pointers (row 0) or null for a column in the padding
#pragma unroll
    for (int c = 0; c < KW; ++c) {
      const int sx = x - pad_x + c;
      colp[c] = (sx >= 0 && sx < SW) ? base + (long)sx * g.C : nullptr;
    }
    Vec16<T> win[NS][2 * PR];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
#pragma unroll
      for (int c = KW; c < 2 * PR; ++c) win[sl][c].raw = make_uint4(0u, 0u, 0u, 0u);      // the odd column's partner
    auto load_row = [&](int sy, Vec16<T>(&row)[2 * PR]) {
      const bool oky = sy >= 0 && sy < SH;
      const long ro = (long)sy * rowC;
#pragma unroll
      for (int c = 0; c < KW; ++c) row[c].load((oky && colp[c] != nullptr) ? colp[c] + ro : zero);
    };
    // rows of the first output: slots 0 .. KH-1 (PF) / 0 .. KH-2 (the last one is loaded in the loop)
#pragma unroll
    for (int k = 0; k < (PF ? KH : KH - 1); ++k) load_row(y0 - pad_y + k, win[k]);
    T* outp = out + ((long)(n * PH + y0) * PW + x) * g.C + ch * VEC;
    for (int yb = y0; yb < y1; yb += NS) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int y = yb + i;
        if (y < y1) {
          // source rows y - pad_y + k sit in slots (i + k) % NS
          if constexpr (PF) {
            if (y + 1 < y1) load_row(y + 1 - pad_y + KH - 1, win[(i + KH) % NS]);
          } else {
            load_row(y - pad_y + KH - 1, win[(i + KH - 1) % NS]);
          }
          float acc[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[j] = (!BWD && bias != nullptr) ? bias[ch * VEC + j] : 0.f;
          // the filter words are re-read from LDS for every output: kept in registers across the strip (what the
          // compiler does with a loop-invariant address) they cost 48 (3x3) / 120 (5x5) VGPRs and the occupancy
          int wofs = cl * VEC;
          asm volatile("" : "+v"(wofs));
#pragma unroll
          for (int k = 0; k < KH; ++k)
#pragma unroll
            for (int cp = 0; cp < PR; ++cp) {
              const uint4* wp = reinterpret_cast<const uint4*>(&wl2[k * PR + cp][wofs]);
              dw_pair<T>(win[(i + k) % NS][2 * cp].raw, win[(i + k) % NS][2 * cp + 1].raw, wp[0], wp[1], acc);
            }
          Vec16<T> o;
#pragma unroll
          for (int j = 0; j < VEC; ++j) o.set(j, relu ? fmaxf(acc[j], 0.f) : acc[j]);
          o.store(outp);
          outp += (long)PW * g.C;
          if (!BWD && stats != nullptr) {          // on the rounded values, what BatchNorm will read back
            if (npx == 0.f) {
#pragma unroll
              for (int j = 0; j < VEC; ++j) kk[j] = o.get(j);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              const float f = o.get(j) - kk[j];
              s1[j] += f;
              s2[j] += f * f;
            }
          }
          npx += 1.f;
        }
      }
    }
  }
  if (!BWD && stats != nullptr) {
    // as in dw_conv_kernel: re-base on pixel lane 0's shift, one thread per channel adds the pixel lanes
    if (pl == 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) kbuf[cl][j] = kk[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float d = kk[j] - kbuf[cl][j];
      red[0][threadIdx.x][j] = s1[j] + npx * d;
      red[1][threadIdx.x][j] = s2[j] + 2.f * d * s1[j] + npx * d * d;
    }
    red[0][threadIdx.x][VEC] = npx;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < CSL * VEC) {
      const int c2 = t / VEC, j = t - c2 * VEC;
      const int chn = blockIdx.y * CSL + c2;
      if (chn < chunks) {
        const float K = kbuf[c2][j];
        float N = 0.f, S = 0.f, Q = 0.f;
        for (int p = 0; p < PXL; ++p) {
          const int u = p * CSL + c2;
          N += red[0][u][VEC];
          S += red[0][u][j];
          Q += red[1][u][j];
        }
        const float m2 = N > 0.f ? Q - S * S / N : 0.f;
        stats[((size_t)blockIdx.x * 2 + 0) * g.C + chn * VEC + j] = S + N * K;
        stats[((size_t)blockIdx.x * 2 + 1) * g.C + chn * VEC + j] = fmaxf(m2, 0.f);
        if (blockIdx.y == 0 && t == 0) stats[(size_t)gridDim.x * 2 * g.C + blockIdx.x] = N;
      }
    }
  }
}

// Backward-data of a stride-2 depthwise convolution (K = 3 pad 1, K = 5 pad 2, even input extents; 16-bit types).
// dx[2i + a] only meets the taps t = a + pad (mod 2), source i + (a + pad - t) / 2: a lane that owns a 2 x 2 block of dx
// loads the (K + 1) / 2 squared dy chunks around (i, j) ONCE and every filter tap is used exactly once across the four
// parities -- 4 loads and 9 multiplies (3x3) for four outputs.  The gather-form kernel asked each dx pixel for all K * K
// taps and multiplied the off-lattice ones (three quarters) by the zero chunk: 400 us = 1.9 TB/s on MobileNetV2's
// 112x112x96 layer, its most expensive depthwise launch.
// The taps of one parity are paired for dot2 in enumeration order (ty outer, tx inner); s2_pair_taps() gives the staging
// loop the same order.
template <int K>
__device__ __forceinline__ void s2_pair_taps(int tp, int& ta, int& tb) {
  constexpr int pad = K / 2;
  int idx = 0;
  ta = tb = -1;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      int cnt = 0, prev = -1;
      for (int ty = 0; ty < K; ++ty) {
        if (((ty + a + pad) & 1) != 0) continue;
        for (int tx = 0; tx < K; ++tx) {
          if (((tx + b + pad) & 1) != 0) continue;
          if ((cnt & 1) == 0) prev = ty * K + tx;
          else {
            if (idx == tp) { ta = prev; tb = ty * K + tx; }
            ++idx;
          }
          ++cnt;
        }
      }
      if (cnt & 1) {
        if (idx == tp) { ta = prev; tb = -1; }
        ++idx;
      }
    }
}

template <typename T, typename TW, int K>
__global__ __launch_bounds__(256) void dw_bwd_s2_kernel(const T* __restrict__ dy, const TW* __restrict__ w,
                                                        T* __restrict__ dx, DwGeom g, int CSL, FastDiv div_w,
                                                        FastDiv div_h) {
  static_assert(sizeof(T) == 2 && (K == 3 || K == 5), "16-bit types, 3x3 / 5x5");
  constexpr int VEC = 8, pad = K / 2;
  constexpr int WN = (K + 1) / 2, W0 = K == 3 ? 0 : -1;      // dy window: rows / columns i + W0 .. i + W0 + WN - 1
  constexpr int NP = K == 3 ? 5 : 13;                        // tap pairs over the four parities
  constexpr int taps = K * K;
  __shared__ __attribute__((aligned(16))) uint32_t wl2[NP][kDwSlab * VEC];
  const int chunks = g.C / VEC;
  const int cl = threadIdx.x % CSL, pl = threadIdx.x / CSL, PXL = 256 / CSL;
  const int ch = blockIdx.y * CSL + cl;
  const int c_slab = blockIdx.y * CSL * VEC;
  for (int i = threadIdx.x; i < NP * CSL * VEC; i += 256) {
    const int tp = i / (CSL * VEC), cc = i - tp * (CSL * VEC);
    const int c = c_slab + cc;
    int ta, tb;
    s2_pair_taps<K>(tp, ta, tb);
    const uint32_t lo = (c < g.C && ta >= 0) ? dw_bits16<T>(to_float(w[(size_t)c * taps + ta])) : 0u;
    const uint32_t hi = (c < g.C && tb >= 0) ? dw_bits16<T>(to_float(w[(size_t)c * taps + tb])) : 0u;
    wl2[tp][cc] = lo | (hi << 16);
  }
  __syncthreads();
  const int BH = g.IH / 2, BW = g.IW / 2;                    // 2 x 2 blocks of dx
  const uint32_t nitems = (uint32_t)g.NB * BH * BW;
  const bool active = ch < chunks && pl < PXL;
  const T* zero = reinterpret_cast<const T*>(g_dw_zero);
  const uint32_t span = ((nitems + gridDim.x - 1) / gridDim.x + PXL - 1) / PXL * PXL;
  const uint32_t i_begin = blockIdx.x * span, i_end = min(nitems, i_begin + span);
  for (uint32_t item = i_begin + pl; item < i_end; item += PXL) {
    if (!active) continue;
    const uint32_t q = fdiv(item, div_w);
    const int j = (int)(item - q * BW);
    const uint32_t n = fdiv(q, div_h);
    const int i = (int)(q - n * BH);
    const T* base = dy + (long)n * g.OH * g.OW * g.C + ch * VEC;
    Vec16<T> win[WN][WN];
#pragma unroll
    for (int r = 0; r < WN; ++r) {
      const int sy = i + W0 + r;
      const bool oky = sy >= 0 && sy < g.OH;
#pragma unroll
      for (int c = 0; c < WN; ++c) {
        const int sx = j + W0 + c;
        win[r][c].load((oky && sx >= 0 && sx < g.OW) ? base + ((long)sy * g.OW + sx) * g.C : zero);
      }
    }
    int wofs = cl * VEC;
    asm volatile("" : "+v"(wofs));
    const uint4 zz = make_uint4(0u, 0u, 0u, 0u);
    int idx = 0;                                 // compile-time after unrolling: the pair counter of s2_pair_taps
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float acc[VEC];
#pragma unroll
        for (int jj = 0; jj < VEC; ++jj) acc[jj] = 0.f;
        int cnt = 0, pr = 0, pc = 0;
#pragma unroll
        for (int ty = 0; ty < K; ++ty) {
          if (((ty + a + pad) & 1) != 0) continue;
#pragma unroll
          for (int tx = 0; tx < K; ++tx) {
            if (((tx + b + pad) & 1) != 0) continue;
            const int wr = (a + pad - ty) / 2 - W0, wc = (b + pad - tx) / 2 - W0;
            if ((cnt & 1) == 0) { pr = wr; pc = wc; }
            else {
              const uint4* wp = reinterpret_cast<const uint4*>(&wl2[idx][wofs]);
              dw_pair<T>(win[pr][pc].raw, win[wr][wc].raw, wp[0], wp[1], acc);
              ++idx;
            }
            ++cnt;
          }
        }
        if (cnt & 1) {
          const uint4* wp = reinterpret_cast<const uint4*>(&wl2[idx][wofs]);
          dw_pair<T>(win[pr][pc].raw, zz, wp[0], wp[1], acc);
          ++idx;
        }
        Vec16<T> o;
#pragma unroll
        for (int jj = 0; jj < VEC; ++jj) o.set(jj, acc[jj]);
        o.store(dx + (((long)n * g.IH + 2 * i + a) * g.IW + 2 * j + b) * g.C + ch * VEC);
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
    // one contiguous span of pixels per workgroup (see dw_conv_kernel), 2 * PL of them per iteration
    const uint32_t step = PL;
    constexpr bool PK = sizeof(T) == 2;        // 16-bit types: the two pixels of an iteration are one dot2 (see dw_pair)
    const T* zero = reinterpret_cast<const T*>(g_dw_zero);
    const uint32_t span = ((npix + gridDim.x - 1) / gridDim.x + 2 * PL - 1) / (2 * PL) * (2 * PL);
    const uint32_t p_begin = blockIdx.x * span, p_end = min(npix, p_begin + span);
    for (uint32_t pix0 = p_begin + pl; pix0 < p_end; pix0 += 2 * step) {
      Vec16<T> gv[2], xv[2][KW];
      bool ok[2][KW];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t pix = pix0 + u * step;
        const bool live = pix < p_end;
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
        for (int s = 0; s < KW; ++s) {           // a tap in the padding reads the zero chunk (16-bit) / is masked (fp32)
          const int ix = ix0 + s;
          ok[u][s] = oky && s < g.KW && ix >= 0 && ix < g.IW;
          if constexpr (PK) xv[u][s].load(ok[u][s] ? row + (long)ix * g.C : zero);
          else xv[u][s].load(row + (long)(ok[u][s] ? ix : 0) * g.C);
        }
      }
      if constexpr (PK) {
        const uint32_t g0[4] = {gv[0].raw.x, gv[0].raw.y, gv[0].raw.z, gv[0].raw.w};
        const uint32_t g1[4] = {gv[1].raw.x, gv[1].raw.y, gv[1].raw.z, gv[1].raw.w};
        uint32_t glo[4], ghi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          glo[q] = __builtin_amdgcn_perm(g1[q], g0[q], 0x05040100u);
          ghi[q] = __builtin_amdgcn_perm(g1[q], g0[q], 0x07060302u);
        }
#pragma unroll
        for (int s = 0; s < KW; ++s) {
          const uint32_t x0[4] = {xv[0][s].raw.x, xv[0][s].raw.y, xv[0][s].raw.z, xv[0][s].raw.w};
          const uint32_t x1[4] = {xv[1][s].raw.x, xv[1][s].raw.y, xv[1][s].raw.z, xv[1][s].raw.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t xlo = __builtin_amdgcn_perm(x1[q], x0[q], 0x05040100u);
            const uint32_t xhi = __builtin_amdgcn_perm(x1[q], x0[q], 0x07060302u);
            acc[s][2 * q] = dw_dot2<T>(glo[q], xlo, acc[s][2 * q]);
            acc[s][2 * q + 1] = dw_dot2<T>(ghi[q], xhi, acc[s][2 * q + 1]);
          }
        }
      } else {
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

// Channel chunks per workgroup: the whole row when it fits, otherwise the divisor of `chunks` in [6, 16] that wastes the
// fewest of the 256 lanes (C = 144: 18 chunks -> 9 x 28 pixel lanes; the power-of-two slabs idled a quarter of the
// lanes there and a third at C = 96), otherwise 8 with a ragged last slab.
int slab_lanes(int chunks) {
  if (chunks <= kDwSlab) return chunks;
  int best = 8, best_used = 0;
  for (int c = kDwSlab; c >= 6; --c) {
    if (chunks % c != 0) continue;
    const int used = (256 / c) * c;
    if (used > best_used) { best = c; best_used = used; }
  }
  return best;
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

int g_dw_s1 = 1;       // 0: never the sliding-window / 2x2-block kernels (A/B hook, dw_set_sliding); >= 2: strip height
void dw_set_sliding(int on) { g_dw_s1 = on; }

int dw_conv_rows(long npix, int C, int dtype) {
  const int chunks = C / (dtype == kF32 ? 4 : 8);
  const int CSL = slab_lanes(chunks);
  const int gy = ceil_div(chunks, CSL);
  long bx = ceil_div(npix, (long)(256 / CSL) * 4);      // >= 4 pixels per lane: the statistics epilogue is per workgroup
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
  if (g_dw_s1 && backward && dtype != kF32 && sh == 2 && sw == 2 && KH == KW && (KH == 3 || KH == 5) && ph == KH / 2 &&
      pw == KW / 2 && IH % 2 == 0 && IW % 2 == 0) {
    const FastDiv dvw = make_fastdiv((uint32_t)(IW / 2)), dvh = make_fastdiv((uint32_t)(IH / 2));
#define DPA_DW_S2(K)                                                                                                  \
  DPA_DISPATCH_DTYPE(dtype, {                                                                                         \
    if constexpr (sizeof(T) == 2) {                                                                                   \
      if (w_dtype == kF32)                                                                                            \
        hipLaunchKernelGGL((dw_bwd_s2_kernel<T, float, K>), grid, dim3(256), 0, s, (const T*)in, (const float*)w,     \
                           (T*)out, g, CSL, dvw, dvh);                                                                \
      else                                                                                                            \
        hipLaunchKernelGGL((dw_bwd_s2_kernel<T, T, K>), grid, dim3(256), 0, s, (const T*)in, (const T*)w, (T*)out, g, \
                           CSL, dvw, dvh);                                                                            \
    }                                                                                                                 \
  })
    if (KH == 3) { DPA_DW_S2(3); } else { DPA_DW_S2(5); }
#undef DPA_DW_S2
    DPA_CHECK_HIP(hipGetLastError());
    return;
  }
  if (g_dw_s1 && dtype != kF32 && sh == 1 && sw == 1 && KH == KW && (KH == 3 || KH == 5) && ph < KH && pw < KW) {
    // sliding-window kernel: strips of R rows; shorter strips when the problem has few columns to hand out
    const int PH = backward ? IH : OH, PW = backward ? IW : OW;
    // rows per strip: 8 unless that leaves fewer columns than ~4 workgroups per CU have lanes for (workgroups beyond
    // the columns idle and write empty statistics rows); g_dw_s1 >= 2 forces the strip height (tests)
    int R = 8;
    while (R > 1 && (long)NB * ceil_div(PH, R) * PW * (long)grid.y < 1024l * (256 / CSL)) R >>= 1;
    if (g_dw_s1 >= 2) R = g_dw_s1;
    const int nstrips = ceil_div(PH, R);
    if ((long)NB * nstrips * PW >= (1L << 31)) throw std::invalid_argument("depthwise: more than 2^31 columns");
    const FastDiv dvw = make_fastdiv((uint32_t)PW), dvs = make_fastdiv((uint32_t)nstrips);
#define DPA_DW_S1(BW, K, PFV)                                                                                          \
  DPA_DISPATCH_DTYPE(dtype, {                                                                                          \
    if constexpr (sizeof(T) == 2) {                                                                                    \
      if (w_dtype == kF32)                                                                                             \
        hipLaunchKernelGGL((dw_conv_s1_kernel<T, float, BW, K, K, PFV>), grid, dim3(256), 0, s, (const T*)in,          \
                           (const float*)w, bias, (T*)out, stats, g, CSL, relu, R, dvw, dvs);                          \
      else                                                                                                             \
        hipLaunchKernelGGL((dw_conv_s1_kernel<T, T, BW, K, K, PFV>), grid, dim3(256), 0, s, (const T*)in, (const T*)w, \
                           bias, (T*)out, stats, g, CSL, relu, R, dvw, dvs);                                           \
    }                                                                                                                  \
  })
    if (backward) { if (KH == 3) { DPA_DW_S1(true, 3, true); } else { DPA_DW_S1(true, 5, false); } }
    else { if (KH == 3) { DPA_DW_S1(false, 3, true); } else { DPA_DW_S1(false, 5, false); } }
#undef DPA_DW_S1
    DPA_CHECK_HIP(hipGetLastError());
    return;
  }
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
