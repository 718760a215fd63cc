// Weight-gradient implicit GEMM on gfx950 MFMA:
//
//     dW[n][t*Cred + c] = sum_m dY[m][n] * X[pix(m) + tap_t][c]
//
// The reduction axis (pixels, up to millions) is the slow memory axis of BOTH operands, so the
// tiles are staged pixel-major in LDS ([pixel][channel], exactly as they sit in HBM) and the MFMA
// fragments are fetched with the gfx950 transposed LDS read (ds_read_b64_tr_b16): no explicit
// transpose anywhere.  fp32 uses the 32x32x2 f32 MFMA whose one-element-per-lane operands read
// the pixel-major tile directly.
//
// The pixel range is split across workgroups; each writes an fp32 partial slab and a second
// kernel sums the slabs in a fixed order (deterministic, no float atomics) straight into the
// parameter's gradient (which may be a view into a DDP bucket).
//
// Replaces cuDNN's backward-filter + cuBLAS dW GEMMs (reference: loss.backward() at
// nd_imagenet.py:302, imagenet_ddp_apex.py:381) -- SURVEY.md N-K3, N-K10.
#include "../common.h"
#include "kernels.h"
#include <stdexcept>
#include <string>
#include <type_traits>

namespace dpa {
namespace {

typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

__device__ __forceinline__ uint2 tr_read(const unsigned char* p) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  uint2 r;
  __builtin_memcpy(&r, &v, 8);
  return r;
}

// LIN: 1x1 / stride 1 / no padding and both operands < 4 GiB (host checked): the im2col row of pixel m is row m
// of X, so both operands are walked with 32-bit byte offsets that advance by a constant per K tile -- no
// divisions, no 64-bit address arithmetic in the loop (PMC counters showed the generic gather issue bound).
// XF: X is the raw input z of a BatchNorm + ReLU; the staged operand is relu(x_scale[c] * z + x_shift[c]).  A thread
// stages one fixed 16-byte chunk column for the whole kernel, so its 8 (scale, shift) pairs live in registers.
template <typename T, int BI, int BJ, int KB, int WI, int WJ, bool LIN, int PF, bool XF = false>
__global__ __launch_bounds__(64 * WI * WJ) void wgrad_kernel(const WgradParams p) {
  static_assert(!XF || sizeof(T) == 2, "apply-on-load: 16-bit problems");
  constexpr int NT = 64 * WI * WJ;                     // WI x WJ waves, each owns a (BI/WI) x (BJ/WJ) sub-tile
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int BKP = KB / (int)sizeof(T);             // pixels per K tile (KB bytes of K per column)
  constexpr int LDP = BI * (int)sizeof(T) + 64;        // row stride = 64 (mod 128): conflict-free tr reads
  constexpr int LDQ = BJ * (int)sizeof(T) + 64;
  constexpr int CPP = BI * (int)sizeof(T) / 16, CPQ = BJ * (int)sizeof(T) / 16;
  constexpr int P_IT = BKP * CPP / NT, Q_IT = BKP * CPQ / NT;
  constexpr int WTI = BI / WI, WTJ = BJ / WJ;
  constexpr int MI = WTI / 32, NJ = WTJ / 32;
  constexpr int STAGE = BKP * (LDP + LDQ);
  static_assert(P_IT >= 1 && Q_IT >= 1 && MI >= 1 && NJ >= 1, "tile too small");
  static_assert(NT % CPP == 0 && NT % CPQ == 0, "chunk mapping");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave / WJ, wj = wave % WJ;

  const uint32_t tiles_i1 = (p.N + BI - 1) / BI;
  const uint32_t tiles_i = tiles_i1 + (p.N2 + BI - 1) / BI, tiles_j = (p.Ktot + BJ - 1) / BJ;
  const uint32_t tiles = tiles_i * tiles_j;
  const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t split = lid / tiles, tile = lid - split * tiles;
  const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
  // output-row tiles past the first operand's belong to the second gradient operand (WgradParams::dY2): uniform per
  // workgroup, so it is a choice of base pointer, row stride and row count made once
  const bool second = (uint32_t)ti >= tiles_i1;
  const int i0 = (second ? ti - (int)tiles_i1 : ti) * BI, j0 = tj * BJ;
  const int Nloc = second ? p.N2 : p.N, ldyl = second ? p.ldy2 : p.ldy;
  const int row_base = second ? p.N : 0;            // first slab row of this operand

  const int m_begin = split * p.pix_per_split;
  const int m_end = min(p.M, m_begin + p.pix_per_split);

  const char* __restrict__ Yg = reinterpret_cast<const char*>(second ? p.dY2 : p.dY) + (size_t)blockIdx.y * p.y_gstride * sizeof(T);
  const char* __restrict__ Xg = reinterpret_cast<const char*>(p.X) + (size_t)blockIdx.y * p.x_gstride * sizeof(T);

  // fixed per-thread columns
  const int cp = tid % CPP;                   // dY chunk column
  const bool pok = i0 + cp * VEC < Nloc;      // N is a multiple of VEC
  const int cq = tid % CPQ;                   // im2col chunk column
  const int kcol = j0 + cq * VEC;
  const bool qok = kcol < p.Ktot;
  int qdy = 0, qdx = 0, qc = 0;
  if (qok) {
    const int t = kcol / p.Cred;
    qc = kcol - t * p.Cred;
    qdy = p.taps[3 * t + 0];
    qdx = p.taps[3 * t + 1];
  }
  float xsc[XF ? VEC : 1], xsh[XF ? VEC : 1];
  if constexpr (XF) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      xsc[j] = qok ? p.x_scale[qc + j] : 0.f;
      xsh[j] = qok ? p.x_shift[qc + j] : 0.f;
    }
  }

  // two register sets: tiles kt+1 and kt+2 are in flight while tile kt is consumed from LDS (one tile ahead
  // left every workgroup waiting a full memory latency per 8 MFMAs)
  uint4 rp[PF][P_IT], rq[PF][Q_IT];
  uint32_t poff[P_IT], qoff[Q_IT], pmask[PF][P_IT], qmask[PF][Q_IT];
  // LIN: 32-bit byte offsets relative to the first pixel of THIS workgroup's split (64-bit bases), so only one split's
  // rows have to span less than 4 GiB (host checked) -- the tensors themselves may be any size
  const char* __restrict__ Yl = Yg + (LIN ? (size_t)m_begin * ldyl * sizeof(T) : (size_t)0);
  const char* __restrict__ Xl = Xg + (LIN ? (size_t)m_begin * p.ldx * sizeof(T) : (size_t)0);
  if (LIN) {
#pragma unroll
    for (int i = 0; i < P_IT; ++i)
      poff[i] = (uint32_t)(((uint32_t)((tid + i * NT) / CPP) * (uint32_t)ldyl + i0 + cp * VEC) * sizeof(T));
#pragma unroll
    for (int i = 0; i < Q_IT; ++i)
      qoff[i] = (uint32_t)(((uint32_t)((tid + i * NT) / CPQ) * (uint32_t)p.ldx + kcol) * sizeof(T));
  }
  const uint32_t pstep = (uint32_t)(BKP * ldyl * (int)sizeof(T)), qstep = (uint32_t)(BKP * p.ldx * (int)sizeof(T));
  // generic gather: incremental pixel coordinates of the rows this thread stages
  const int OWg = (int)p.div_ow.d, OHg = (int)(p.div_ohow.d / p.div_ow.d);
  const int step_n = BKP / (OHg * OWg), step_r = BKP - step_n * OHg * OWg;
  const int step_oy = step_r / OWg, step_ox = step_r - step_oy * OWg;
  int qn_[Q_IT], qoy_[Q_IT], qox_[Q_IT];
  int q_next = -1;
  auto load_tile = [&](int m_base, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    if (LIN) {
      // unconditional loads (rows past the range read offset 0 and are masked at the LDS write)
#pragma unroll
      for (int i = 0; i < P_IT; ++i) {
        const bool ok = pok && m_base + (tid + i * NT) / CPP < m_end;
        pmask[S][i] = ok ? 0xffffffffu : 0u;
        rp[S][i] = *reinterpret_cast<const uint4*>(Yl + (ok ? poff[i] : 0u));
        poff[i] += pstep;
      }
#pragma unroll
      for (int i = 0; i < Q_IT; ++i) {
        const bool ok = qok && m_base + (tid + i * NT) / CPQ < m_end;
        qmask[S][i] = ok ? 0xffffffffu : 0u;
        rq[S][i] = *reinterpret_cast<const uint4*>(Xl + (ok ? qoff[i] : 0u));
        qoff[i] += qstep;
      }
      return;
    }
    // Generic gather.  Like the linear path: unconditional loads (invalid rows read offset 0) with the masks applied
    // at the LDS write -- a select right behind the load would make the wave wait for it before the MFMAs of the
    // tile in flight.  The (image, row, column) of every staged pixel advances incrementally from tile to tile
    // (tiles are requested in order); the two divisions per row per tile made this loop issue bound.
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const int row = (tid + i * NT) / CPP;
      const int m = m_base + row;
      const bool ok = pok && m < m_end;
      pmask[S][i] = ok ? 0xffffffffu : 0u;
      rp[S][i] = *reinterpret_cast<const uint4*>(Yg + (ok ? ((size_t)m * ldyl + i0 + cp * VEC) * sizeof(T) : (size_t)0));
    }
    if (m_base != q_next) {                     // (re)seed the incremental coordinates
#pragma unroll
      for (int i = 0; i < Q_IT; ++i) {
        const uint32_t m = (uint32_t)(m_base + (tid + i * NT) / CPQ);
        const uint32_t n = fdiv(m, p.div_ohow);
        const uint32_t rem = m - n * p.div_ohow.d;
        const uint32_t oy = fdiv(rem, p.div_ow);
        qn_[i] = (int)n; qoy_[i] = (int)oy; qox_[i] = (int)(rem - oy * p.div_ow.d);
      }
    }
    q_next = m_base + BKP;
#pragma unroll
    for (int i = 0; i < Q_IT; ++i) {
      const int m = m_base + (tid + i * NT) / CPQ;
      const int iy = qoy_[i] * p.SY + qdy, ix = qox_[i] * p.SX + qdx;
      const bool ok = qok && m < m_end && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const size_t off = ((size_t)((qn_[i] * p.IH + iy) * p.IW + ix) * p.ldx + qc) * sizeof(T);
      qmask[S][i] = ok ? 0xffffffffu : 0u;
      rq[S][i] = *reinterpret_cast<const uint4*>(Xg + (ok ? off : (size_t)0));
      // advance by BKP pixels: BKP = step_oy * OW + step_ox, at most one image wrap (host checks OH*OW >= BKP,
      // otherwise step_n carries whole images and the loop below still terminates after <= 2 trips)
      int ox = qox_[i] + step_ox, oy = qoy_[i] + step_oy, n = qn_[i] + step_n;
      if (ox >= OWg) { ox -= OWg; ++oy; }
      while (oy >= OHg) { oy -= OHg; ++n; }
      qox_[i] = ox; qoy_[i] = oy; qn_[i] = n;
    }
  };
  auto store_tile = [&](int buf, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    unsigned char* sp = smem + buf * STAGE;
    unsigned char* sq = sp + BKP * LDP;
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const int row = (tid + i * NT) / CPP;
      uint4 v = rp[S][i];
      v.x &= pmask[S][i]; v.y &= pmask[S][i]; v.z &= pmask[S][i]; v.w &= pmask[S][i];
      *reinterpret_cast<uint4*>(sp + row * LDP + cp * 16) = v;
    }
#pragma unroll
    for (int i = 0; i < Q_IT; ++i) {
      const int row = (tid + i * NT) / CPQ;
      uint4 v = rq[S][i];
      if constexpr (XF) v = bn_relu_chunk<T>(v, xsc, xsh);   // as bn_apply stores it; padding is masked afterwards
      v.x &= qmask[S][i]; v.y &= qmask[S][i]; v.z &= qmask[S][i]; v.w &= qmask[S][i];
      *reinterpret_cast<uint4*>(sq + row * LDQ + cq * 16) = v;
    }
  };

  f32x16_t acc[MI][NJ];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < NJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = (m_end - m_begin + BKP - 1) / BKP;
  const int l31 = lane & 31, lh = lane >> 5;
  const int g = lane >> 4, gi = lane & 15;

  auto compute = [&](int cur) {
    const unsigned char* sp = smem + cur * STAGE;
    const unsigned char* sq = sp + BKP * LDP;
    if constexpr (sizeof(T) == 2) {
      // lane address inside a 4-row x 16-col transposed block; see cdna guide T10
      const int trow = 8 * (g >> 1) + (gi >> 2);
      const int tcol = 16 * (g & 1) + 4 * (gi & 3);
#pragma unroll
      for (int ks = 0; ks < BKP / 16; ++ks) {
        uint4 af[MI], bf[NJ];
#pragma unroll
        for (int a = 0; a < MI; ++a) {
          const unsigned char* base = sp + (ks * 16 + trow) * LDP + (wi * WTI + a * 32 + tcol) * 2;
          const uint2 lo = tr_read(base), hi = tr_read(base + 4 * LDP);
          af[a] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int b = 0; b < NJ; ++b) {
          const unsigned char* base = sq + (ks * 16 + trow) * LDQ + (wj * WTJ + b * 32 + tcol) * 2;
          const uint2 lo = tr_read(base), hi = tr_read(base + 4 * LDQ);
          bf[b] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
          for (int b = 0; b < NJ; ++b) {
            if constexpr (TypeInfo<T>::kCode == kBF16) {
              bf16x8_t av, bv;
              __builtin_memcpy(&av, &af[a], 16);
              __builtin_memcpy(&bv, &bf[b], 16);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[a][b], 0, 0, 0);
            } else {
              f16x8_t av, bv;
              __builtin_memcpy(&av, &af[a], 16);
              __builtin_memcpy(&bv, &bf[b], 16);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[a][b], 0, 0, 0);
            }
          }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BKP / 2; ++ks) {
        float af[MI], bf[NJ];
#pragma unroll
        for (int a = 0; a < MI; ++a)
          af[a] = *reinterpret_cast<const float*>(sp + (2 * ks + lh) * LDP + (wi * WTI + a * 32 + l31) * 4);
#pragma unroll
        for (int b = 0; b < NJ; ++b)
          bf[b] = *reinterpret_cast<const float*>(sq + (2 * ks + lh) * LDQ + (wj * WTJ + b * 32 + l31) * 4);
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
          for (int b = 0; b < NJ; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
      }
    }
  };

  constexpr std::integral_constant<int, 0> kSet0{};
  constexpr std::integral_constant<int, 1> kSet1{};
  // Tiles past the range load nothing (all rows masked / zero filled) and add zeros, so the loop needs no
  // tail cases: one straight-line body of two K tiles keeps a single live copy of the accumulators.
  // PF = K tiles in flight in registers while one is consumed from LDS (1: one register set, 2: two).
  load_tile(m_begin, kSet0);
  if constexpr (PF == 2) load_tile(m_begin + BKP, kSet1);
  store_tile(0, kSet0);
  __syncthreads();

  // LDS buffer b holds tile kt (b = kt & 1)
  for (int kt = 0; kt < nk; kt += 2) {
    if constexpr (PF == 2) {        // register set (kt+1)&1 holds tile kt+1, the other one is free
      load_tile(m_begin + (kt + 2) * BKP, kSet0);
      compute(0);
      store_tile(1, kSet1);
      __syncthreads();
      load_tile(m_begin + (kt + 3) * BKP, kSet1);
      compute(1);
      store_tile(0, kSet0);
      __syncthreads();
    } else {
      load_tile(m_begin + (kt + 1) * BKP, kSet0);
      compute(0);
      store_tile(1, kSet0);
      __syncthreads();
      load_tile(m_begin + (kt + 2) * BKP, kSet0);
      compute(1);
      store_tile(0, kSet0);
      __syncthreads();
    }
  }

  // acc[a][b][r]: i = wi*WTI + a*32 + (r&3) + 8*(r>>2) + 4*lh ; j = wj*WTJ + b*32 + l31
  float* __restrict__ out = p.slab + ((size_t)split * p.slab_rows + (size_t)blockIdx.y * p.N + row_base) * p.Ktot;
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < NJ; ++b) {
      const int j = j0 + wj * WTJ + b * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi * WTI + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (i < Nloc && j < p.Ktot) out[(size_t)i * p.Ktot + j] = acc[a][b][r];
      }
    }
}

// out[i] = beta*out[i] + sum_s slab[s][i].  A workgroup owns 64 float4 columns; its 4 waves each fold every
// 4th slab (4 independent loads in flight per lane) and meet in LDS, so a small gradient with hundreds of
// partial slabs is not one long dependent chain per thread.  Fixed summation order => deterministic.
template <typename TO>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int splits, long n,
                                                          TO* __restrict__ out, float beta) {
  __shared__ float4 red[4][64];
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long i = ((long)blockIdx.x * 64 + col) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i + 4 <= n) {
    int k = sl;
    for (; k + 12 < splits; k += 16) {
      const float4 v0 = *reinterpret_cast<const float4*>(slab + (size_t)k * n + i);
      const float4 v1 = *reinterpret_cast<const float4*>(slab + (size_t)(k + 4) * n + i);
      const float4 v2 = *reinterpret_cast<const float4*>(slab + (size_t)(k + 8) * n + i);
      const float4 v3 = *reinterpret_cast<const float4*>(slab + (size_t)(k + 12) * n + i);
      acc.x += (v0.x + v1.x) + (v2.x + v3.x);
      acc.y += (v0.y + v1.y) + (v2.y + v3.y);
      acc.z += (v0.z + v1.z) + (v2.z + v3.z);
      acc.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; k < splits; k += 4) {
      const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)k * n + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  } else if (i < n) {       // ragged tail (n not a multiple of 4): scalar
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = sl; k < splits; k += 4)
      for (int j = 0; j < 4 && i + j < n; ++j) t[j] += slab[(size_t)k * n + i + j];
    acc = make_float4(t[0], t[1], t[2], t[3]);
  }
  red[sl][col] = acc;
  __syncthreads();
  if (sl == 0 && i < n) {
    const float4 a = red[0][col], b = red[1][col], c = red[2][col], d = red[3][col];
    const float r[4] = {(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z),
                        (a.w + b.w) + (c.w + d.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i + j >= n) break;
      float o = r[j];
      if (beta != 0.f) o += beta * to_float(out[i + j]);
      out[i + j] = from_float<TO>(o);
    }
  }
}

template <typename T, int BI, int BJ, int KB, int WI, int WJ, int PF, bool XF>
void launch_cfg_x(const WgradParams& p, hipStream_t stream) {
  constexpr int BKP = KB / (int)sizeof(T);
  constexpr int lds = 2 * BKP * ((BI + BJ) * (int)sizeof(T) + 128);
  const long blocks = (long)(ceil_div(p.N, BI) + ceil_div(p.N2, BI)) * ceil_div(p.Ktot, BJ) * p.splits;
  const int es = (int)sizeof(T);
  const bool lin = p.ntaps == 1 && p.SY == 1 && p.SX == 1 && p.taps_zero &&
                   (long)p.div_ohow.d == (long)p.IH * p.IW && (int)p.div_ow.d == p.IW &&
                   ((double)p.pix_per_split + 2.0 * BKP) * p.ldy * es < 4294967296.0 &&
                   ((double)p.pix_per_split + 2.0 * BKP) * p.ldy2 * es < 4294967296.0 &&
                   ((double)p.pix_per_split + 2.0 * BKP) * p.ldx * es < 4294967296.0;
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<T, BI, BJ, KB, WI, WJ, false, PF, XF>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<T, BI, BJ, KB, WI, WJ, true, PF, XF>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  if (lin)
    hipLaunchKernelGGL((wgrad_kernel<T, BI, BJ, KB, WI, WJ, true, PF, XF>), dim3((unsigned)blocks, (unsigned)p.groups),
                       dim3(64 * WI * WJ), lds, stream, p);
  else
    hipLaunchKernelGGL((wgrad_kernel<T, BI, BJ, KB, WI, WJ, false, PF, XF>), dim3((unsigned)blocks, (unsigned)p.groups),
                       dim3(64 * WI * WJ), lds, stream, p);
}

template <typename T, int BI, int BJ, int KB, int WI = 2, int WJ = 2, int PF = 1>
void launch_cfg(const WgradParams& p, hipStream_t stream) {
  if constexpr (sizeof(T) == 2) {
    if (p.x_scale != nullptr) {
      launch_cfg_x<T, BI, BJ, KB, WI, WJ, PF, true>(p, stream);
      return;
    }
  }
  launch_cfg_x<T, BI, BJ, KB, WI, WJ, PF, false>(p, stream);
}

// Tile configurations (output channels x im2col columns x bytes of pixels per K tile); picked per
// problem shape from measurements (bench/conv_layers.py --sweep-wgrad; wgrad_force_config).
struct WCfg { int bi, bj, kb, threads; };
constexpr WCfg kWCfgs[] = {
    {128, 128, 64, 256}, {64, 64, 64, 256}, {64, 256, 64, 256}, {64, 128, 64, 256}, {128, 64, 64, 256},
    {256, 256, 64, 512}, {256, 128, 64, 512}, {128, 256, 64, 512},      // 5..7: 8-wave workgroups
    {128, 128, 64, 256}, {256, 128, 64, 512}, {64, 128, 64, 256},       // 8..10: cfg 0 / 6 / 3 with two tiles in flight
};
constexpr int kNumWCfgs = sizeof(kWCfgs) / sizeof(kWCfgs[0]);
int g_force_wcfg = -1;
int g_target_blocks = 768;

// From the per-layer sweeps (profiles/r1_wgrad_sweep.md, GPU-side timing through hipGraph replay): two K tiles
// in flight (PF = 2) win on the 1x1 layers, the 8-wave 256x128 tile on the big 3x3 layers (it halves the
// re-reads of the im2col operand), and the 64-channel layers want a tile as wide as their short side allows.
int choose_wcfg(long M, int N, int Ktot) {
  (void)M;
  if (g_force_wcfg >= 0 && g_force_wcfg < kNumWCfgs) return g_force_wcfg;
  if (N <= 64) return Ktot <= 64 ? 1 : (Ktot <= 256 ? 10 : 2);
  if (Ktot <= 64) return 4;
  if (Ktot >= 2304 && N >= 256) return 9;
  if (Ktot >= 1152 && N <= 128) return 0;
  return 8;
}

template <typename T>
void dispatch(const WgradParams& p, hipStream_t stream) {
  switch (choose_wcfg(p.M, p.N, p.Ktot)) {
    case 0: launch_cfg<T, 128, 128, 64>(p, stream); return;
    case 1: launch_cfg<T, 64, 64, 64>(p, stream); return;
    case 2: launch_cfg<T, 64, 256, 64>(p, stream); return;
    case 3: launch_cfg<T, 64, 128, 64>(p, stream); return;
    case 4: launch_cfg<T, 128, 64, 64>(p, stream); return;
    case 5: launch_cfg<T, 256, 256, 64, 2, 4>(p, stream); return;
    case 6: launch_cfg<T, 256, 128, 64, 4, 2>(p, stream); return;
    case 7: launch_cfg<T, 128, 256, 64, 2, 4>(p, stream); return;
    case 8: launch_cfg<T, 128, 128, 64, 2, 2, 2>(p, stream); return;
    case 9: launch_cfg<T, 256, 128, 64, 4, 2, 2>(p, stream); return;
    case 10: launch_cfg<T, 64, 128, 64, 2, 2, 2>(p, stream); return;
    default: throw std::logic_error("wgrad: bad tile configuration");
  }
}

}  // namespace

void wgrad_force_config(int cfg) { g_force_wcfg = cfg; }
void wgrad_set_target_blocks(int n) { g_target_blocks = n > 0 ? n : 768; }
int wgrad_num_configs() { return kNumWCfgs; }

int wgrad_choose_splits(long M, int N, int Ktot, int dtype, int* pix_per_split, int groups) {
  if (groups < 1) groups = 1;
  const WCfg c = kWCfgs[choose_wcfg(M, N, Ktot)];
  const int bkp = c.kb / (dtype == kF32 ? 4 : 2);
  const long tiles = (long)ceil_div(N, c.bi) * ceil_div(Ktot, c.bj) * groups;
  // aim for ~3 workgroups per CU, keep >= 8 K tiles per workgroup, and bound the slab size
  const long target = (long)g_target_blocks * 256 / c.threads;     // same number of waves in flight
  long splits = (target + tiles - 1) / tiles;
  const long max_by_work = (M + 8l * bkp - 1) / (8l * bkp);
  if (splits > max_by_work) splits = max_by_work;
  const long out_elems = (long)N * Ktot * groups;
  const long max_by_slab = (64l << 20) / (out_elems * 4) + 1;   // <= ~64 MiB of partials
  if (splits > max_by_slab) splits = max_by_slab;
  if (splits < 1) splits = 1;
  long pps = (M + splits - 1) / splits;
  pps = (pps + bkp - 1) / bkp * bkp;
  splits = (M + pps - 1) / pps;
  *pix_per_split = (int)pps;
  return (int)splits;
}

void wgrad_launch(const WgradParams& p_in, int dtype, hipStream_t stream) {
  WgradParams p = p_in;
  const int vec = dtype == kF32 ? 4 : 8;
  if (p.M <= 0 || p.N <= 0 || p.Ktot <= 0) return;
  if (p.dY2 == nullptr) p.N2 = p.ldy2 = 0;
  if (p.N2 > 0 && (p.groups > 1 || !(p.ntaps == 1 && p.taps_zero) || p.N2 % vec != 0 || p.ldy2 % vec != 0))
    throw std::invalid_argument("wgrad: a second gradient operand needs a dense 1x1 problem and 16-byte aligned rows");
  if (p.groups <= 1) {
    p.groups = 1;
    p.y_gstride = p.x_gstride = 0;
    p.slab_rows = p.N + p.N2;
  } else {
    if (p.y_gstride % vec || p.x_gstride % vec || p.groups > 65535 || p.slab_rows != p.groups * p.N)
      throw std::invalid_argument("wgrad: bad group layout");
  }
  if (p.N % vec != 0 || p.Cred % vec != 0 || p.ldy % vec != 0 || p.ldx % vec != 0)
    throw std::invalid_argument("wgrad: channel counts / strides must be multiples of 16 bytes");
  if (p.Ktot != p.ntaps * p.Cred) throw std::invalid_argument("wgrad: Ktot != ntaps*Cred");
  if ((p.x_scale != nullptr) != (p.x_shift != nullptr) || (p.x_scale != nullptr && (dtype == kF32 || p.groups > 1)))
    throw std::invalid_argument("wgrad: apply-on-load needs scale AND shift, a 16-bit dtype and a dense problem");
  if ((long)p.splits * p.pix_per_split < p.M) throw std::invalid_argument("wgrad: splits do not cover M");
  {
    const int bkp = kWCfgs[choose_wcfg(p.M, p.N, p.Ktot)].kb / (dtype == kF32 ? 4 : 2);
    if (p.splits > 1 && p.pix_per_split % bkp != 0)
      throw std::invalid_argument("wgrad: pix_per_split must be a multiple of the K tile");
  }
  if (dtype == kF32) dispatch<float>(p, stream);
  else if (dtype == kBF16) dispatch<bf16_t>(p, stream);
  else if (dtype == kF16) dispatch<f16_t>(p, stream);
  else throw std::invalid_argument("wgrad: unsupported dtype");
  DPA_CHECK_HIP(hipGetLastError());
}

void slab_reduce_launch(const float* slab, int splits, long n, void* out, int out_dtype, float beta,
                        hipStream_t stream) {
  if (n <= 0) return;
  const long blocks = ((n + 3) / 4 + 63) / 64;
  if (out_dtype == kF32)
    hipLaunchKernelGGL(slab_reduce_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, slab, splits, n,
                       reinterpret_cast<float*>(out), beta);
  else if (out_dtype == kBF16)
    hipLaunchKernelGGL(slab_reduce_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, stream, slab, splits, n,
                       reinterpret_cast<bf16_t*>(out), beta);
  else
    hipLaunchKernelGGL(slab_reduce_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, stream, slab, splits, n,
                       reinterpret_cast<f16_t*>(out), beta);
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
