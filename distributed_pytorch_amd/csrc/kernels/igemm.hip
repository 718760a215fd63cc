// Implicit-GEMM NHWC convolution on gfx950 MFMA (32x32x16 bf16/f16, 32x32x2 f32).
//
// One kernel serves convolution forward, data-gradient (stride 1 directly, stride s as s*s
// parity sub-problems) and Linear forward / input-gradient (a 1x1 "image"):
//
//     C[m][n] = sum_{t, c}  A[pix(m) + tap_t][c] * B[n][wtap_t * Cred + c]   (+ bias, ReLU)
//
// m enumerates the pixels of the (sub-)output grid, "tap_t" is a table entry (dy, dx, wtap).
// A rows are gathered 16 bytes at a time (global -> registers -> padded LDS), so every im2col
// chunk is one vector load; out-of-image taps are zero filled in registers.
//
// MFMA orientation: the weight tile is the A operand (rows = output channels) and the pixel
// tile the B operand, so each lane's accumulator registers run along output channels of ONE
// pixel.  The epilogue packs 4 channels per lane, bounces through LDS and leaves as full
// 16-byte row-contiguous stores; the same pass can emit per-channel (sum, sum^2) partials for
// the following BatchNorm (deterministic slab, no atomics).
//
// Replaces what cuDNN ran for `model(images)` / `loss.backward()` in the reference
// (nd_imagenet.py:291,302; imagenet_ddp_apex.py:372,381) -- SURVEY.md N-K1, N-K2, N-K10.
#include "../common.h"
#include "kernels.h"
#include <stdexcept>
#include <string>
#include <type_traits>

namespace dpa {

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    bf16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    f16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // 16 bytes = 4 consecutive k per lane half; both operands use the same k permutation
  // (step j covers k = {j, 4 + j} of the 8-wide slice), so the sum is unchanged.
  static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};

constexpr int kAxfMaxCred = 2048;     // LDS table: 8 bytes per reduction channel

// ---- BatchNorm + ReLU applied while the A tile is staged (kIgemmAxf) ---------------------------------------------
// The workgroup keeps (scale, shift) of every reduction channel in LDS, laid out per 16-byte chunk of the activation:
// [VEC scales][VEC shifts] (64 bytes for the 16-bit types), so the thread that stages chunk q reads four b128.
// The value written to LDS is bit-identical to what the stand-alone bn_apply pass stores: fp32 fma, ReLU, one
// rounding to T.  Out-of-image taps are masked AFTER this (the padding of the activation is zero, not relu(shift)).
template <typename T>
__device__ __forceinline__ void axf_fill_table(float* tab, const float* __restrict__ scale,
                                               const float* __restrict__ shift, int cred, int tid, int nthreads) {
  constexpr int VEC = 16 / (int)sizeof(T);
  for (int c = tid; c < cred; c += nthreads) {
    float* t = tab + (c / VEC) * (2 * VEC) + (c % VEC);
    t[0] = scale[c];
    t[VEC] = shift[c];
  }
}
template <typename T>
__device__ __forceinline__ uint4 axf_apply(const uint4& raw, const float* tab_q) {
  constexpr int VEC = 16 / (int)sizeof(T);
  static_assert(VEC == 8, "apply-on-load is instantiated for the 16-bit types only");
  const float4* t4 = reinterpret_cast<const float4*>(tab_q);
  const float4 s0 = t4[0], s1 = t4[1], h0 = t4[2], h1 = t4[3];
  const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
  return bn_relu_chunk<T>(raw, sc, sh);
}

// ---- epilogue shared by both main loops: registers -> LDS tile [pixel][channel] -> 16-byte global
// stores (+ bias, ReLU, fused accumulate, BatchNorm partial statistics).  All waves must be past their
// last read of the staging buffers (barrier) and have no LDS-DMA in flight when they enter.
// BNB (data-gradient launches with kIgemmBnBwd): the BatchNorm-backward partial sums are produced here.
// XM (kIgemmXMask, filter-row data-gradient launches): the output is the gradient of a ReLU output x (= p.bn_x, same
// layout as C): elements with x <= 0 are stored as 0, so the producer of x needs no relu_bwd pass of its own.
template <typename TOut, int BM, int BN, bool BNB, bool XM = false>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x16_t (&acc)[BN / 64][BM / 64],
                                               unsigned char* smem, int m0, int n0, int mt) {
  const long gofs = (long)blockIdx.y * p.c_gstride;      // first output channel of this group
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int LDE = BN * (int)sizeof(TOut) + 16;   // epilogue tile row bytes
  constexpr int CE = BN * (int)sizeof(TOut) / 16;    // 16-byte chunks per epilogue row
  constexpr int E_IT = BM * CE / 256;
  constexpr int VECO = 16 / (int)sizeof(TOut);
  static_assert(E_IT >= 1 && 256 % CE == 0, "chunk mapping");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  // ---- epilogue: registers -> LDS tile [pixel][channel] -> 16-byte global stores -----------
  // acc[a][b][r]: channel = wn*WTN + a*32 + (r&3) + 8*(r>>2) + 4*lh, pixel = wm*WTM + b*32 + l31
  unsigned char* se = smem;
  const bool relu = p.flags & kIgemmRelu;
  const int ce = tid % CE;
  const int ncol = n0 + ce * VECO;
  // fused BatchNorm-backward reduction: the extra operands (BN input, BN output or its bit mask) of the first half
  // of the rows are requested BEFORE the accumulators go through LDS, those of the second half right behind the
  // barrier: loaded one dependent row at a time these launches ran at half the memory rate (profiles/r2_notes.md)
  uint4 px[E_IT], pz[E_IT];
  uint32_t pbits[E_IT];           // bn_relu == 3: bn_z is the BN's bit mask (one byte per 16-byte chunk)
  const bool bn_fold = BNB && (p.flags & kIgemmBnFold);     // masked store, sum dz_m only, bn_x not read (kernels.h)
  auto bnb_request = [&](int first, int last) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < E_IT; ++i) {
      if (i < first || i >= last) continue;
      const uint32_t m = m0 + (tid + i * 256) / CE;
      const bool ok = m < (uint32_t)p.M && ncol + VECO <= p.N;       // BNB launches: N % VECO == 0
      size_t orow = m;
      if (p.flags & kIgemmScatter) {       // a parity class of a strided data gradient: the row this tile row is stored to
        const uint32_t n = fdiv(ok ? m : 0u, p.div_ohow);
        const uint32_t rem = (ok ? m : 0u) - n * p.div_ohow.d;
        const uint32_t oy = fdiv(rem, p.div_ow);
        const uint32_t ox = rem - oy * p.div_ow.d;
        orow = ((size_t)n * p.OHF + oy * p.OSY + p.OY0) * p.OWF + ox * p.OSX + p.OX0;
      }
      const size_t goff = ok ? (orow * p.ldc + gofs + ncol) * sizeof(TOut) : 0;   // grouped: this group's slice
      if (bn_fold) px[i] = make_uint4(0u, 0u, 0u, 0u);
      else px[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.bn_x) + goff);
      if (p.bn_relu == 1) pz[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.bn_z) + goff);
      pbits[i] = p.bn_relu == 3 ? reinterpret_cast<const uint8_t*>(p.bn_z)[goff / 16] : 0u;
    }
  };
  // fused BatchNorm-backward reduction: per-channel constants of this thread's chunk column, requested AHEAD of the
  // rows' extra operands -- vmcnt retires in order, behind them they would make the first row wait for the last
  // request (profiles/r3/lean_epilogue_vmcnt.md)
  float bmu[VECO], bis[VECO], bsc[VECO], bsh[VECO];
#pragma unroll
  for (int j = 0; j < VECO; ++j) bmu[j] = bis[j] = bsc[j] = bsh[j] = 0.f;
  if constexpr (BNB) {
    if (!bn_fold) {
      // coef[4][stats_ld]: stats_ld = all channels of the BN (== N for a dense launch).  BNB launches have whole chunks
      // (N % VECO == 0): 16-byte loads -- per-element selects made these 32 separate dword loads per thread
      const long c = gofs + (ncol + VECO <= p.N ? ncol : 0);
      const bool recompute = (p.bn_relu & 3) == 2;
#pragma unroll
      for (int q = 0; q < VECO / 4; ++q) {
        const float4 m4 = reinterpret_cast<const float4*>(p.bn_coef + c)[q];
        const float4 i4 = reinterpret_cast<const float4*>(p.bn_coef + p.stats_ld + c)[q];
        bmu[4 * q] = m4.x; bmu[4 * q + 1] = m4.y; bmu[4 * q + 2] = m4.z; bmu[4 * q + 3] = m4.w;
        bis[4 * q] = i4.x; bis[4 * q + 1] = i4.y; bis[4 * q + 2] = i4.z; bis[4 * q + 3] = i4.w;
        if (recompute) {
          const float4 s4 = reinterpret_cast<const float4*>(p.bn_coef + 2 * (long)p.stats_ld + c)[q];
          const float4 h4 = reinterpret_cast<const float4*>(p.bn_coef + 3 * (long)p.stats_ld + c)[q];
          bsc[4 * q] = s4.x; bsc[4 * q + 1] = s4.y; bsc[4 * q + 2] = s4.z; bsc[4 * q + 3] = s4.w;
          bsh[4 * q] = h4.x; bsh[4 * q + 1] = h4.y; bsh[4 * q + 2] = h4.z; bsh[4 * q + 3] = h4.w;
        }
      }
    }
  }
  constexpr int E_HALF = E_IT > 1 ? E_IT / 2 : 1;
  if constexpr (BNB || XM) bnb_request(0, E_HALF);
#pragma unroll
  for (int a = 0; a < NI; ++a) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cch = wn * WTN + a * 32 + 8 * g + 4 * lh;
      float bs[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias != nullptr) {
        if (n0 + cch + 3 < p.N && ((gofs + n0) & 3) == 0) {       // one 16-byte load (cch is a multiple of 4)
          const float4 b4 = *reinterpret_cast<const float4*>(p.bias + gofs + n0 + cch);
          bs[0] = b4.x; bs[1] = b4.y; bs[2] = b4.z; bs[3] = b4.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bs[j] = (n0 + cch + j < p.N) ? p.bias[gofs + n0 + cch + j] : 0.f;
        }
      }
#pragma unroll
      for (int b = 0; b < MI; ++b) {
        const int pix = wm * WTM + b * 32 + l31;
        TOut o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[a][b][4 * g + j] + bs[j];
          if (relu) v = fmaxf(v, 0.f);
          o[j] = from_float<TOut>(v);
        }
        unsigned char* dst = se + pix * LDE + cch * sizeof(TOut);
        if (sizeof(TOut) == 2) {
          uint2 w;
          __builtin_memcpy(&w, o, 8);
          *reinterpret_cast<uint2*>(dst) = w;
        } else {
          uint4 w;
          __builtin_memcpy(&w, o, 16);
          *reinterpret_cast<uint4*>(dst) = w;
        }
      }
    }
  }
  // LDS-only barrier: __syncthreads() would also wait for the requests above (vmcnt(0))
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if constexpr (BNB || XM) bnb_request(E_HALF, E_IT);

  float s1[VECO], s2[VECO];
#pragma unroll
  for (int j = 0; j < VECO; ++j) s1[j] = s2[j] = 0.f;
  // BatchNorm statistics of this tile leave as (sum, M2 = sum of squared deviations from the TILE mean) -- the
  // partials of Chan's parallel variance -- never as a raw sum of squares, which cancels catastrophically in fp32
  // when |mean| >> std.  They are accumulated as sums of (v - k) and (v - k)^2 with k = the tile's first row (a
  // sample, so |v - k| ~ std) and converted when the tile's waves meet.
  float kshift[VECO];
#pragma unroll
  for (int j = 0; j < VECO; ++j) kshift[j] = 0.f;
  if constexpr (!BNB) {
    if (p.stats != nullptr) {
      Vec16<TOut> k0;
      k0.load(se + ce * 16);
#pragma unroll
      for (int j = 0; j < VECO; ++j) kshift[j] = k0.get(j);
    }
  }
  char* __restrict__ Cg = reinterpret_cast<char*>(p.C) + gofs * sizeof(TOut);
#pragma unroll
  for (int i = 0; i < E_IT; ++i) {
    const int row = (tid + i * 256) / CE;
    const uint32_t m = m0 + row;
    if (m >= (uint32_t)p.M || ncol >= p.N) continue;
    Vec16<TOut> v;
    v.load(se + row * LDE + ce * 16);
    size_t orow = m;
    if (p.flags & kIgemmScatter) {
      const uint32_t n = fdiv(m, p.div_ohow);
      const uint32_t rem = m - n * p.div_ohow.d;
      const uint32_t oy = fdiv(rem, p.div_ow);
      const uint32_t ox = rem - oy * p.div_ow.d;
      orow = ((size_t)n * p.OHF + oy * p.OSY + p.OY0) * p.OWF + ox * p.OSX + p.OX0;
    }
    const size_t goff = (orow * p.ldc + ncol) * sizeof(TOut);
    char* dst = Cg + goff;
    if (p.R != nullptr) {      // fused accumulate (gradient of a tensor with two consumers): one rounding
      const char* rsrc = reinterpret_cast<const char*>(p.R) + gofs * sizeof(TOut) + goff;
      // masked accumulate: R is a gradient that still has to pass a ReLU (its mask bits live next to it)
      const uint32_t rbits = p.r_mask != nullptr ? p.r_mask[(gofs * sizeof(TOut) + goff) / 16] : 0xffu;
      if (ncol + VECO <= p.N) {
        Vec16<TOut> r;
        r.load(rsrc);
#pragma unroll
        for (int j = 0; j < VECO; ++j) v.set(j, v.get(j) + (((rbits >> j) & 1u) ? r.get(j) : 0.f));
      } else {
#pragma unroll
        for (int j = 0; j < VECO; ++j) {   // constant j: keeps v in registers (a runtime index sends it to scratch)
          if (ncol + j < p.N && ((rbits >> j) & 1u)) {
            TOut t = *reinterpret_cast<const TOut*>(rsrc + j * sizeof(TOut));
            v.set(j, v.get(j) + to_float(t));
          }
        }
      }
    }
    if constexpr (BNB) {
      if (bn_fold && p.bn_relu == 3) {       // the gradient leaves already masked (kIgemmBnFold)
#pragma unroll
        for (int j = 0; j < VECO; ++j)
          if (!((pbits[i] >> j) & 1u)) v.set(j, 0.f);
      }
    }
    if constexpr (XM) {        // whole chunks only (the launch checks N % VECO == 0)
      Vec16<TOut> xb;
      xb.raw = px[i];
#pragma unroll
      for (int j = 0; j < VECO; ++j)
        if (!(xb.get(j) > 0.f)) v.set(j, 0.f);
    }
    if (ncol + VECO <= p.N) {
      if (p.flags & kIgemmNtStore) v.store_nt(dst);
      else v.store(dst);
    } else {
#pragma unroll
      for (int j = 0; j < VECO; ++j)
        if (ncol + j < p.N) *reinterpret_cast<TOut*>(dst + j * sizeof(TOut)) = from_float<TOut>(v.get(j));
    }
    if constexpr (BNB) {
      // v holds the values as stored (rounded to TOut): exactly what BatchNorm backward will read back
      if (ncol + VECO <= p.N) {
        Vec16<TOut> xb, zb;
        xb.raw = px[i];
        zb.raw = pz[i];
#pragma unroll
        for (int j = 0; j < VECO; ++j) {
          float d = v.get(j);
          const float xv = xb.get(j);
          if (p.bn_relu == 1 && !(zb.get(j) > 0.f)) d = 0.f;
          if (p.bn_relu == 3 && !((pbits[i] >> j) & 1u)) d = 0.f;
          if ((p.bn_relu & 3) == 2) {           // mask recomputed from the BN input; bit 2: ReLU6 (0 < y < 6)
            const float yv = to_float(from_float<TOut>(bsc[j] * xv + bsh[j]));
            if (!(yv > 0.f) || ((p.bn_relu & 4) && !(yv < 6.f))) d = 0.f;
          }
          s1[j] += d;
          s2[j] += d * ((xv - bmu[j]) * bis[j]);
        }
      }
    } else if (p.stats != nullptr) {
#pragma unroll
      for (int j = 0; j < VECO; ++j) {
        const float f = v.get(j) - kshift[j];
        s1[j] += f;
        s2[j] += f * f;
      }
    }
  }
  if (p.stats != nullptr) {
    // threads sharing a chunk column: tid = ce (mod CE). reduce inside the wave, then over waves.
#pragma unroll
    for (int j = 0; j < VECO; ++j) {
#pragma unroll
      for (int o = 32; o >= CE; o >>= 1) {
        s1[j] += __shfl_xor(s1[j], o, 64);
        s2[j] += __shfl_xor(s2[j], o, 64);
      }
    }
    // LDS-only barrier: __syncthreads() would first drain the tile's global stores (vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // epilogue tile fully consumed
    asm volatile("" ::: "memory");
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][BN] + shifts [BN]
    if (lane < CE) {
#pragma unroll
      for (int j = 0; j < VECO; ++j) {
        red[(wave * 2 + 0) * BN + ce * VECO + j] = s1[j];
        red[(wave * 2 + 1) * BN + ce * VECO + j] = s2[j];
        if (!BNB && wave == 0) red[8 * BN + ce * VECO + j] = kshift[j];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (BNB) {
      for (int c = tid; c < 2 * BN; c += 256) {
        const int which = c / BN, col = c - which * BN;
        if (n0 + col < p.N) {
          const float s = red[(0 * 2 + which) * BN + col] + red[(1 * 2 + which) * BN + col] +
                          red[(2 * 2 + which) * BN + col] + red[(3 * 2 + which) * BN + col];
          p.stats[((size_t)mt * 2 + which) * p.stats_ld + gofs + n0 + col] = s;
        }
      }
    } else {
      const float nt_rows = (float)min(BM, p.M - m0);
      for (int col = tid; col < BN; col += 256) {
        if (n0 + col < p.N) {
          const float a = red[0 * BN + col] + red[2 * BN + col] + red[4 * BN + col] + red[6 * BN + col];
          const float b = red[1 * BN + col] + red[3 * BN + col] + red[5 * BN + col] + red[7 * BN + col];
          const float k = red[8 * BN + col];
          p.stats[((size_t)mt * 2 + 0) * p.stats_ld + gofs + n0 + col] = a + nt_rows * k;
          p.stats[((size_t)mt * 2 + 1) * p.stats_ld + gofs + n0 + col] = fmaxf(b - a * a / nt_rows, 0.f);
        }
      }
      // rows of this part, behind the slab: [tiles_m][2][stats_ld] floats, then [tiles_m] counts
      if (tid == 0 && n0 == 0 && blockIdx.y == 0) {
        const uint32_t tiles_m = gridDim.x / ((p.N + BN - 1) / BN);
        p.stats[(size_t)tiles_m * 2 * p.stats_ld + mt] = nt_rows;
      }
    }
  }
}

// BM: pixel rows per block, BN: output channels per block, BKB: bytes of K per row per K tile.
// 256 threads = 2 (pixel) x 2 (channel) waves.
// TAIL (with ALIGNED): Cred is a multiple of 16 bytes but not of the K tile (grouped convolutions with 24 / 56 / 120
// channels per group -- RegNet): every tap takes ceil(Cred / BK) K tiles and the chunks of the last one that lie
// beyond the tap's channels are loaded from offset 0 and zeroed when the tile is written to LDS.
template <typename T, typename TOut, int BM, int BN, int BKB, int MINB, bool ALIGNED, bool BNB, bool AXF = false,
          bool TAIL = false>
__global__ __launch_bounds__(256, MINB) void igemm_kernel(const IgemmParams p) {
  static_assert(!AXF || (ALIGNED && sizeof(T) == 2), "apply-on-load: aligned 16-bit problems");
  static_assert(!TAIL || (ALIGNED && !AXF && !BNB), "K-tile tails: plain aligned launches");
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int BK = BKB / (int)sizeof(T);
  constexpr int CPR = BKB / 16;          // 16-byte chunks per tile row
  constexpr int LDR = BKB + 16;          // padded LDS row (odd multiple of 16 B: conflict-free b128)
  constexpr int A_IT = BM * CPR / 256;
  constexpr int B_IT = BN * CPR / 256;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int STAGE = (BM + BN) * LDR;
  static_assert(A_IT >= 1 && B_IT >= 1 && MI >= 1 && NI >= 1, "tile too small");
  static_assert(256 % CPR == 0, "chunk mapping");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const uint32_t tiles_n = (p.N + BN - 1) / BN;
  const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / tiles_n, nt = lid - mt * tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  // The gather base is the first image this tile touches: the 32-bit byte offsets of the aligned path then only have
  // to span the few images of ONE tile, so activation tensors of any size (> 4 GiB at large per-GPU batches) stay on
  // the fast path (the host checks the per-tile span, dispatch_tiles).
  const uint32_t n_tile0 = fdiv((uint32_t)m0, p.div_ohow);
  const char* __restrict__ Ag = reinterpret_cast<const char*>(p.A) + (size_t)blockIdx.y * p.a_gstride * sizeof(T) +
                                (size_t)n_tile0 * p.IH * p.IW * p.lda * sizeof(T);
  const char* __restrict__ Bg = reinterpret_cast<const char*>(p.B) + (size_t)blockIdx.y * p.b_gstride * sizeof(T);

  // ---- per-thread gather state: rows are fixed for the whole K loop -----------------------
  const int ch = tid % CPR;
  int iy0[A_IT], ix0[A_IT], pbase[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int row = (tid + i * 256) / CPR;
    const uint32_t m = m0 + row;
    if (m < (uint32_t)p.M) {
      const uint32_t n = fdiv(m, p.div_ohow);
      const uint32_t rem = m - n * p.div_ohow.d;
      const uint32_t oy = fdiv(rem, p.div_ow);
      const uint32_t ox = rem - oy * p.div_ow.d;
      iy0[i] = oy * p.SY;
      ix0[i] = ox * p.SX;
      pbase[i] = (n - n_tile0) * p.IH * p.IW;
    } else {
      iy0[i] = -(1 << 28);
      ix0[i] = 0;
      pbase[i] = 0;
    }
  }
  size_t brow[B_IT];
  bool bok[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int n = n0 + (tid + i * 256) / CPR;
    bok[i] = n < p.N;
    brow[i] = (size_t)(bok[i] ? n : 0) * p.ldb;
  }

  uint4 ra[A_IT], rb[B_IT];
  // AXF: (scale, shift) table behind the two staging buffers; chunk index of the K tile held in ra[]
  float* axf_tab = reinterpret_cast<float*>(smem + 2 * STAGE);
  int axf_q = 0;
  if constexpr (AXF) {
    axf_fill_table<T>(axf_tab, p.a_scale, p.a_shift, p.Cred, tid, 256);
    __syncthreads();
  }
  const int kpt = ALIGNED ? (TAIL ? (p.Cred + BK - 1) / BK : p.Cred / BK) : 1;           // K tiles per tap
  const int cpt = p.Cred / VEC;                        // chunks per tap
  const int nk = ALIGNED ? p.ntaps * kpt : (p.ntaps * cpt + CPR - 1) / CPR;

  // ALIGNED (Cred % BK == 0, operands < 4 GiB -- checked on the host): the gather address of a row only
  // changes when the TAP changes, so the per-row work (bounds test, pixel offset) is done once per tap into
  // 32-bit byte offsets; a K tile inside the tap is "scalar base + offset" loads with no vector address
  // arithmetic at all.  Counters showed the old per-tile 64-bit address math made the loop issue bound.
  uint32_t aoff[A_IT], amask[A_IT], boff[B_IT], bmask[B_IT];
  uint32_t tmask = 0xffffffffu;                        // TAIL: this thread's chunk of the tile in ra/rb is inside the tap
  size_t kb_tap = 0;
  int lt_tap = 0, lt_kc = 0;                           // next K tile to load: tap, tile inside the tap
  if (ALIGNED) {
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      boff[i] = (uint32_t)((brow[i] + ch * VEC) * sizeof(T));
      bmask[i] = bok[i] ? 0xffffffffu : 0u;
    }
  }
  auto set_tap = [&](int t) {
    // wave-uniform values: keep them in SGPRs so the per-tile bases stay scalar
    const int dy = __builtin_amdgcn_readfirstlane(p.taps[3 * t + 0]);
    const int dx = __builtin_amdgcn_readfirstlane(p.taps[3 * t + 1]);
    kb_tap = (size_t)__builtin_amdgcn_readfirstlane(p.taps[3 * t + 2]) * p.Cred;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int iy = iy0[i] + dy, ix = ix0[i] + dx;
      const bool ok = (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const uint32_t pix = (uint32_t)(pbase[i] + iy * p.IW + ix);
      aoff[i] = ok ? (pix * (uint32_t)p.lda + (uint32_t)(ch * VEC)) * (uint32_t)sizeof(T) : 0u;
      amask[i] = ok ? 0xffffffffu : 0u;
    }
  };

  auto load_tile = [&](int kt) {
    if (ALIGNED) {
      if (lt_kc == 0) set_tap(lt_tap);
      const int c0 = lt_kc * BK;
      if constexpr (AXF) axf_q = lt_kc * CPR + ch;
      const char* Ak = Ag + (size_t)c0 * sizeof(T);
      const char* Bk = Bg + (kb_tap + c0) * sizeof(T);
#pragma unroll
      // unconditional loads (out-of-image rows read offset 0 and are masked when the tile is written to
      // LDS, after the MFMAs of the current tile): no branch and no wait between the loads and the compute
      for (int i = 0; i < A_IT; ++i) {
        if constexpr (TAIL) {
          const bool cv = c0 + ch * VEC < p.Cred;
          tmask = cv ? 0xffffffffu : 0u;
          ra[i] = *reinterpret_cast<const uint4*>(Ak + (cv ? aoff[i] : 0u));
        } else {
          ra[i] = *reinterpret_cast<const uint4*>(Ak + aoff[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        if constexpr (TAIL) rb[i] = *reinterpret_cast<const uint4*>(Bk + (tmask ? boff[i] : 0u));
        else rb[i] = *reinterpret_cast<const uint4*>(Bk + boff[i]);
      }
      if (++lt_kc == kpt) {
        lt_kc = 0;
        ++lt_tap;
      }
      return;
    }
    const int kc = kt * CPR + ch;
    int t = kc / cpt;
    const int cA = (kc - t * cpt) * VEC;
    const bool tok = t < p.ntaps;
    t = tok ? t : 0;
    const int dy = p.taps[3 * t + 0];
    const int dx = p.taps[3 * t + 1];
    const size_t kB = (size_t)p.taps[3 * t + 2] * p.Cred + cA;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int iy = iy0[i] + dy, ix = ix0[i] + dx;
      const bool ok = tok && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const size_t off = ((size_t)(pbase[i] + iy * p.IW + ix) * p.lda + cA) * sizeof(T);
      // unconditional load + mask at the LDS write (a select behind the load makes the wave wait for it here)
      amask[i] = ok ? 0xffffffffu : 0u;
      ra[i] = *reinterpret_cast<const uint4*>(Ag + (ok ? off : (size_t)0));
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const bool ok = tok && bok[i];
      bmask[i] = ok ? 0xffffffffu : 0u;
      rb[i] = *reinterpret_cast<const uint4*>(Bg + (ok ? (brow[i] + kB) * sizeof(T) : (size_t)0));
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* sa = smem + buf * STAGE;
    unsigned char* sb = sa + BM * LDR;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = (tid + i * 256) / CPR;
      uint4 v = ra[i];
      if constexpr (AXF) v = axf_apply<T>(v, axf_tab + axf_q * (32 / (int)sizeof(T)));
      const uint32_t am = TAIL ? amask[i] & tmask : amask[i];
      v.x &= am; v.y &= am; v.z &= am; v.w &= am;
      *reinterpret_cast<uint4*>(sa + row * LDR + ch * 16) = v;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = (tid + i * 256) / CPR;
      uint4 v = rb[i];
      const uint32_t bm = TAIL ? bmask[i] & tmask : bmask[i];
      v.x &= bm; v.y &= bm; v.z &= bm; v.w &= bm;
      *reinterpret_cast<uint4*>(sb + row * LDR + ch * 16) = v;
    }
  };

  f32x16_t acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int l31 = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const unsigned char* sa = smem + cur * STAGE;
    const unsigned char* sb = sa + BM * LDR;
#pragma unroll
    for (int ks = 0; ks < BKB / 32; ++ks) {
      uint4 wf[NI], xf[MI];
#pragma unroll
      for (int a = 0; a < NI; ++a)
        wf[a] = *reinterpret_cast<const uint4*>(sb + (wn * WTN + a * 32 + l31) * LDR + ks * 32 + lh * 16);
#pragma unroll
      for (int b = 0; b < MI; ++b)
        xf[b] = *reinterpret_cast<const uint4*>(sa + (wm * WTM + b * 32 + l31) * LDR + ks * 32 + lh * 16);
#pragma unroll
      for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  igemm_epilogue<TOut, BM, BN, BNB>(p, acc, smem, m0, n0, mt);
}

// ------------------------------------------------------------------------------------------------
// Filter-row kernel for 3x3-like stride-1 "same" convolutions (kIgemmRow3): the three taps of a filter row read
// the SAME input rows shifted by one pixel, so ONE staged A tile of BM + 2 consecutive (flattened) pixels serves
// all three -- tap t of output pixel i reads LDS row i + t.  The implicit-GEMM kernels above are bound by the
// L2 -> LDS operand traffic (profiles/r1_notes.md); this cuts the activation side of it 3x (9 -> 3 tile loads per
// pixel and channel tile).  A lane whose shifted pixel falls off its image row (ox + dx outside [0, IW)) gets a
// zero fragment: that is the convolution's padding, and it also kills the wrap into the neighbouring row / image.
// Needs: taps in groups of 3 with equal dy and dx = x0, x0+1, x0+2 (same x0 for all groups), SX == 1, the output
// grid as wide as the gathered image, Cred % BK == 0, operands addressable with 32 bits.
// One LDS stage (the next stage waits in registers): 41 KB for 128x128x64B, three workgroups per CU.
// TAIL: Cred % BK != 0 (16-byte aligned group widths): ceil(Cred / BK) K tiles per filter row, the chunks beyond the
// row's channels masked exactly as in igemm_kernel.
template <typename T, typename TOut, int BM, int BN, int BKB, int MINB, bool BNB, bool AXF = false, bool TAIL = false,
          bool XM = false>
__global__ __launch_bounds__(256, MINB) void igemm_row3_kernel(const IgemmParams p) {
  static_assert(!XM || (!BNB && !AXF && !TAIL), "masked store: plain launches");
  static_assert(!AXF || sizeof(T) == 2, "apply-on-load: 16-bit problems");
  static_assert(!TAIL || (!AXF && !BNB), "K-tile tails: plain launches");
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int BK = BKB / (int)sizeof(T);
  constexpr int CPR = BKB / 16;
  constexpr int LDR = BKB + 16;
  constexpr int AR = BM + 2;                              // staged pixel rows
  constexpr int A_IT = (AR * CPR + 255) / 256;
  constexpr int B_IT = BN * CPR / 256;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  static_assert(B_IT >= 1 && MI >= 1 && NI >= 1 && 256 % CPR == 0, "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sa = smem;
  unsigned char* sb = smem + AR * LDR;                    // [3][BN] rows

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const uint32_t tiles_n = (p.N + BN - 1) / BN;
  const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / tiles_n, nt = lid - mt * tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  const int x0 = __builtin_amdgcn_readfirstlane(p.taps[1]);        // dx of the first tap of every group
  // gather base = first image of the staged pixel range (see igemm_kernel): 32-bit offsets span one tile only
  const long q_first = (long)m0 + x0;
  const uint32_t n_tile0 = fdiv((uint32_t)(q_first > 0 ? q_first : 0), p.div_ohow);
  const char* __restrict__ Ag = reinterpret_cast<const char*>(p.A) + (size_t)blockIdx.y * p.a_gstride * sizeof(T) +
                                (size_t)n_tile0 * p.IH * p.IW * p.lda * sizeof(T);
  const char* __restrict__ Bg = reinterpret_cast<const char*>(p.B) + (size_t)blockIdx.y * p.b_gstride * sizeof(T);

  const int ngroups = p.ntaps / 3;
  const int kpt = TAIL ? (p.Cred + BK - 1) / BK : p.Cred / BK;

  // ---- staged rows of this thread: flattened pixel q = m0 + row + x0, fixed for the whole kernel ------------
  const int ch = tid % CPR;
  int oys[A_IT], pixb[A_IT];         // oy * SY, and pixel index of (n, 0, ox) in the gathered image; < 0: no such pixel
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int row = (tid + i * 256) / CPR;
    const long q = (long)m0 + row + x0;
    if (row < AR && q >= 0 && q < (long)p.M) {
      const uint32_t n = fdiv((uint32_t)q, p.div_ohow);
      const uint32_t rem = (uint32_t)q - n * p.div_ohow.d;
      const uint32_t oy = fdiv(rem, p.div_ow);
      const uint32_t ox = rem - oy * p.div_ow.d;
      oys[i] = (int)oy * p.SY;
      pixb[i] = (int)((n - n_tile0) * p.IH * p.IW + ox);
    } else {
      oys[i] = 0;
      pixb[i] = -1;
    }
  }
  uint32_t boff[B_IT], bmask[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int n = n0 + (tid + i * 256) / CPR;
    const bool ok = n < p.N;
    boff[i] = (uint32_t)(((size_t)(ok ? n : 0) * p.ldb + ch * VEC) * sizeof(T));
    bmask[i] = ok ? 0xffffffffu : 0u;
  }
  // ---- per-lane padding masks of the fragment rows this lane feeds to the MFMAs ------------------------------
  uint32_t fmask[MI][3];
#pragma unroll
  for (int b = 0; b < MI; ++b) {
    const uint32_t m = (uint32_t)(m0 + wm * WTM + b * 32 + l31);
    int ox = 0;
    if (m < (uint32_t)p.M) {
      const uint32_t n = fdiv(m, p.div_ohow);
      const uint32_t rem = m - n * p.div_ohow.d;
      ox = (int)(rem - fdiv(rem, p.div_ow) * p.div_ow.d);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
      fmask[b][t] = ((unsigned)(ox + x0 + t) < (unsigned)p.IW) ? 0xffffffffu : 0u;
  }

  uint4 ra[A_IT], rb[3][B_IT];
  uint32_t aoff[A_IT], amask[A_IT];
  uint32_t tmask = 0xffffffffu;          // TAIL: this thread's chunk of the staged K tile lies inside the channels
  size_t kb_tap[3] = {0, 0, 0};
  int lt_g = 0, lt_kc = 0;
  float* axf_tab = reinterpret_cast<float*>(smem + (AR + 3 * BN) * LDR);   // behind the stage
  int axf_q = 0;
  if constexpr (AXF) {
    axf_fill_table<T>(axf_tab, p.a_scale, p.a_shift, p.Cred, tid, 256);
    __syncthreads();
  }
  auto set_group = [&](int g) {
    const int dy = __builtin_amdgcn_readfirstlane(p.taps[9 * g]);
#pragma unroll
    for (int t = 0; t < 3; ++t)
      kb_tap[t] = (size_t)__builtin_amdgcn_readfirstlane(p.taps[9 * g + 3 * t + 2]) * p.Cred;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int iy = oys[i] + dy;
      const bool ok = pixb[i] >= 0 && (unsigned)iy < (unsigned)p.IH;
      aoff[i] = ok ? ((uint32_t)(pixb[i] + iy * p.IW) * (uint32_t)p.lda + (uint32_t)(ch * VEC)) * (uint32_t)sizeof(T) : 0u;
      amask[i] = ok ? 0xffffffffu : 0u;
    }
  };
  auto load_stage = [&]() {
    if (lt_kc == 0) set_group(lt_g);
    const int c0 = lt_kc * BK;
    if constexpr (AXF) axf_q = lt_kc * CPR + ch;
    const char* Ak = Ag + (size_t)c0 * sizeof(T);
    if constexpr (TAIL) tmask = (c0 + ch * VEC < p.Cred) ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) ra[i] = *reinterpret_cast<const uint4*>(Ak + (TAIL && !tmask ? 0u : aoff[i]));
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const char* Bk = Bg + (kb_tap[t] + c0) * sizeof(T);
#pragma unroll
      for (int i = 0; i < B_IT; ++i) rb[t][i] = *reinterpret_cast<const uint4*>(Bk + (TAIL && !tmask ? 0u : boff[i]));
    }
    if (++lt_kc == kpt) {
      lt_kc = 0;
      ++lt_g;
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = (tid + i * 256) / CPR;
      if (row < AR) {
        uint4 v = ra[i];
        if constexpr (AXF) v = axf_apply<T>(v, axf_tab + axf_q * (32 / (int)sizeof(T)));
        const uint32_t am = TAIL ? amask[i] & tmask : amask[i];
        v.x &= am; v.y &= am; v.z &= am; v.w &= am;
        *reinterpret_cast<uint4*>(sa + row * LDR + ch * 16) = v;
      }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int row = (tid + i * 256) / CPR;
        uint4 v = rb[t][i];
        const uint32_t bm = TAIL ? bmask[i] & tmask : bmask[i];
        v.x &= bm; v.y &= bm; v.z &= bm; v.w &= bm;
        *reinterpret_cast<uint4*>(sb + (t * BN + row) * LDR + ch * 16) = v;
      }
  };

  f32x16_t acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nstages = ngroups * kpt;
  load_stage();
  store_stage();
  __syncthreads();
  for (int st = 0; st < nstages; ++st) {
    if (st + 1 < nstages) load_stage();                   // in flight while this stage is consumed
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int ks = 0; ks < BKB / 32; ++ks) {
        uint4 wf[NI], xf[MI];
#pragma unroll
        for (int a = 0; a < NI; ++a)
          wf[a] = *reinterpret_cast<const uint4*>(sb + (t * BN + wn * WTN + a * 32 + l31) * LDR + ks * 32 + lh * 16);
#pragma unroll
        for (int b = 0; b < MI; ++b) {
          uint4 v = *reinterpret_cast<const uint4*>(sa + (wm * WTM + b * 32 + l31 + t) * LDR + ks * 32 + lh * 16);
          const uint32_t k = fmask[b][t];
          v.x &= k; v.y &= k; v.z &= k; v.w &= k;
          xf[b] = v;
        }
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
          for (int b = 0; b < MI; ++b) Mma<T>::run(acc[a][b], wf[a], xf[b]);
      }
    }
    __syncthreads();                                      // every wave is done reading this stage
    if (st + 1 < nstages) {
      store_stage();
      __syncthreads();
    }
  }

  igemm_epilogue<TOut, BM, BN, BNB, XM>(p, acc, smem, m0, n0, mt);
}

template <typename T, typename TOut, int BM, int BN, int BKB, int MINB, bool BNB = false, bool AXF = false,
          bool TAIL = false, bool XM = false>
void launch_row3(const IgemmParams& p, hipStream_t stream) {
  constexpr int LDR = BKB + 16;
  constexpr int stage_bytes = (BM + 2 + 3 * BN) * LDR;
  constexpr int epi_bytes = BM * (BN * (int)sizeof(TOut) + 16);
  constexpr int red_bytes = 9 * BN * 4;
  constexpr int lds0 = stage_bytes > epi_bytes ? (stage_bytes > red_bytes ? stage_bytes : red_bytes)
                                               : (epi_bytes > red_bytes ? epi_bytes : red_bytes);
  int lds = lds0;
  if (AXF) {            // (scale, shift) table behind the stage
    if (p.Cred > kAxfMaxCred) throw std::invalid_argument("igemm: apply-on-load supports at most 2048 reduction channels");
    if (lds < stage_bytes + 8 * p.Cred) lds = stage_bytes + 8 * p.Cred;
  }
  auto kern = igemm_row3_kernel<T, TOut, BM, BN, BKB, MINB, BNB, AXF, TAIL, XM>;
  static bool attr_done = false;
  if (!attr_done && lds0 + 8 * kAxfMaxCred > 64 * 1024) {
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds0 + 8 * kAxfMaxCred));
    attr_done = true;
  }
  const long tiles = (long)ceil_div(p.M, BM) * ceil_div(p.N, BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)p.groups), dim3(256), lds, stream, p);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA main loop: the same tile computation, staged with global_load_lds_dwordx4 into an S-deep ring
// of unpadded LDS tiles so that S-1 K tiles (not one) are in flight per workgroup while one is consumed --
// the register-staged loop above is bound by bytes in flight per CU on the memory-bound layers.
//  * one wave-instruction moves 64 x 16 B = 1 KiB: RPI = 1024/BKB whole tile rows, lane -> (row, chunk)
//  * the LDS image is lane-linear, so the bank swizzle lives on the SOURCE address and on the fragment
//    read:  physical chunk = logical chunk ^ ((row / (256/BKB)) & (BKB/16 - 1))   (conflict-free b128 reads)
//  * out-of-image taps / rows beyond M or N read a zero page instead of being zero-filled in registers
//  * one raw s_barrier per K tile; the wave waits only for ITS OWN pieces of tile kt (counted vmcnt) before
//    the barrier, leaving the S-2 younger tiles in flight.  A stage is re-filled one barrier after its last read.
// Needs Cred % BK == 0 (a K tile never straddles taps).
__device__ uint4 g_zero_page[4];

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int OFF> __device__ __forceinline__ uint4 lds_read_b128_asm(uint32_t addr) {
  uint4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

// One K tile of the ring: fragments of step ks+1 are requested before the MFMAs of step ks are issued and
// land while they run.  Waits are lgkmcnt(0) (robust against any scalar load the compiler interleaves), each
// followed by sched_barrier(0) so no MFMA is hoisted above the wait that produces its operands.
template <typename T, int NI, int MI, int KS, int BKB>
__device__ __forceinline__ void ring_tile_mma(f32x16_t (&acc)[NI][MI], const uint32_t (&offA)[KS],
                                              const uint32_t (&offB)[KS], uint32_t sbase) {
  static_assert(NI <= 2 && MI <= 2, "immediate offsets below are written for <= 2 fragments per operand");
  uint4 wf[2][2], xf[2][2];
  wf[0][0] = lds_read_b128_asm<0>(offB[0] + sbase);
  if (NI > 1) wf[0][1] = lds_read_b128_asm<32 * BKB>(offB[0] + sbase);
  xf[0][0] = lds_read_b128_asm<0>(offA[0] + sbase);
  if (MI > 1) xf[0][1] = lds_read_b128_asm<32 * BKB>(offA[0] + sbase);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int c = ks & 1, n = c ^ 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < KS) {
      wf[n][0] = lds_read_b128_asm<0>(offB[ks + 1 < KS ? ks + 1 : 0] + sbase);
      if (NI > 1) wf[n][1] = lds_read_b128_asm<32 * BKB>(offB[ks + 1 < KS ? ks + 1 : 0] + sbase);
      xf[n][0] = lds_read_b128_asm<0>(offA[ks + 1 < KS ? ks + 1 : 0] + sbase);
      if (MI > 1) xf[n][1] = lds_read_b128_asm<32 * BKB>(offA[ks + 1 < KS ? ks + 1 : 0] + sbase);
    }
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) Mma<T>::run(acc[a][b], wf[c][a], xf[c][b]);
  }
}

template <typename T, typename TOut, int BM, int BN, int BKB, int S, int MINB>
__global__ __launch_bounds__(256, MINB) void igemm_glds_kernel(const IgemmParams p) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int BK = BKB / (int)sizeof(T);
  constexpr int CPR = BKB / 16;            // 16-byte chunks per tile row (4 or 8)
  constexpr int RPI = 64 / CPR;            // tile rows per wave-instruction
  constexpr int A_INST = BM / RPI / 4, B_INST = BN / RPI / 4;
  constexpr int IPT = A_INST + B_INST;     // LDS-DMA instructions per wave per K tile
  constexpr int STAGE = (BM + BN) * BKB;
  constexpr int SWR = 256 / BKB;           // tile rows per 256-byte LDS bank row
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  static_assert(A_INST >= 1 && B_INST >= 1 && MI >= 1 && NI >= 1, "tile too small");
  static_assert(S >= 2 && S <= 4 && (BKB == 64 || BKB == 128), "ring depth / K tile");
  static_assert((S - 1) * IPT < 64, "vmcnt range");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const uint32_t tiles_n = (p.N + BN - 1) / BN;
  const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / tiles_n, nt = lid - mt * tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  const char* __restrict__ Ag = reinterpret_cast<const char*>(p.A) + (size_t)blockIdx.y * p.a_gstride * sizeof(T);
  const char* __restrict__ Bg = reinterpret_cast<const char*>(p.B) + (size_t)blockIdx.y * p.b_gstride * sizeof(T);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  const int r_in = lane / CPR, cphys = lane % CPR;
  int iy0[A_INST], ix0[A_INST], pbase[A_INST], acol[A_INST];
#pragma unroll
  for (int i = 0; i < A_INST; ++i) {
    const int row = (wave * A_INST + i) * RPI + r_in;
    acol[i] = (cphys ^ ((row / SWR) & (CPR - 1))) * VEC;
    const uint32_t m = m0 + row;
    if (m < (uint32_t)p.M) {
      const uint32_t n = fdiv(m, p.div_ohow);
      const uint32_t rem = m - n * p.div_ohow.d;
      const uint32_t oy = fdiv(rem, p.div_ow);
      const uint32_t ox = rem - oy * p.div_ow.d;
      iy0[i] = oy * p.SY;
      ix0[i] = ox * p.SX;
      pbase[i] = n * p.IH * p.IW;
    } else {
      iy0[i] = -(1 << 28);
      ix0[i] = 0;
      pbase[i] = 0;
    }
  }
  size_t boff[B_INST];
  bool bok[B_INST];
#pragma unroll
  for (int i = 0; i < B_INST; ++i) {
    const int row = (wave * B_INST + i) * RPI + r_in;
    const int n = n0 + row;
    bok[i] = n < p.N;
    boff[i] = (size_t)(bok[i] ? n : 0) * p.ldb + (cphys ^ ((row / SWR) & (CPR - 1))) * VEC;
  }

  const int kpt = p.Cred / BK;
  const int nk = p.ntaps * kpt;

  // The tap table lives in lanes (tap t in lane t, ntaps <= 64) and is fetched with v_readlane: an ordinary
  // global load inside the loop would make the compiler drain the whole ring with vmcnt(0) every K tile.
  int tdy = 0, tdx = 0, twt = 0;
  if (lane < p.ntaps) {
    tdy = p.taps[3 * lane + 0];
    tdx = p.taps[3 * lane + 1];
    twt = p.taps[3 * lane + 2];
  }

  auto issue = [&](int t, int c0, int stage) {
    const int dy = __builtin_amdgcn_readlane(tdy, t);
    const int dx = __builtin_amdgcn_readlane(tdx, t);
    const size_t kB = (size_t)__builtin_amdgcn_readlane(twt, t) * p.Cred + c0;
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + BM * BKB;
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
      const int iy = iy0[i] + dy, ix = ix0[i] + dx;
      const bool ok = (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const size_t off = ((size_t)(pbase[i] + iy * p.IW + ix) * p.lda + c0 + acol[i]) * sizeof(T);
      const char* src = ok ? Ag + off : zero;
      __builtin_amdgcn_global_load_lds((gbl_void_ptr)src, (lds_void_ptr)(sa + (wave * A_INST + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
      const char* src = bok[i] ? Bg + (boff[i] + kB) * sizeof(T) : zero;
      __builtin_amdgcn_global_load_lds((gbl_void_ptr)src, (lds_void_ptr)(sb + (wave * B_INST + i) * 1024), 16, 0, 0);
    }
  };

  f32x16_t acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // (tap, channel offset) of the next tile to issue
  int it = 0, ic = 0, issued = 0;
  auto issue_next = [&](int stage) {
    issue(it, ic, stage);
    ++issued;
    ic += BK;
    if (ic == p.Cred) {
      ic = 0;
      ++it;
    }
  };
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) issue_next(s);

  // Fragment reads are inline asm: a compiler-visible ds_read after an LDS-DMA makes hipcc wait vmcnt(0).
  constexpr int KS = BKB / 32;
  const int l31 = lane & 31, lh = lane >> 5;
  const int fx = (l31 / SWR) & (CPR - 1);     // swizzle term of this lane's fragment rows (period 16 rows)
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
  uint32_t offA[KS], offB[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int coff = ((ks * 2 + lh) ^ fx) * 16;
    offA[ks] = lds0 + (wm * WTM + l31) * BKB + coff;
    offB[ks] = lds0 + BM * BKB + (wn * WTN + l31) * BKB + coff;
  }

  int cur = 0, nxt = S - 1;                   // stage of tile kt, stage of tile kt + S - 1
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = nk - 1 - kt;            // tiles issued after kt and still allowed in flight
    if (ahead >= S - 2) wait_vmcnt<(S - 2) * IPT>();
    else if (S > 3 && ahead == 1) wait_vmcnt<IPT>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (issued < nk) issue_next(nxt);
    const uint32_t sbase = cur * STAGE;
    ring_tile_mma<T, NI, MI, KS, BKB>(acc, offA, offB, sbase);
    cur = cur + 1 == S ? 0 : cur + 1;
    nxt = nxt + 1 == S ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();      // every wave is done reading the ring before the epilogue reuses it
  asm volatile("" ::: "memory");

  igemm_epilogue<TOut, BM, BN, false>(p, acc, smem, m0, n0, mt);
}

template <typename T, typename TOut, int BM, int BN, int BKB, int S, int MINB>
void launch_glds(const IgemmParams& p, hipStream_t stream) {
  constexpr int ring_bytes = S * (BM + BN) * BKB;
  constexpr int epi_bytes = BM * (BN * (int)sizeof(TOut) + 16);
  constexpr int red_bytes = 9 * BN * 4;
  constexpr int lds = ring_bytes > epi_bytes ? (ring_bytes > red_bytes ? ring_bytes : red_bytes)
                                             : (epi_bytes > red_bytes ? epi_bytes : red_bytes);
  auto kern = igemm_glds_kernel<T, TOut, BM, BN, BKB, S, MINB>;
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  const long tiles = (long)ceil_div(p.M, BM) * ceil_div(p.N, BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)p.groups), dim3(256), lds, stream, p);
}

template <typename T, typename TOut, int BM, int BN, int BKB, int MINB, bool ALIGNED, bool BNB = false, bool AXF = false,
          bool TAIL = false>
void launch_cfg(const IgemmParams& p, hipStream_t stream) {
  constexpr int LDR = BKB + 16;
  constexpr int stage_bytes = (BM + BN) * LDR;
  constexpr int epi_bytes = BM * (BN * (int)sizeof(TOut) + 16);
  constexpr int red_bytes = 9 * BN * 4;
  constexpr int axf_bytes = AXF ? 8 * kAxfMaxCred : 0;
  constexpr int max_lds = 2 * stage_bytes + axf_bytes > epi_bytes ? 2 * stage_bytes + axf_bytes : epi_bytes;
  // a single K tile (1x1 convolutions with few input channels) needs no second staging buffer
  constexpr int BK = BKB / (int)sizeof(T), VEC = 16 / (int)sizeof(T), CPR = BKB / 16;
  const int nk = ALIGNED ? p.ntaps * (TAIL ? (p.Cred + BK - 1) / BK : p.Cred / BK)
                         : (p.ntaps * (p.Cred / VEC) + CPR - 1) / CPR;
  int lds = (nk > 1 ? 2 : 1) * stage_bytes;
  if (AXF) {            // the (scale, shift) table sits behind BOTH staging buffers
    if (p.Cred > kAxfMaxCred) throw std::invalid_argument("igemm: apply-on-load supports at most 2048 reduction channels");
    lds = 2 * stage_bytes + 8 * p.Cred;
  }
  if (lds < epi_bytes) lds = epi_bytes;
  if (lds < red_bytes) lds = red_bytes;
  auto kern = igemm_kernel<T, TOut, BM, BN, BKB, MINB, ALIGNED, BNB, AXF, TAIL>;
  static bool attr_done = false;
  if (!attr_done && max_lds > 64 * 1024) {
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    attr_done = true;
  }
  const long tiles = (long)ceil_div(p.M, BM) * ceil_div(p.N, BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)p.groups), dim3(256), lds, stream, p);
}

// Tile configurations.  0..3 exist for every dtype pair; further experimental ones can be appended for
// 16-bit in == out (the training hot path) and timed per layer with `bench/conv_layers.py --sweep`
// (igemm_force_config); profiles/r1_igemm_sweep.md holds the sweep the heuristic below came from.
struct TileCfg { int bm, bn, bkb, minb; bool glds; };
constexpr TileCfg kCfgs[] = {
    {128, 128, 128, 2, false}, {128, 64, 128, 2, false}, {128, 128, 64, 2, false}, {128, 64, 64, 2, false},   // 0..3 (all dtypes)
    // LDS-DMA ring kernels (16-bit in == out): 4: 64B x3  5: 64B x4  6: 128B x3  7: narrow 64B x4
    // 8: 128B x2  9: narrow 128B x3 (tap table in lanes: <= 64 taps)
    {128, 128, 64, 3, true}, {128, 128, 64, 2, true}, {128, 128, 128, 1, true}, {128, 64, 64, 3, true},
    {128, 128, 128, 2, true}, {128, 64, 128, 2, true},
    // 10..13: register-staged 256x128 / 128x256 tiles (a wave owns 128x64 / 64x128): half the operand bytes per
    // MFMA through L2 and LDS; for the deep 3x3 layers, whose 128x128 tiles re-read the whole filter per tile
    {256, 128, 64, 2, false}, {128, 256, 64, 2, false}, {256, 128, 128, 1, false}, {128, 256, 128, 1, false},
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

int g_force_cfg = -1;
int g_row3_mode = -1;
thread_local bool t_xmask_done = false;      // the launched kernel applied kIgemmXMask itself
int g_nt_store = 0;
int g_tail_row3 = 1;      // filter-row kernel with masked tails (set through igemm_set_tail_min: negative = off)
int g_tail_min = 16;      // K-tile tails (TAIL kernels) from this many reduction channels per tap; 0: never (8: measured a loss)

// Pure function of the problem: the stats slab allocation (igemm_stats_rows) must agree with the launch.
// Measured (MI355X, batch 256 ResNet-50 shapes): 64-byte K tiles (40 KB LDS, 3-4 blocks/CU) win whenever
// there are enough tiles to fill the chip more than twice and the reduction is short -- those layers are
// bound by bytes in flight per CU; 128-byte K tiles (half the barriers) win for long reductions and for
// launches of <= 2 tiles per CU.
int choose_cfg(int M, int N, int Cred, int ntaps, int esize, bool ext) {
  const bool k128 = Cred % (128 / esize) == 0;
  if (g_force_cfg >= 0 && ext && g_force_cfg < kNumCfgs) {
    const TileCfg c = kCfgs[g_force_cfg];
    if (Cred % (c.bkb / esize) == 0 && (!c.glds || ntaps <= 64)) return g_force_cfg;
  }
  const bool wide = N > 64;
  const long tiles = (long)ceil_div(M, 128) * ceil_div(N, wide ? 128 : 64);
  const long kbytes = (long)ntaps * Cred * esize;
  const bool deep = k128 && (kbytes >= 4096 || tiles <= 512);
  if (deep) return wide ? 0 : 1;
  return wide ? 2 : 3;   // also the non-aligned gather (64-byte K tiles)
}

// The aligned kernels address the activation with 32-bit byte offsets from the first image of their tile: the images a
// tile of <= 256 (+2) output pixels can touch must span less than 4 GiB.  (The tensor itself may be any size.)
bool tile_span_fits32(long ohow, int IH, int IW, int lda, int es) {
  const double images = 258.0 / (double)(ohow > 0 ? ohow : 1) + 2.0;
  return images * IH * IW * lda * es < 4294967296.0;
}

template <typename T, typename TOut>
void dispatch_tiles(const IgemmParams& p, hipStream_t stream) {
  constexpr int ES = (int)sizeof(T);
  constexpr bool EXT = sizeof(T) == 2 && std::is_same<T, TOut>::value;   // tuning hook applies here only
  // the aligned kernels address both operands with 32-bit byte offsets
  const bool fits32 = tile_span_fits32(p.div_ohow.d, p.IH, p.IW, p.lda, ES) && (double)p.N * p.ldb * ES < 4294967296.0;
  const bool aligned = p.Cred % (64 / ES) == 0 && fits32;
  const int cfg = choose_cfg(p.M, p.N, p.Cred, p.ntaps, ES, EXT);
  if (p.flags & kIgemmAxf) {
    // forward convolution whose input is a BatchNorm + ReLU applied while the A tile is staged
    if constexpr (EXT) {
      if (aligned && !(p.flags & kIgemmBnBwd) && p.groups == 1) {
        if ((p.flags & kIgemmRow3) && g_row3_mode != 0 && p.ntaps % 3 == 0 && p.SX == 1 && (int)p.div_ow.d == p.IW) {
          const bool k128 = p.Cred % (128 / ES) == 0 && (g_row3_mode == 2 || (g_row3_mode < 0 && p.Cred * ES >= 1024));
          if (p.N > 64) {
            if (k128) launch_row3<T, TOut, 128, 128, 128, 2, false, true>(p, stream);
            else launch_row3<T, TOut, 128, 128, 64, 3, false, true>(p, stream);
          } else {
            if (k128) launch_row3<T, TOut, 128, 64, 128, 3, false, true>(p, stream);
            else launch_row3<T, TOut, 128, 64, 64, 4, false, true>(p, stream);
          }
          return;
        }
        // every AXF kernel has 128 pixel rows per tile, which is what igemm_stats_rows promised for cfg 0..9
        if (kCfgs[cfg].bm != 128) throw std::invalid_argument("igemm: apply-on-load with a forced 256-row tile");
        switch (cfg <= 3 ? cfg : choose_cfg(p.M, p.N, p.Cred, p.ntaps, ES, false)) {
          case 0: launch_cfg<T, TOut, 128, 128, 128, 2, true, false, true>(p, stream); return;
          case 1: launch_cfg<T, TOut, 128, 64, 128, 2, true, false, true>(p, stream); return;
          case 2: launch_cfg<T, TOut, 128, 128, 64, 2, true, false, true>(p, stream); return;
          default: launch_cfg<T, TOut, 128, 64, 64, 2, true, false, true>(p, stream); return;
        }
      }
    }
    throw std::invalid_argument("igemm: apply-on-load needs a dense 16-bit problem with 64-byte aligned channels");
  }
  if constexpr (EXT) {
    if ((p.flags & kIgemmRow3) && aligned && g_force_cfg < 0 && g_row3_mode != 0 &&
        p.ntaps % 3 == 0 && p.SX == 1 && (int)p.div_ow.d == p.IW) {
      if (p.flags & kIgemmBnBwd) {           // data-gradient launch that also reduces the producing BN's backward sums
        if (p.N > 64) launch_row3<T, TOut, 128, 128, 64, 2, true>(p, stream);
        else launch_row3<T, TOut, 128, 64, 64, 3, true>(p, stream);
        return;
      }
      // measured (scratch/row3_check.py, batch 1024): 64-byte K tiles win up to 256 channels, 128-byte from 512
      const bool k128 = p.Cred % (128 / ES) == 0 && (g_row3_mode == 2 || (g_row3_mode < 0 && p.Cred * ES >= 1024));
      if (p.flags & kIgemmXMask) {           // data gradient stored through the ReLU mask of the tensor it belongs to
        if (p.N > 64) {
          if (k128) launch_row3<T, TOut, 128, 128, 128, 2, false, false, false, true>(p, stream);
          else launch_row3<T, TOut, 128, 128, 64, 3, false, false, false, true>(p, stream);
        } else {
          if (k128) launch_row3<T, TOut, 128, 64, 128, 3, false, false, false, true>(p, stream);
          else launch_row3<T, TOut, 128, 64, 64, 4, false, false, false, true>(p, stream);
        }
        t_xmask_done = true;
        return;
      }
      if (p.N > 64) {
        if (k128) launch_row3<T, TOut, 128, 128, 128, 2>(p, stream);
        else launch_row3<T, TOut, 128, 128, 64, 3>(p, stream);
      } else {
        if (k128) launch_row3<T, TOut, 128, 64, 128, 3>(p, stream);
        else launch_row3<T, TOut, 128, 64, 64, 4>(p, stream);
      }
      return;
    }
  }
  if (!aligned) {
    if (p.flags & kIgemmBnBwd)
      throw std::invalid_argument("igemm: fused BN-backward reduction needs a 16-bit, 64-byte aligned problem");
    if constexpr (EXT) {
      // 16-byte but not 64-byte aligned reduction (RegNet group widths): per-tap K tiles with a masked tail keep
      // the per-tap 32-bit addressing of the aligned kernels instead of the per-chunk 64-bit gather below
      if (fits32 && g_force_cfg < 0 && g_tail_min > 0 && p.Cred >= g_tail_min) {
        if ((p.flags & kIgemmRow3) && g_row3_mode != 0 && g_tail_row3 && p.ntaps % 3 == 0 && p.SX == 1 &&
            (int)p.div_ow.d == p.IW) {
          if (p.N > 64) launch_row3<T, TOut, 128, 128, 64, 3, false, false, true>(p, stream);
          else launch_row3<T, TOut, 128, 64, 64, 4, false, false, true>(p, stream);
          return;
        }
        if (p.N > 64) launch_cfg<T, TOut, 128, 128, 64, 2, true, false, false, true>(p, stream);
        else launch_cfg<T, TOut, 128, 64, 64, 2, true, false, false, true>(p, stream);
        return;
      }
    }
    if (p.N > 64) launch_cfg<T, TOut, 128, 128, 64, 2, false>(p, stream);
    else launch_cfg<T, TOut, 128, 64, 64, 2, false>(p, stream);
    return;
  }
  if (p.flags & kIgemmBnBwd) {
    if constexpr (EXT) {
      switch (cfg) {
        case 0: launch_cfg<T, TOut, 128, 128, 128, 2, true, true>(p, stream); return;
        case 1: launch_cfg<T, TOut, 128, 64, 128, 2, true, true>(p, stream); return;
        case 2: launch_cfg<T, TOut, 128, 128, 64, 2, true, true>(p, stream); return;
        case 3: launch_cfg<T, TOut, 128, 64, 64, 2, true, true>(p, stream); return;
        default: break;
      }
    }
    throw std::invalid_argument("igemm: fused BN-backward reduction needs a 16-bit, 64-byte aligned problem");
  }
  switch (cfg) {
    case 0: launch_cfg<T, TOut, 128, 128, 128, 2, true>(p, stream); return;
    case 1: launch_cfg<T, TOut, 128, 64, 128, 2, true>(p, stream); return;
    case 2: launch_cfg<T, TOut, 128, 128, 64, 2, true>(p, stream); return;
    case 3: launch_cfg<T, TOut, 128, 64, 64, 2, true>(p, stream); return;
    default: break;
  }
  if constexpr (EXT) {
    switch (cfg) {
      case 4: launch_glds<T, TOut, 128, 128, 64, 3, 3>(p, stream); return;
      case 5: launch_glds<T, TOut, 128, 128, 64, 4, 2>(p, stream); return;
      case 6: launch_glds<T, TOut, 128, 128, 128, 3, 1>(p, stream); return;
      case 7: launch_glds<T, TOut, 128, 64, 64, 4, 3>(p, stream); return;
      case 8: launch_glds<T, TOut, 128, 128, 128, 2, 2>(p, stream); return;
      case 9: launch_glds<T, TOut, 128, 64, 128, 3, 2>(p, stream); return;
      case 10: launch_cfg<T, TOut, 256, 128, 64, 2, true>(p, stream); return;
      case 11: launch_cfg<T, TOut, 128, 256, 64, 2, true>(p, stream); return;
      case 12: launch_cfg<T, TOut, 256, 128, 128, 1, true>(p, stream); return;
      case 13: launch_cfg<T, TOut, 128, 256, 128, 1, true>(p, stream); return;
      default: break;
    }
  }
  throw std::logic_error("igemm: bad tile configuration");
}

}  // namespace

void igemm_launch(const IgemmParams& p_in, int dtype, int out_dtype, hipStream_t stream) {
  IgemmParams p = p_in;
  const int vec = dtype == kF32 ? 4 : 8;
  if (p.M <= 0 || p.N <= 0) return;
  if (p.groups <= 1) {
    p.groups = 1;
    p.a_gstride = p.b_gstride = p.c_gstride = 0;
  } else {
    const int veco_g = out_dtype == kF32 ? 4 : 8;
    if (p.a_gstride % vec || p.b_gstride % vec || p.c_gstride % veco_g || p.groups > 65535)
      throw std::invalid_argument("igemm: group slices must start on 16-byte boundaries");
    if ((p.flags & kIgemmBnBwd) && (p.c_gstride != p.N || p.stats_ld != p.ldc || (long)p.groups * p.N != p.ldc))
      throw std::invalid_argument("igemm: grouped BN-backward reduction needs the groups to tile the channel rows");
  }
  if (p.stats_ld <= 0) p.stats_ld = p.N;
  {
    const long mb = nt_min_bytes();
    if (g_nt_store > 0 || (g_nt_store == 0 && mb >= 0 && (long)p.M * p.N * p.groups * (out_dtype == kF32 ? 4 : 2) >= mb))
      p.flags |= kIgemmNtStore;
  }
  if (p.Cred % vec != 0 || p.lda % vec != 0 || p.ldb % vec != 0)
    throw std::invalid_argument("igemm: reduction channels / strides must be multiples of 16 bytes");
  const int veco = out_dtype == kF32 ? 4 : 8;
  if (p.ldc % veco != 0) throw std::invalid_argument("igemm: ldc must be a multiple of 16 bytes");
  if ((long)p.M >= (1l << 31) || p.ntaps <= 0) throw std::invalid_argument("igemm: bad problem size");
  if (p.stats != nullptr && out_dtype != dtype)
    throw std::invalid_argument("igemm: stats need out dtype == in dtype");
  if ((p.flags & kIgemmAxf) && (p.a_scale == nullptr || p.a_shift == nullptr))
    throw std::invalid_argument("igemm: apply-on-load needs a_scale and a_shift");
  if (p.flags & kIgemmBnBwd) {
    if (p.stats == nullptr || p.bn_x == nullptr || p.bn_coef == nullptr ||
        ((p.bn_relu == 1 || p.bn_relu == 3) && p.bn_z == nullptr))
      throw std::invalid_argument("igemm: fused BN-backward reduction needs stats slab, bn_x, bn_coef (and bn_z)");
    if (p.N % veco != 0 || p.ldc != p.stats_ld)
      throw std::invalid_argument("igemm: fused BN-backward reduction needs dense rows of whole 16-byte chunks");
  }
  if ((p.flags & kIgemmBnFold) && (!(p.flags & kIgemmBnBwd) || !(p.bn_relu == 0 || p.bn_relu == 3)))
    throw std::invalid_argument("igemm: kIgemmBnFold needs the fused BN-backward reduction with no ReLU or a bit mask");
  if (p.flags & kIgemmXMask) {
    if (p.bn_x == nullptr || (p.flags & (kIgemmBnBwd | kIgemmScatter | kIgemmAxf)) || p.groups != 1 || p.ldc != p.N ||
        p.N % veco != 0 || dtype != out_dtype)
      throw std::invalid_argument("igemm: masked store needs a dense, unscattered launch with the mask source in bn_x");
    p.bn_relu = 0;
    p.bn_z = nullptr;
  }
  t_xmask_done = false;
  const bool lean_only = p.A2 != nullptr || (p.flags & kIgemmReluMask);
  if (g_force_cfg < 0 && !lean_only && conv8_try(p, dtype, out_dtype, stream)) return;
  if (!(p.flags & kIgemmXMask) && (p.flags & kIgemmDense1) && (g_force_cfg < 0 || lean_only) &&
      gemm_lean_try(p, dtype, out_dtype, stream)) {
    DPA_CHECK_HIP(hipGetLastError());
    return;
  }
  if (p.flags & kIgemmReluMask)
    throw std::invalid_argument("igemm: relu(acc + bias + R) with a bit-mask output is only implemented by the lean GEMM (dense 1x1, 16-bit)");
  if (p.flags & kIgemmRStrided)
    throw std::invalid_argument("igemm: a compact strided accumulate is only implemented by the lean GEMM");
  if (p.A2 != nullptr)
    throw std::invalid_argument("igemm: a second gathered operand is only implemented by the lean GEMM (dense 1x1, 16-bit)");
  if (dtype == kF32 && out_dtype == kF32) dispatch_tiles<float, float>(p, stream);
  else if (dtype == kBF16 && out_dtype == kBF16) dispatch_tiles<bf16_t, bf16_t>(p, stream);
  else if (dtype == kBF16 && out_dtype == kF32) dispatch_tiles<bf16_t, float>(p, stream);
  else if (dtype == kF16 && out_dtype == kF16) dispatch_tiles<f16_t, f16_t>(p, stream);
  else if (dtype == kF16 && out_dtype == kF32) dispatch_tiles<f16_t, float>(p, stream);
  else throw std::invalid_argument("igemm: unsupported dtype combination");
  DPA_CHECK_HIP(hipGetLastError());
  if ((p.flags & kIgemmXMask) && !t_xmask_done)      // no masking kernel for this shape: one in-place elementwise pass
    relu_bwd_launch(p.bn_x, p.C, p.C, (long)p.M * p.N, out_dtype, stream);
}

bool igemm_axf_ok(long ohow, int IH, int IW, int lda, int N, int Cred, int ldb, int dtype, int out_dtype, int groups) {
  if (dtype == kF32 || dtype != out_dtype || groups > 1) return false;
  const int es = 2;
  if (Cred % (64 / es) != 0 || Cred > kAxfMaxCred) return false;
  return tile_span_fits32(ohow, IH, IW, lda, es) && (double)N * ldb * es < 4294967296.0;
}

int igemm_stats_rows(int M, int N, int Cred, int ntaps, int dtype, int out_dtype) {
  const int es = dtype == kF32 ? 4 : 2;
  const int cfg = choose_cfg(M, N, Cred, ntaps, es, es == 2 && dtype == out_dtype);
  return ceil_div(M, kCfgs[cfg].bm);
}

void igemm_force_config(int cfg) { g_force_cfg = cfg; }
void igemm_set_row3(int mode) { g_row3_mode = mode; }
void igemm_set_tail_min(int cred) {       // negative: |cred| with the filter-row tail kernels off (A/B runs)
  g_tail_row3 = cred >= 0;
  g_tail_min = cred >= 0 ? cred : -cred;
}
void igemm_set_nt_store(int mode) { g_nt_store = mode; }   // 0: by size (nt_min_bytes), 1: always, -1: never
int igemm_num_configs() { return kNumCfgs; }

int igemm_tile_m(int /*dtype*/) { return 128; }

}  // namespace dpa
