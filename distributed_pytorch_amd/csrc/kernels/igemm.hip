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
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

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

// BM: pixel rows per block, BN: output channels per block, BKB: bytes of K per row per K tile.
// 256 threads = 2 (pixel) x 2 (channel) waves.
template <typename T, typename TOut, int BM, int BN, int BKB, bool ALIGNED>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int BK = BKB / (int)sizeof(T);
  constexpr int CPR = BKB / 16;          // 16-byte chunks per tile row
  constexpr int LDR = BKB + 16;          // padded LDS row (odd multiple of 16 B: conflict-free b128)
  constexpr int A_IT = BM * CPR / 256;
  constexpr int B_IT = BN * CPR / 256;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int STAGE = (BM + BN) * LDR;
  constexpr int LDE = BN * (int)sizeof(TOut) + 16;   // epilogue tile row bytes
  constexpr int CE = BN * (int)sizeof(TOut) / 16;    // 16-byte chunks per epilogue row
  constexpr int E_IT = BM * CE / 256;
  constexpr int VECO = 16 / (int)sizeof(TOut);
  static_assert(A_IT >= 1 && B_IT >= 1 && MI >= 1 && NI >= 1 && E_IT >= 1, "tile too small");
  static_assert(256 % CPR == 0 && 256 % CE == 0, "chunk mapping");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const uint32_t tiles_n = (p.N + BN - 1) / BN;
  const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / tiles_n, nt = lid - mt * tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  const char* __restrict__ Ag = reinterpret_cast<const char*>(p.A);
  const char* __restrict__ Bg = reinterpret_cast<const char*>(p.B);

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
      pbase[i] = n * p.IH * p.IW;
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
  const int kpt = ALIGNED ? p.Cred / BK : 1;           // K tiles per tap
  const int cpt = p.Cred / VEC;                        // chunks per tap
  const int nk = ALIGNED ? p.ntaps * kpt : (p.ntaps * cpt + CPR - 1) / CPR;

  auto load_tile = [&](int kt) {
    int dy, dx, cA;
    size_t kB;
    bool tok = true;
    if (ALIGNED) {
      const int t = kt / kpt;
      const int c0 = (kt - t * kpt) * BK;
      dy = p.taps[3 * t + 0];
      dx = p.taps[3 * t + 1];
      cA = c0 + ch * VEC;
      kB = (size_t)p.taps[3 * t + 2] * p.Cred + cA;
    } else {
      const int kc = kt * CPR + ch;
      int t = kc / cpt;
      cA = (kc - t * cpt) * VEC;
      tok = t < p.ntaps;
      t = tok ? t : 0;
      dy = p.taps[3 * t + 0];
      dx = p.taps[3 * t + 1];
      kB = (size_t)p.taps[3 * t + 2] * p.Cred + cA;
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int iy = iy0[i] + dy, ix = ix0[i] + dx;
      const bool ok = tok && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const size_t off = ((size_t)(pbase[i] + iy * p.IW + ix) * p.lda + cA) * sizeof(T);
      ra[i] = ok ? *reinterpret_cast<const uint4*>(Ag + off) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const bool ok = tok && bok[i];
      rb[i] = ok ? *reinterpret_cast<const uint4*>(Bg + (brow[i] + kB) * sizeof(T))
                 : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* sa = smem + buf * STAGE;
    unsigned char* sb = sa + BM * LDR;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = (tid + i * 256) / CPR;
      *reinterpret_cast<uint4*>(sa + row * LDR + ch * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = (tid + i * 256) / CPR;
      *reinterpret_cast<uint4*>(sb + row * LDR + ch * 16) = rb[i];
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

  // ---- epilogue: registers -> LDS tile [pixel][channel] -> 16-byte global stores -----------
  // acc[a][b][r]: channel = wn*WTN + a*32 + (r&3) + 8*(r>>2) + 4*lh, pixel = wm*WTM + b*32 + l31
  unsigned char* se = smem;
  const bool relu = p.flags & kIgemmRelu;
#pragma unroll
  for (int a = 0; a < NI; ++a) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cch = wn * WTN + a * 32 + 8 * g + 4 * lh;
      float bs[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bs[j] = (n0 + cch + j < p.N) ? p.bias[n0 + cch + j] : 0.f;
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
  __syncthreads();

  const int ce = tid % CE;
  const int ncol = n0 + ce * VECO;
  float s1[VECO], s2[VECO];
#pragma unroll
  for (int j = 0; j < VECO; ++j) s1[j] = s2[j] = 0.f;
  char* __restrict__ Cg = reinterpret_cast<char*>(p.C);
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
      const char* rsrc = reinterpret_cast<const char*>(p.R) + goff;
      if (ncol + VECO <= p.N) {
        Vec16<TOut> r;
        r.load(rsrc);
#pragma unroll
        for (int j = 0; j < VECO; ++j) v.set(j, v.get(j) + r.get(j));
      } else {
        for (int j = 0; j < VECO && ncol + j < p.N; ++j) {
          TOut t;
          __builtin_memcpy(&t, rsrc + j * sizeof(TOut), sizeof(TOut));
          v.set(j, v.get(j) + to_float(t));
        }
      }
    }
    if (ncol + VECO <= p.N) {
      v.store(dst);
    } else {
      for (int j = 0; j < VECO && ncol + j < p.N; ++j)
        __builtin_memcpy(dst + j * sizeof(TOut), reinterpret_cast<const char*>(&v.raw) + j * sizeof(TOut),
                         sizeof(TOut));
    }
    if (p.stats != nullptr) {
#pragma unroll
      for (int j = 0; j < VECO; ++j) {
        const float f = v.get(j);
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
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][BN]
    if (lane < CE) {
#pragma unroll
      for (int j = 0; j < VECO; ++j) {
        red[(wave * 2 + 0) * BN + ce * VECO + j] = s1[j];
        red[(wave * 2 + 1) * BN + ce * VECO + j] = s2[j];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int c = tid; c < 2 * BN; c += 256) {
      const int which = c / BN, col = c - which * BN;
      if (n0 + col < p.N) {
        const float s = red[(0 * 2 + which) * BN + col] + red[(1 * 2 + which) * BN + col] +
                        red[(2 * 2 + which) * BN + col] + red[(3 * 2 + which) * BN + col];
        p.stats[((size_t)mt * 2 + which) * p.N + n0 + col] = s;
      }
    }
  }
}

template <typename T, typename TOut, int BM, int BN, int BKB, bool ALIGNED>
void launch_cfg(const IgemmParams& p, hipStream_t stream) {
  constexpr int LDR = BKB + 16;
  constexpr int stage_bytes = (BM + BN) * LDR;
  constexpr int epi_bytes = BM * (BN * (int)sizeof(TOut) + 16);
  constexpr int red_bytes = 4 * 2 * BN * 4;
  constexpr int max_lds = 2 * stage_bytes > epi_bytes ? 2 * stage_bytes : epi_bytes;
  // a single K tile (1x1 convolutions with few input channels) needs no second staging buffer
  constexpr int BK = BKB / (int)sizeof(T), VEC = 16 / (int)sizeof(T), CPR = BKB / 16;
  const int nk = ALIGNED ? p.ntaps * (p.Cred / BK) : (p.ntaps * (p.Cred / VEC) + CPR - 1) / CPR;
  int lds = (nk > 1 ? 2 : 1) * stage_bytes;
  if (lds < epi_bytes) lds = epi_bytes;
  if (lds < red_bytes) lds = red_bytes;
  auto kern = igemm_kernel<T, TOut, BM, BN, BKB, ALIGNED>;
  static bool attr_done = false;
  if (!attr_done && max_lds > 64 * 1024) {
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    attr_done = true;
  }
  const long tiles = (long)ceil_div(p.M, BM) * ceil_div(p.N, BN);
  if (getenv("DPA_DEBUG")) { fprintf(stderr, "  launch tiles=%ld lds=%d nk=%d\n", tiles, lds, nk); fflush(stderr); }
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  if (getenv("DPA_DEBUG")) { fprintf(stderr, "  launched\n"); fflush(stderr); }
}

template <typename T, typename TOut>
void dispatch_tiles(const IgemmParams& p, hipStream_t stream) {
  constexpr int BK64 = 64 / (int)sizeof(T), BK128 = 128 / (int)sizeof(T);
  const bool wide = p.N > 64;
  if (p.Cred % BK128 == 0) {
    if (wide) launch_cfg<T, TOut, 128, 128, 128, true>(p, stream);
    else launch_cfg<T, TOut, 128, 64, 128, true>(p, stream);
  } else if (p.Cred % BK64 == 0) {
    if (wide) launch_cfg<T, TOut, 128, 128, 64, true>(p, stream);
    else launch_cfg<T, TOut, 128, 64, 64, true>(p, stream);
  } else {
    if (wide) launch_cfg<T, TOut, 128, 128, 64, false>(p, stream);
    else launch_cfg<T, TOut, 128, 64, 64, false>(p, stream);
  }
}

}  // namespace

void igemm_launch(const IgemmParams& p, int dtype, int out_dtype, hipStream_t stream) {
  if (getenv("DPA_DEBUG")) {
    fprintf(stderr, "igemm M=%d N=%d Cred=%d ntaps=%d lda=%d ldb=%d ldc=%d IH=%d IW=%d S=%d,%d flags=%d dt=%d/%d\n", p.M, p.N,
            p.Cred, p.ntaps, p.lda, p.ldb, p.ldc, p.IH, p.IW, p.SY, p.SX, p.flags, dtype, out_dtype);
    fflush(stderr);
  }
  const int vec = dtype == kF32 ? 4 : 8;
  if (p.M <= 0 || p.N <= 0) return;
  if (p.Cred % vec != 0 || p.lda % vec != 0 || p.ldb % vec != 0)
    throw std::invalid_argument("igemm: reduction channels / strides must be multiples of 16 bytes");
  const int veco = out_dtype == kF32 ? 4 : 8;
  if (p.ldc % veco != 0) throw std::invalid_argument("igemm: ldc must be a multiple of 16 bytes");
  if ((long)p.M >= (1l << 31) || p.ntaps <= 0) throw std::invalid_argument("igemm: bad problem size");
  if (p.stats != nullptr && out_dtype != dtype)
    throw std::invalid_argument("igemm: stats need out dtype == in dtype");
  if (dtype == kF32 && out_dtype == kF32) dispatch_tiles<float, float>(p, stream);
  else if (dtype == kBF16 && out_dtype == kBF16) dispatch_tiles<bf16_t, bf16_t>(p, stream);
  else if (dtype == kBF16 && out_dtype == kF32) dispatch_tiles<bf16_t, float>(p, stream);
  else if (dtype == kF16 && out_dtype == kF16) dispatch_tiles<f16_t, f16_t>(p, stream);
  else if (dtype == kF16 && out_dtype == kF32) dispatch_tiles<f16_t, float>(p, stream);
  else throw std::invalid_argument("igemm: unsupported dtype combination");
  DPA_CHECK_HIP(hipGetLastError());
}

int igemm_tile_m(int /*dtype*/) { return 128; }

}  // namespace dpa
