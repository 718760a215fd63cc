// Weight gradient of wide layers on the 256 x 256 multi-phase pipeline (gemm8.hip's structure); dense 1x1 problems:
//
//     dW[n][c] = sum_m dY[m][n] * X[m][c]          (rows N .. N + N2 - 1: dY2^T X, the Gram matrix of the BN fold)
//
// wgrad.hip stages both operands through registers into a two-buffer LDS tile with one barrier per K tile and plateaus at
// 0.6-0.7 PFLOP/s on the deep layers (14x14 / 7x7, K = 50-200k pixels), whatever the tile shape or K depth
// (profiles/r3/wgrad_cfg_sweep_k64.txt).  Here:
//
//  * 8 waves (2 along the dY channels x 4 along the X channels), one workgroup per CU, a wave owns 128 x 64 of the tile;
//  * operands go global -> LDS with global_load_lds_dwordx4 in HALF-tiles of 64 pixels x 128 channels (16 KiB, exactly
//    as they sit in HBM: pixel-major), ring of 8 half-tile slots = two K tiles;
//  * the MFMA fragments are fetched with the transposed LDS read (ds_read_b64_tr_b16, two per 32x32x16 operand);
//  * phases, barriers, the stagger of the two wave groups and the counted vmcnt are gemm8's: see the ordering argument
//    there (half-tile order inside a K tile: X cols 0..127, X cols 128..255, dY cols 0..127, dY cols 128..255; the X
//    fragments are read in phases 0 and 1, the dY fragments in phases 0 and 2).
//
// LDS layout of a half-tile: [pixel 0..63][256 bytes], the 64-byte quarter q (32 channels) of pixel row k stored at
// quarter q ^ (k & 3): the source address is permuted, the DMA writes linearly.  A transposed read touches, per half
// wave, 4 consecutive pixel rows x 64 bytes of ONE logical quarter -> 4 distinct physical quarters = all 64 banks once.
//
// Convolutions with taps (3x3, strided 1x1, ...): dW[n][t*Cred + c] = sum_m dY[m][n] * X[pix(m) + tap_t][c], template
// parameter GATHER below.
//
// The pixel range is split over workgroups as in wgrad.hip: fp32 partial slabs [split][N + N2][C], folded in a fixed
// order by slab_reduce_kernel.  Pixels past the end of a split and channels past N / C read a zero page.
//
// Reference: the dW GEMMs of loss.backward() (nd_imagenet.py:302, imagenet_ddp_apex.py:381) -- SURVEY.md N-K3.
#include "../common.h"
#include "kernels.h"
#include <stdexcept>

namespace dpa {
namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr_w8;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr_w8;

__device__ uint4 g_wgrad8_zero[4];     // never written: zero-initialised

constexpr int kHalfW = 64 * 256;       // bytes of a half-tile: 64 pixels x 128 channels x 2 B
constexpr int kRingW = 8 * kHalfW;     // 128 KiB

// two transposed 8-byte reads = the 8 k values of one 32x32x16 operand lane (rows k .. k+3 and k+4 .. k+7)
template <int OFF> __device__ __forceinline__ uint4 lds_tr_frag(uint32_t addr) {
  uint2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 1024));
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

template <typename T> struct MmaW8;
template <> struct MmaW8<bf16_t> {
  static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    bf16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
};
template <> struct MmaW8<f16_t> {
  static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    f16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  }
};

// GATHER: X is an NHWC image and column j = tap * Cred + c of the gradient reads pixel (oy * SY + dy_tap, ox * SX + dx_tap)
// (Cred a multiple of 128: a 128-column half-tile lies inside one tap).  The (image, row, column) of the two pixel rows a
// lane stages advance incrementally from K tile to K tile; out-of-image taps read the zero page.
template <typename T, bool GATHER>
__global__ __launch_bounds__(512, 1) void wgrad8_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, lh = lane >> 5;

  const uint32_t tiles_i1 = (p.N + 255) / 256;
  const uint32_t tiles_i = tiles_i1 + (p.N2 + 255) / 256, tiles_j = (p.Ktot + 255) / 256;
  const uint32_t tiles = tiles_i * tiles_j;
  const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t split = lid / tiles, tile = lid - split * tiles;
  const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
  const bool second = (uint32_t)ti >= tiles_i1;
  const int i0 = (second ? ti - (int)tiles_i1 : ti) * 256, j0 = tj * 256;
  const int Nloc = second ? p.N2 : p.N, ldyl = second ? p.ldy2 : p.ldy;
  const int row_base = second ? p.N : 0;

  const int m_begin = split * p.pix_per_split;
  const int len = min(p.M, m_begin + p.pix_per_split) - m_begin;      // > 0 (host checked)
  const int nk = (len + 63) / 64;
  const int nhalf = 4 * nk;

  // ---- staging: wave-instruction i (0, 1) of this wave covers pixels (wave * 2 + i) * 4 + lane / 16 of a half-tile;
  // the lane's LDS slot is chunk lane % 16 of that pixel row and holds the source chunk with its quarter XORed
  const char* zero = reinterpret_cast<const char*>(g_wgrad8_zero);
  const char* src[4][2];          // [half-tile kind: X0, X1, Y0, Y1][instruction]
  int step[4][2];                 // bytes per K tile (0 for chunks that read the zero page)
  int pixl[2];                    // pixel row of this lane inside a K tile, per instruction
#pragma unroll
  for (int i = 0; i < 2; ++i) pixl[i] = (wave * 2 + i) * 4 + (lane >> 4);
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pix = pixl[i];
      const int pc = lane & 15;
      const int lc = ((((pc >> 2) ^ (pix & 3))) << 2) | (pc & 3);
      const bool isY = h >= 2;
      const int col = (isY ? i0 : j0) + (h & 1) * 128 + lc * 8;
      const bool ok = col < (isY ? Nloc : p.Ktot);
      const char* base = reinterpret_cast<const char*>(isY ? (second ? p.dY2 : p.dY) : p.X);
      const long ld = isY ? ldyl : p.ldx;
      if (GATHER && !isY) {         // per-tap addressing below
        src[h][i] = zero;
        step[h][i] = 0;
        continue;
      }
      src[h][i] = ok ? base + (((long)m_begin + pix) * ld + col) * (long)sizeof(T) : zero;
      step[h][i] = ok ? (int)(64 * ld * (long)sizeof(T)) : 0;
    }
  // GATHER state: the tap of each X half-tile (workgroup-uniform), the lane's column inside it, and the coordinates of
  // the lane's two pixel rows in the K tile that X0 stages next (X1 of the same K tile follows and advances them)
  int tap_dy[2] = {0, 0}, tap_dx[2] = {0, 0}, colb[2] = {0, 0};
  bool colok[2] = {false, false};
  int gn[2] = {0, 0}, goy[2] = {0, 0}, gox[2] = {0, 0};
  int OHg = 1, OWg = 1, step_n = 0, step_oy = 0, step_ox = 0;
  if constexpr (GATHER) {
    const int pc = lane & 15;
    const int lc = ((((pc >> 2) ^ (lane >> 4))) << 2) | (pc & 3);     // pix & 3 == lane >> 4 for both instructions
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int col0 = j0 + h * 128;
      const int t = min(col0, p.Ktot - 1) / p.Cred;
      tap_dy[h] = p.taps[3 * t + 0];
      tap_dx[h] = p.taps[3 * t + 1];
      colb[h] = (col0 - t * p.Cred + lc * 8) * (int)sizeof(T);
      colok[h] = col0 + lc * 8 < p.Ktot;
    }
    OWg = (int)p.div_ow.d;
    OHg = (int)(p.div_ohow.d / p.div_ow.d);
    step_n = 64 / (OHg * OWg);
    const int rem = 64 - step_n * OHg * OWg;
    step_oy = rem / OWg;
    step_ox = rem - step_oy * OWg;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t m = (uint32_t)(m_begin + pixl[i]);
      const uint32_t n = fdiv(m, p.div_ohow);
      const uint32_t r = m - n * p.div_ohow.d;
      const uint32_t oy = fdiv(r, p.div_ow);
      gn[i] = (int)n; goy[i] = (int)oy; gox[i] = (int)(r - oy * p.div_ow.d);
    }
  }
  const long ldxb = (long)p.ldx * (long)sizeof(T);
  // request half-tile g (kind g & 3 of K tile g >> 2) into slot g & 7
  auto stage = [&](int g, auto kind_c) {
    constexpr int h = decltype(kind_c)::value;
    unsigned char* dst = smem + (g & 7) * kHalfW + wave * 2048;
    const int lim = len - (g >> 2) * 64;            // pixel rows of that K tile inside the split
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const char* s;
      if constexpr (GATHER && h < 2) {
        const int iy = goy[i] * p.SY + tap_dy[h], ix = gox[i] * p.SX + tap_dx[h];
        const bool ok = pixl[i] < lim && colok[h] && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        const int pixoff = (gn[i] * p.IH + iy) * p.IW + ix;
        s = ok ? reinterpret_cast<const char*>(p.X) + (long)pixoff * ldxb + colb[h] : zero;
        if constexpr (h == 1) {     // both X half-tiles of this K tile are requested: on to the next 64 pixels
          int ox = gox[i] + step_ox, oy = goy[i] + step_oy, n = gn[i] + step_n;
          if (ox >= OWg) { ox -= OWg; ++oy; }
          if (oy >= OHg) { oy -= OHg; ++n; }
          gox[i] = ox; goy[i] = oy; gn[i] = n;
        }
      } else {
        s = pixl[i] < lim ? src[h][i] : zero;
        src[h][i] += step[h][i];
      }
      __builtin_amdgcn_global_load_lds((gbl_void_ptr_w8)s, (lds_void_ptr_w8)(dst + i * 1024), 16, 0, 0);
    }
  };
  constexpr std::integral_constant<int, 0> kX0{};
  constexpr std::integral_constant<int, 1> kX1{};
  constexpr std::integral_constant<int, 2> kY0{};
  constexpr std::integral_constant<int, 3> kY1{};

  // ---- fragment addresses (bytes from the start of the ring, K-tile parity added per tile).  Lane (g, gi) of a
  // transposed read addresses pixel row 8 * (g >> 1) + (gi >> 2) (+ 4 for the upper half, + 16 per MFMA step: immediates)
  // and channels 16 * (g & 1) + 4 * (gi & 3) of a 32-channel block; the row's swizzle term is gi >> 2.
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
  const int g4 = lane >> 4, gi = lane & 15;
  const uint32_t lane_off = (uint32_t)((8 * (g4 >> 1) + (gi >> 2)) * 256 + (16 * (g4 & 1) + 4 * (gi & 3)) * 2);
  const uint32_t sw = (uint32_t)(gi >> 2);
  uint32_t cY[4], cX[2];
#pragma unroll
  for (int b = 0; b < 4; ++b) cY[b] = lds0 + (2 + wr) * kHalfW + lane_off + (((uint32_t)b ^ sw) * 64);
#pragma unroll
  for (int jq = 0; jq < 2; ++jq)
    cX[jq] = lds0 + (wc >> 1) * kHalfW + lane_off + ((((uint32_t)((wc & 1) * 2 + jq)) ^ sw) * 64);

  f32x16_t acc[2][2][2];                  // [iq][it][jq]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;

  // ---- prologue: half-tiles 0..4, K tile 0 complete (one half-tile may stay in flight)
  stage(0, kX0);
  stage(1, kX1);
  stage(2, kY0);
  stage(3, kY1);
  if (4 < nhalf) {
    stage(4, kX0);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();       // this group runs one barrier behind

  uint4 fy[2][4], fx[2][4];               // dY fragments of the current iq [it][ks]; X fragments [jq][ks]
  for (int t = 0; t < nk; ++t) {
    const uint32_t tb = (uint32_t)(t & 1) * (4 * kHalfW);
    const int g = 4 * t + 5;              // half-tile requested in phase 0 of this K tile
    // ---------------- phase 0: quadrant (i 0, j 0)
    {
      const uint32_t ax = cX[0] + tb, ay0 = cY[0] + tb, ay1 = cY[1] + tb;
      fx[0][0] = lds_tr_frag<0>(ax); fx[0][1] = lds_tr_frag<4096>(ax);
      fx[0][2] = lds_tr_frag<8192>(ax); fx[0][3] = lds_tr_frag<12288>(ax);
      fy[0][0] = lds_tr_frag<0>(ay0); fy[0][1] = lds_tr_frag<4096>(ay0);
      fy[0][2] = lds_tr_frag<8192>(ay0); fy[0][3] = lds_tr_frag<12288>(ay0);
      fy[1][0] = lds_tr_frag<0>(ay1); fy[1][1] = lds_tr_frag<4096>(ay1);
      fy[1][2] = lds_tr_frag<8192>(ay1); fy[1][3] = lds_tr_frag<12288>(ay1);
    }
    if (g < nhalf) stage(g, kX1);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      MmaW8<T>::run(acc[0][0][0], fy[0][ks], fx[0][ks]);
      MmaW8<T>::run(acc[0][1][0], fy[1][ks], fx[0][ks]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 1: quadrant (i 0, j 1)
    {
      const uint32_t ax = cX[1] + tb;
      fx[1][0] = lds_tr_frag<0>(ax); fx[1][1] = lds_tr_frag<4096>(ax);
      fx[1][2] = lds_tr_frag<8192>(ax); fx[1][3] = lds_tr_frag<12288>(ax);
    }
    if (g + 1 < nhalf) stage(g + 1, kY0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      MmaW8<T>::run(acc[0][0][1], fy[0][ks], fx[1][ks]);
      MmaW8<T>::run(acc[0][1][1], fy[1][ks], fx[1][ks]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 2: quadrant (i 1, j 1)
    {
      const uint32_t ay0 = cY[2] + tb, ay1 = cY[3] + tb;
      fy[0][0] = lds_tr_frag<0>(ay0); fy[0][1] = lds_tr_frag<4096>(ay0);
      fy[0][2] = lds_tr_frag<8192>(ay0); fy[0][3] = lds_tr_frag<12288>(ay0);
      fy[1][0] = lds_tr_frag<0>(ay1); fy[1][1] = lds_tr_frag<4096>(ay1);
      fy[1][2] = lds_tr_frag<8192>(ay1); fy[1][3] = lds_tr_frag<12288>(ay1);
    }
    if (g + 2 < nhalf) stage(g + 2, kY1);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      MmaW8<T>::run(acc[1][0][1], fy[0][ks], fx[1][ks]);
      MmaW8<T>::run(acc[1][1][1], fy[1][ks], fx[1][ks]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 3: quadrant (i 1, j 0): operands are in registers; K tile t + 1 must have landed
    if (g + 3 < nhalf) {
      stage(g + 3, kX0);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      MmaW8<T>::run(acc[1][0][0], fy[0][ks], fx[0][ks]);
      MmaW8<T>::run(acc[1][1][0], fy[1][ks], fx[0][ks]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();       // balance the stagger

  // ---- epilogue: acc[iq][it][jq][r] is dW[i][j] with i = wr*128 + iq*64 + it*32 + (r & 3) + 8 * (r >> 2) + 4 * lh,
  //      j = wc*64 + jq*32 + l31: a store instruction writes two 128-byte row segments of the fp32 slab
  float* __restrict__ out = p.slab + ((size_t)split * p.slab_rows + row_base) * p.Ktot;
#pragma unroll
  for (int iq = 0; iq < 2; ++iq)
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) {
        const int j = j0 + wc * 64 + jq * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = i0 + wr * 128 + iq * 64 + it * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (i < Nloc && j < p.Ktot) out[(size_t)i * p.Ktot + j] = acc[iq][it][jq][r];
        }
      }
}

int g_wgrad8_mode = -1;   // -1 automatic, 0 never, 1 whenever covered

}  // namespace

void wgrad8_set_mode(int mode) { g_wgrad8_mode = mode; }

// Covered: 16-bit problems without groups / apply-on-load, 16-byte aligned rows; `gather` false is the DENSE case (one
// (0, 0) tap, stride 1: the caller vouches for it), otherwise the gather variant, which needs Cred % 128 == 0 and no
// second operand.  Automatic: both sides of the gradient at least 256 wide and enough pixels per workgroup to amortise
// the pipeline fill.
bool wgrad8_covers(long M, int N, int N2, int Cred, int ntaps, bool gather, int dtype) {
  if (g_wgrad8_mode == 0 || dtype == kF32 || ntaps < 1 || (!gather && ntaps != 1)) return false;
  if (N % 8 != 0 || N2 % 8 != 0 || Cred % 8 != 0 || M <= 0 || M >= (1l << 31)) return false;
  if (gather && (Cred % 128 != 0 || N2 != 0)) return false;
  if (g_wgrad8_mode == 1) return true;
  const long Ktot = (long)ntaps * Cred;
  const long tiles = (long)(ceil_div(N, 256) + ceil_div(N2, 256)) * ceil_div(Ktot, 256);
  return N >= 256 && Ktot >= 256 && tiles <= 128 && M >= tiles * 8 * 64;
}

// One workgroup per CU: tiles x splits <= 256 (a second round of workgroups would double the time), >= 8 K tiles each.
int wgrad8_choose_splits(long M, int N, int N2, int Ktot, int* pix_per_split) {
  const long tiles = (long)(ceil_div(N, 256) + ceil_div(N2, 256)) * ceil_div(Ktot, 256);
  long splits = tiles >= 256 ? 1 : 256 / tiles;
  const long max_by_work = (M + 8l * 64 - 1) / (8l * 64);
  if (splits > max_by_work) splits = max_by_work;
  const long out_elems = (long)(N + N2) * Ktot;
  const long max_by_slab = (64l << 20) / (out_elems * 4) + 1;
  if (splits > max_by_slab) splits = max_by_slab;
  if (splits < 1) splits = 1;
  long pps = (M + splits - 1) / splits;
  pps = (pps + 63) / 64 * 64;
  splits = (M + pps - 1) / pps;
  *pix_per_split = (int)pps;
  return (int)splits;
}

void wgrad8_launch(const WgradParams& p_in, int dtype, hipStream_t stream) {
  WgradParams p = p_in;
  if (p.dY2 == nullptr) p.N2 = p.ldy2 = 0;
  if (p.M <= 0 || p.N <= 0 || p.Ktot <= 0) return;
  if (dtype == kF32) throw std::invalid_argument("wgrad8: 16-bit types only");
  if (p.groups > 1 || p.x_scale != nullptr || p.ntaps < 1 || p.Ktot != p.ntaps * p.Cred)
    throw std::invalid_argument("wgrad8: dense problems without groups / apply-on-load only");
  const bool gather = !(p.ntaps == 1 && p.taps_zero && p.SY == 1 && p.SX == 1);
  if (gather) {
    const long OHOW = (long)p.div_ohow.d, OW = (long)p.div_ow.d;
    if (p.Cred % 128 != 0 || p.N2 != 0 || p.taps == nullptr || OW <= 0 || OHOW <= 0 || OHOW % OW != 0 || p.M % OHOW != 0 ||
        p.IH <= 0 || p.IW <= 0 || p.SY <= 0 || p.SX <= 0 || (p.M / OHOW) * p.IH * p.IW >= (1l << 31))
      throw std::invalid_argument("wgrad8: the gather variant needs Cred % 128 == 0, no second operand, a tap table and "
                                  "M = images * OH * OW");
  }
  if (p.N % 8 || p.N2 % 8 || p.Cred % 8 || p.ldy % 8 || p.ldx % 8 || p.ldy2 % 8 || p.ldy < p.N || p.ldx < p.Cred ||
      (p.N2 > 0 && p.ldy2 < p.N2))
    throw std::invalid_argument("wgrad8: channel counts / strides must be multiples of 16 bytes");
  if (((uintptr_t)p.dY | (uintptr_t)p.X | (uintptr_t)p.dY2) & 15)
    throw std::invalid_argument("wgrad8: 16-byte aligned operands required");
  if (p.splits < 1 || (long)p.splits * p.pix_per_split < p.M || (long)(p.splits - 1) * p.pix_per_split >= p.M ||
      (p.splits > 1 && p.pix_per_split % 64 != 0))
    throw std::invalid_argument("wgrad8: bad pixel split");
  p.groups = 1;
  p.slab_rows = p.N + p.N2;
  const long blocks = (long)(ceil_div(p.N, 256) + ceil_div(p.N2, 256)) * ceil_div(p.Ktot, 256) * p.splits;
  static bool attr_done = false;
  if (!attr_done) {
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad8_kernel<bf16_t, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kRingW));
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad8_kernel<f16_t, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kRingW));
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad8_kernel<bf16_t, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kRingW));
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad8_kernel<f16_t, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kRingW));
    attr_done = true;
  }
  const dim3 grid((unsigned)blocks), block(512);
  if (dtype == kBF16) {
    if (gather) hipLaunchKernelGGL((wgrad8_kernel<bf16_t, true>), grid, block, kRingW, stream, p);
    else hipLaunchKernelGGL((wgrad8_kernel<bf16_t, false>), grid, block, kRingW, stream, p);
  } else {
    if (gather) hipLaunchKernelGGL((wgrad8_kernel<f16_t, true>), grid, block, kRingW, stream, p);
    else hipLaunchKernelGGL((wgrad8_kernel<f16_t, false>), grid, block, kRingW, stream, p);
  }
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
