// Lean MFMA GEMM for the single-tap problems of the implicit-GEMM interface: 1x1 convolutions without padding
// (forward, stride 1 or s) and their stride-1 data gradients, Linear layers seen as a 1x1 "image".
//
//     C[m][n] = sum_c A[pix(m)][c] * B[n][c]          16-bit in == out, fp32 accumulate
//
// Why a second kernel next to igemm.hip: on these shapes the reduction is short (Cred = 64 .. 2048, i.e. 1 .. 32
// K tiles of 128 bytes) and the general kernel is bound by VALU / SALU issue, not by MFMA or memory -- rocprofv3
// counters on 14x14 256->1024 at batch 1024 (profiles/r2/pmc_1x1_256_1024.md): 1,073 VALU + 494 SALU instructions
// per wave and tile against 64 MFMAs, MFMA busy 20 %, 2.2 TB/s.  Four fifths of those instructions sit outside the
// K loop: tap bookkeeping, 64-bit address arithmetic, bias / ReLU / scatter / tail selects of the epilogue.  This
// kernel carries none of that:
//   * one gather offset per staged row, computed once; the K loop advances two scalar base pointers, every
//     LDS address is a loop-invariant register plus an immediate (the loop is unrolled over both stages);
//   * no padding masks (every output pixel reads an in-range input pixel; rows past M are clamped on load and
//     dropped at the store; N is a multiple of the tile width);
//   * the epilogue is specialised at compile time for what the launch needs: BatchNorm statistics (STATS), the
//     fused accumulate of a second gradient with its ReLU mask (RACC), the fused BatchNorm-backward reduction
//     (BNB) -- nothing else is compiled in -- and it requests the extra operands of RACC / BNB half a tile ahead.
// The accumulator layout, the LDS transposition of the output tile, the Chan-partial statistics slab and the
// [tiles_m][2][N] + [tiles_m] slab format are those of igemm.hip (igemm_stats_rows() agrees: 128-row tiles).
//
// Replaces what cuDNN / cuBLAS ran for the 1x1 convolutions and Linear layers of `model(images)` /
// `loss.backward()` in the reference (nd_imagenet.py:291,302) -- SURVEY.md N-K1, N-K2, N-K10.
#include "../common.h"
#include "kernels.h"
#include <stdexcept>
#include <type_traits>

namespace dpa {

namespace {

// a 16-byte register value as a native vector: copies of HIP's uint4 STRUCT between two memories become memcpy
// calls that the optimiser does not turn into registers (the staging arrays then live in scratch memory)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <typename T> struct LeanMma;
template <> struct LeanMma<bf16_t> {
  static __device__ __forceinline__ void run(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
    bf16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
};
template <> struct LeanMma<f16_t> {
  static __device__ __forceinline__ void run(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
    f16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  }
};

int g_lean_abl = 0;     // debug ablation of the plain variant (scratch/lean_ablate.py)
int g_lean_mode = -1;    // -1 automatic, 0 never, 1 force 64-byte K tiles, 2 force 128-byte K tiles, 3 force 128 x 64 tiles

// BN: output channels per workgroup (128 | 64), BKB: bytes of K per staged row and K tile (64 | 128).
// 256 threads = 2 (pixel) x 2 (channel) waves, 128 pixel rows per workgroup.
// CAT: the reduction runs over two gathered operands back to back (IgemmParams::A2): when the K tiles of the first are
// used up the staging base and the per-row offsets switch to the second, nothing else in the loop changes.
// RELUM (kIgemmReluMask): C = relu(acc + bias + R), ReLU bit mask written next to it.
// FOLD: the BNB launch is a kIgemmBnFold one (masked store, sum dz_m only): the BN input and the coefficients are never
// requested -- compiled apart because their registers decide between 3 and 4 workgroups per CU.
template <typename T, int BN, int BKB, int MINB, int PF, bool STATS, bool RACC, bool BNB, bool STRIDED, int ABL = 0,
          bool BIAS = false, bool AXF = false, bool CAT = false, bool FOLD = false, bool RELUM = false>
__global__ __launch_bounds__(256, MINB) void gemm_lean_kernel(const IgemmParams p) {
  static_assert(sizeof(T) == 2, "16-bit operands");
  static_assert(!FOLD || BNB, "FOLD is a BNB launch");
  static_assert(!RELUM || (BIAS && (RACC != CAT) && !BNB && !STATS),
                "RELUM: relu(acc + bias [+ R]) with the ReLU bit mask as a second output; R or a second operand");
  constexpr int BM = 128;
  constexpr int VEC = 8;
  constexpr int BK = BKB / 2;
  constexpr int CPR = BKB / 16;                 // 16-byte chunks per staged row
  constexpr int LDR = BKB + 16;                 // LDS row pitch of a stage (conflict-free b128 fragment reads)
  constexpr int RPI = 256 / CPR;                // rows staged per pass of the workgroup
  constexpr int A_IT = BM / RPI, B_IT = BN / RPI;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int STAGE = (BM + BN) * LDR;
  static_assert(A_IT >= 1 && B_IT >= 1 && NI >= 1, "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const uint32_t tiles_n = (uint32_t)p.N / BN;
  const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = (int)(lid / tiles_n), nt = (int)(lid - (uint32_t)mt * tiles_n);
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- global offsets of the rows this thread stages: fixed for the whole kernel -----------------------------
  const int ch = tid % CPR, r0 = tid / CPR;
  auto pixel = [&](uint32_t m) __attribute__((always_inline)) -> uint32_t {     // gathered pixel index of output row m
    if constexpr (!STRIDED) {
      return m;
    } else {
      const uint32_t n = fdiv(m, p.div_ohow);
      const uint32_t rem = m - n * p.div_ohow.d;
      const uint32_t oy = fdiv(rem, p.div_ow);
      const uint32_t ox = rem - oy * p.div_ow.d;
      return (n * (uint32_t)p.IH + oy * (uint32_t)p.SY) * (uint32_t)p.IW + ox * (uint32_t)p.SX;
    }
  };
  const uint32_t pix0 = pixel((uint32_t)m0);                         // uniform
  const char* __restrict__ Ag = reinterpret_cast<const char*>(p.A) + (size_t)pix0 * p.lda * sizeof(T);
  const char* __restrict__ Bg = reinterpret_cast<const char*>(p.B) + (size_t)n0 * p.ldb * sizeof(T);
  uint32_t aoff[A_IT], boff[B_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    uint32_t m = (uint32_t)(m0 + r0 + i * RPI);
    m = m < (uint32_t)p.M ? m : (uint32_t)p.M - 1u;                  // rows past M: any valid row, never stored
    aoff[i] = ((pixel(m) - pix0) * (uint32_t)p.lda + (uint32_t)(ch * VEC)) * (uint32_t)sizeof(T);
  }
#pragma unroll
  for (int i = 0; i < B_IT; ++i)
    boff[i] = ((uint32_t)(r0 + i * RPI) * (uint32_t)p.ldb + (uint32_t)(ch * VEC)) * (uint32_t)sizeof(T);
  uint32_t aoff2[CAT ? A_IT : 1], boff2[CAT ? B_IT : 1];
  const char* Ag2 = nullptr;
  const char* Bg2 = nullptr;
  int cat_left = CAT ? p.Cred / BK : 0;          // K tiles of the first operand still to request
  if constexpr (CAT) {
    Ag2 = reinterpret_cast<const char*>(p.A2) + (size_t)pix0 * p.lda2 * sizeof(T);
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      uint32_t m = (uint32_t)(m0 + r0 + i * RPI);
      m = m < (uint32_t)p.M ? m : (uint32_t)p.M - 1u;
      aoff2[i] = ((pixel(m) - pix0) * (uint32_t)p.lda2 + (uint32_t)(ch * VEC)) * (uint32_t)sizeof(T);
    }
    if (p.B2 != nullptr) {
      Bg2 = reinterpret_cast<const char*>(p.B2) + (size_t)n0 * p.ldb2 * sizeof(T);
#pragma unroll
      for (int i = 0; i < B_IT; ++i)
        boff2[i] = ((uint32_t)(r0 + i * RPI) * (uint32_t)p.ldb2 + (uint32_t)(ch * VEC)) * (uint32_t)sizeof(T);
    }
  }

  // ---- LDS addresses (loop invariant) ---------------------------------------------------------------------
  unsigned char* const wr_a = smem + r0 * LDR + ch * 16;                       // + i * RPI * LDR
  unsigned char* const wr_b = smem + BM * LDR + r0 * LDR + ch * 16;
  const unsigned char* const rd_x = smem + (wm * WTM + l31) * LDR + lh * 16;   // + b * 32 * LDR + ks * 32
  const unsigned char* const rd_w = smem + BM * LDR + (wn * WTN + l31) * LDR + lh * 16;

  // BIAS (Linear layers): the accumulators START at the bias of their channel -- register r of lane half lh is channel
  // (r&3) + 8*(r>>2) + 4*lh of the 32-block -- so the epilogue has nothing to add (one fp32 rounding, as in the
  // general kernel; only the order of the fp32 additions differs).
  // (called BEHIND the first tile's requests: in front of them the bias load's latency would be paid before the
  // pipeline starts)
  f32x16_t acc[NI][MI];
  auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < NI; ++a) {
      float bv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bv[r] = 0.f;
      if constexpr (BIAS) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n0 + wn * WTN + a * 32 + 8 * g + 4 * lh);
          bv[4 * g + 0] = b4.x; bv[4 * g + 1] = b4.y; bv[4 * g + 2] = b4.z; bv[4 * g + 3] = b4.w;
        }
      }
#pragma unroll
      for (int b = 0; b < MI; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = bv[r];
    }
  };

  // PF register sets: PF K tiles are in flight while one is consumed (a K step computes for ~300 cycles, a load
  // takes 1-2k under load: with one tile in flight every K step pays most of a memory latency)
  u32x4_t ra[PF][A_IT], rb[PF][B_IT];
  // AXF (kIgemmAxf): A is the raw input of a BatchNorm + ReLU; relu(scale[c] * z + shift[c]) is applied when a tile
  // goes from registers to LDS (bit-identical to the stand-alone pass, igemm.hip).  The (scale, shift) table sits
  // behind the two stages, one 64-byte entry per 8-channel chunk; a thread stages ONE chunk column per K tile.
  float* const axf_tab = reinterpret_cast<float*>(smem + 2 * STAGE);
  int axf_q[PF] = {}, axf_kt = 0;
  if constexpr (AXF) {
    for (int c = tid; c < p.Cred; c += 256) {
      float* t = axf_tab + (c / VEC) * (2 * VEC) + (c % VEC);
      t[0] = p.a_scale[c];
      t[VEC] = p.a_shift[c];
    }
    __syncthreads();
  }
  auto load_tile = [&](int set) __attribute__((always_inline)) {
    if constexpr (CAT) {
      if (cat_left-- == 0) {                     // (uniform, once per kernel) on to the second operand
        Ag = Ag2;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) aoff[i] = aoff2[i];
        if (Bg2 != nullptr) {
          Bg = Bg2;
#pragma unroll
          for (int i = 0; i < B_IT; ++i) boff[i] = boff2[i];
        }
      }
    }
    if constexpr (AXF) axf_q[set] = (axf_kt++) * CPR + ch;
    if ((ABL == 3 || ABL >= 5) && Ag != reinterpret_cast<const char*>(p.A) + (size_t)pix0 * p.lda * sizeof(T)) {
      Ag += BKB;      // (keeps the first-tile test of store_tile exact)
      return;
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) ra[set][i] = *reinterpret_cast<const u32x4_t*>(Ag + aoff[i]);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) rb[set][i] = *reinterpret_cast<const u32x4_t*>(Bg + boff[i]);
    Ag += BKB;
    Bg += BKB;
  };
  auto store_tile = [&](int set, int stage) __attribute__((always_inline)) {
    if (ABL >= 5 && Ag != reinterpret_cast<const char*>(p.A) + (size_t)pix0 * p.lda * sizeof(T) + BKB) return;
    float asc[VEC], ash[VEC];
    if constexpr (AXF) {
      const float4* t4 = reinterpret_cast<const float4*>(axf_tab + axf_q[set] * (2 * VEC));
      const float4 s0 = t4[0], s1 = t4[1], h0 = t4[2], h1 = t4[3];
      asc[0] = s0.x; asc[1] = s0.y; asc[2] = s0.z; asc[3] = s0.w; asc[4] = s1.x; asc[5] = s1.y; asc[6] = s1.z; asc[7] = s1.w;
      ash[0] = h0.x; ash[1] = h0.y; ash[2] = h0.z; ash[3] = h0.w; ash[4] = h1.x; ash[5] = h1.y; ash[6] = h1.z; ash[7] = h1.w;
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      u32x4_t v = ra[set][i];
      if constexpr (AXF) {
        const uint4 r = bn_relu_chunk<T>(make_uint4(v.x, v.y, v.z, v.w), asc, ash);
        v = u32x4_t{r.x, r.y, r.z, r.w};
      }
      *reinterpret_cast<u32x4_t*>(wr_a + stage * STAGE + i * RPI * LDR) = v;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) *reinterpret_cast<u32x4_t*>(wr_b + stage * STAGE + i * RPI * LDR) = rb[set][i];
  };
  // All fragment reads of a K tile are written ahead of its MFMAs; the compiler interleaves them with counted
  // lgkmcnt waits.  (Measured neutral against read-just-before-use: a pure ds_read + MFMA + barrier loop -- no global
  // loads, no LDS writes -- runs at ~50 % MFMA utilisation either way, profiles/r2/lean_gemm_ablation.txt.)
  auto compute = [&](int stage) __attribute__((always_inline)) {
    if (ABL == 2) return;
    constexpr int KS = BKB / 32;
    u32x4_t wf[KS][NI], xf[KS][MI];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int a = 0; a < NI; ++a)
        wf[ks][a] = *reinterpret_cast<const u32x4_t*>(rd_w + stage * STAGE + a * 32 * LDR + ks * 32);
#pragma unroll
      for (int b = 0; b < MI; ++b)
        xf[ks][b] = *reinterpret_cast<const u32x4_t*>(rd_x + stage * STAGE + b * 32 * LDR + ks * 32);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) {
          if (ABL == 1) asm volatile("" ::"v"(wf[ks][a]), "v"(xf[ks][b]));
          else LeanMma<T>::run(acc[a][b], wf[ks][a], xf[ks][b]);
        }
  };

  // (the lambdas above are force-inlined: an out-of-line call would put the staging registers in scratch memory)
  // two K tiles per trip so that stage and register-set indices are compile-time constants
  auto loop_sync = [&]() __attribute__((always_inline)) {
    if (ABL != 6) __syncthreads();      // ablation 6: how much of the pure read + MFMA loop is barrier skew
  };
  const int nk = p.Cred / BK + (CAT ? p.Cred2 / BK : 0);
  int kt = 0;
  if constexpr (PF == 1) {
    load_tile(0);
    init_acc();
    store_tile(0, 0);
    loop_sync();
    for (; kt + 2 <= nk; kt += 2) {
      load_tile(0);                     // tile kt + 1 (exists)
      compute(0);
      store_tile(0, 1);
      loop_sync();
      const bool more = kt + 2 < nk;
      if (more) load_tile(0);
      compute(1);
      if (more) store_tile(0, 0);
      loop_sync();
    }
  } else {
    load_tile(0);
    if (nk > 1) load_tile(1);
    init_acc();
    store_tile(0, 0);
    loop_sync();
    for (; kt + 2 <= nk; kt += 2) {
      if (kt + 2 < nk) load_tile(0);    // tile kt + 2 into the set tile kt has just left
      compute(0);
      store_tile(1, 1);                 // tile kt + 1: waits for ITS loads only, tile kt + 2 stays in flight
      loop_sync();
      if (kt + 3 < nk) load_tile(1);
      compute(1);
      if (kt + 2 < nk) store_tile(0, 0);
      loop_sync();
    }
  }
  if (kt < nk) {                        // odd tile count: the last tile sits in stage 0
    compute(0);
    loop_sync();
  }

  if (ABL == 6) __syncthreads();
  // ---- epilogue: registers -> LDS tile [pixel][channel] -> 16-byte global stores ---------------------------
  // acc[a][b][r]: channel = wn*WTN + a*32 + (r&3) + 8*(r>>2) + 4*lh, pixel = wm*WTM + b*32 + l31
  constexpr int LDE = BN * 2 + 16;
  constexpr int CE = BN / VEC;                  // 16-byte chunks per output row
  constexpr int RPE = 256 / CE;                 // output rows per pass
  constexpr int E_IT = BM / RPE;
  constexpr int EH = E_IT / 2;                  // the rows are processed as two halves (see prefetch below)
  static_assert(E_IT % 2 == 0, "two halves");
  unsigned char* const se = smem;

  const int ce = tid % CE, er0 = tid / CE;
  // 32-bit offsets from the tile's first output row (a tile spans 128 rows of ldc elements)
  const size_t cbase = ((size_t)m0 * p.ldc + (size_t)n0) * sizeof(T);
  char* const Cg = reinterpret_cast<char*>(p.C) + cbase;
  const uint32_t coff0 = ((uint32_t)er0 * (uint32_t)p.ldc + (uint32_t)(ce * VEC)) * (uint32_t)sizeof(T);
  const uint32_t cstep = (uint32_t)RPE * (uint32_t)p.ldc * (uint32_t)sizeof(T);
  const int rows = min(BM, p.M - m0);           // uniform

  // Operands the epilogue streams besides the tile itself: R (fused accumulate: the second gradient of a tensor
  // with two consumers, with the ReLU bit mask it still has to pass) and the BatchNorm input / ReLU bit mask of the
  // fused BN-backward reduction.  On these launches the epilogue moves 2-4x the bytes of the GEMM operands, so it
  // is written like a streaming kernel: every 16-byte row piece of a half tile is requested up front -- the first
  // half before the accumulators go through LDS, the second before the first is consumed -- instead of one
  // dependent load per row.
  struct Pre {
    u32x4_t r[EH], x[EH];
    uint32_t rbits[EH], xbits[EH];
  };
  const int bn_mode = BNB ? (p.bn_relu & 3) : 0;   // 0 no ReLU, 2 mask recomputed from the BN input, 3 bit mask
  const bool bn_clip = BNB && (p.bn_relu & 4);     // ReLU6: the recomputed mask is 0 < y < 6
  constexpr bool bn_fold = FOLD;                     // masked store, sum dz_m only, bn_x not read (kernels.h)
  const char* const Rg = reinterpret_cast<const char*>(p.R) + cbase;
  const char* const Xg = reinterpret_cast<const char*>(p.bn_x) + cbase;
  const uint8_t* const rmask = RACC ? p.r_mask : nullptr;
  const bool rstr = RACC && (p.flags & kIgemmRStrided);
  const int lsy = rstr ? __builtin_ctz((unsigned)p.OSY) : 0, lsx = rstr ? __builtin_ctz((unsigned)p.OSX) : 0;
  const uint8_t* const xmask = (BNB && bn_mode == 3) ? reinterpret_cast<const uint8_t*>(p.bn_z) : nullptr;
  auto prefetch = [&](int half, Pre& q) __attribute__((always_inline)) {
    if (ABL == 7) {                     // ablation: the epilogue's extra operands are never requested
#pragma unroll
      for (int i = 0; i < EH; ++i) {
        q.r[i] = q.x[i] = u32x4_t{0u, 0u, 0u, 0u};
        q.rbits[i] = q.xbits[i] = 0xffu;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < EH; ++i) {
      const int it = half * EH + i;
      const bool ok = er0 + it * RPE < rows;
      const uint32_t off = ok ? coff0 + (uint32_t)it * cstep : (uint32_t)(ce * 16);   // row 0 always exists
      if constexpr (RACC) {
        if (rstr) {
          // R is COMPACT: the gradient of a stride-s 1x1 branch on its own (OHF x OWF) grid; output pixel (y, x)
          // takes row (y / s, x / s) when both are multiples of s and nothing otherwise
          const uint32_t mm = ok ? (uint32_t)(m0 + er0 + it * RPE) : (uint32_t)m0;
          const uint32_t n = fdiv(mm, p.div_ohow);
          const uint32_t rem = mm - n * p.div_ohow.d;
          const uint32_t y = fdiv(rem, p.div_ow);
          const uint32_t x = rem - y * p.div_ow.d;
          const bool hit = ok && !(y & (uint32_t)(p.OSY - 1)) && !(x & (uint32_t)(p.OSX - 1));
          const size_t cp = hit ? ((size_t)n * p.OHF + (y >> lsy)) * p.OWF + (x >> lsx) : (size_t)0;
          q.r[i] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(p.R) +
                                                     (cp * p.ldc + (size_t)(n0 + ce * VEC)) * sizeof(T));
          q.rbits[i] = hit ? 0xffu : 0u;
        } else {
          q.r[i] = *reinterpret_cast<const u32x4_t*>(Rg + off);
          q.rbits[i] = rmask != nullptr ? rmask[(cbase + off) >> 4] : 0xffu;
        }
      }
      if constexpr (BNB) q.xbits[i] = xmask != nullptr ? xmask[(cbase + off) >> 4] : 0xffu;
    }
    if constexpr (BNB && !FOLD) {
#pragma unroll
      for (int i = 0; i < EH; ++i) {
        const int it = half * EH + i;
        const bool ok = er0 + it * RPE < rows;
        const uint32_t off = ok ? coff0 + (uint32_t)it * cstep : (uint32_t)(ce * 16);
        q.x[i] = *reinterpret_cast<const u32x4_t*>(Xg + off);
      }
    }
  };
  // per-channel coefficients of the BN-backward sums, requested BEFORE the half-tile prefetches: vmcnt retires in
  // order, a load issued behind prefetch(1) would make the first half wait for the second half's operands (that
  // cost 20 % of the fused accumulate + reduction launches, bench/experiments/lean_racc_ablate.py).  kIgemmBnFold
  // sums dz_m only and needs none.
  float bmu[VEC], bis[VEC], bsc[VEC], bsh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) bmu[j] = bis[j] = bsc[j] = bsh[j] = 0.f;
  if constexpr (BNB && !FOLD) {
    {
      const float* cf = p.bn_coef + n0 + ce * VEC;          // [4][N]: mean, invstd, scale, shift
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        bmu[j] = cf[j];
        bis[j] = cf[p.N + j];
        bsc[j] = bn_mode == 2 ? cf[2 * p.N + j] : 0.f;
        bsh[j] = bn_mode == 2 ? cf[3 * p.N + j] : 0.f;
      }
    }
  }
  Pre q0, q1;
  if constexpr (RACC || BNB) prefetch(0, q0);
  const bool relu_out = BIAS && (p.flags & kIgemmRelu);

#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int b = 0; b < MI; ++b) {
        T o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[a][b][4 * g + j];
          if (BIAS && relu_out) v = fmaxf(v, 0.f);
          o[j] = from_float<T>(v);
        }
        uint2 w;
        __builtin_memcpy(&w, o, 8);
        *reinterpret_cast<uint2*>(se + (wm * WTM + b * 32 + l31) * LDE + (wn * WTN + a * 32 + 8 * g + 4 * lh) * 2) = w;
      }
  // LDS-only barrier: __syncthreads() would also wait for the prefetch (vmcnt(0))
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if constexpr (RACC || BNB) prefetch(1, q1);

  float s1[VEC], s2[VEC], kshift[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = kshift[j] = 0.f;
  if constexpr (STATS) {
    // Chan partials accumulated around the tile's first row (a sample: |v - k| ~ std), see igemm.hip
    Vec16<T> k0;
    k0.load(se + ce * 16);
#pragma unroll
    for (int j = 0; j < VEC; ++j) kshift[j] = k0.get(j);
  }
  const bool nt_store = p.flags & kIgemmNtStore;
  // packed accumulators of the bf16 BN-backward epilogue (pair j = channels 2j, 2j+1 of the chunk)
  f32x2_t s1p[VEC / 2], s2p[VEC / 2], bisp[VEC / 2], nbp[VEC / 2];
#pragma unroll
  for (int j = 0; j < VEC / 2; ++j) {
    s1p[j] = s2p[j] = f32x2_t{0.f, 0.f};
    bisp[j] = nbp[j] = f32x2_t{0.f, 0.f};
    if constexpr (BNB) {
      bisp[j] = f32x2_t{bis[2 * j], bis[2 * j + 1]};
      nbp[j] = f32x2_t{-bmu[2 * j] * bis[2 * j], -bmu[2 * j + 1] * bis[2 * j + 1]};      // xhat = x * invstd - mean * invstd
    }
  }
  auto process = [&](int half, const Pre& q, auto mode_c) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
    for (int i = 0; i < EH; ++i) {
      const int it = half * EH + i;
      const int row = er0 + it * RPE;
      if (row >= rows) continue;
      char* dst = Cg + (coff0 + (uint32_t)it * cstep);
      if constexpr (RELUM) {
        uint32_t bits = 0u;
        if constexpr (TypeInfo<T>::kCode == kBF16) {
          const u32x4_t vv = *reinterpret_cast<const u32x4_t*>(se + row * LDE + ce * 16);
          u32x4_t o;
#pragma unroll
          for (int pr = 0; pr < 4; ++pr) {
            uint32_t w = vv[pr];
            if constexpr (RACC) {
              const uint32_t rm = q.r[i][pr] & pair_mask(q.rbits[i], pr);
              w = pack_bf16x2(unpack_bf16x2(w) + unpack_bf16x2(rm));
            }
            const uint32_t m = pair_positive_mask(w);
            o[pr] = w & m;
            bits |= ((m & 1u) | ((m >> 15) & 2u)) << (2 * pr);
          }
          if (nt_store) __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t*>(dst));
          else *reinterpret_cast<u32x4_t*>(dst) = o;
        } else {
          Vec16<T> v, r;
          v.load(se + row * LDE + ce * 16);
          r.raw = RACC ? make_uint4(q.r[i].x, q.r[i].y, q.r[i].z, q.r[i].w) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float f = v.get(j) + ((RACC && ((q.rbits[i] >> j) & 1u)) ? r.get(j) : 0.f);
            v.set(j, fmaxf(f, 0.f));
            bits |= (v.get(j) > 0.f ? 1u : 0u) << j;
          }
          if (nt_store) v.store_nt(dst);
          else v.store(dst);
        }
        p.relu_mask[(cbase + coff0 + (size_t)it * cstep) >> 4] = (uint8_t)bits;
        continue;
      }
      if constexpr (BNB && !STATS && MODE != 2 && TypeInfo<T>::kCode == kBF16) {
        // bf16, mask absent (0: xbits = 0xff) or as bits (3): whole pairs at a time
        const u32x4_t vv = *reinterpret_cast<const u32x4_t*>(se + row * LDE + ce * 16);
        u32x4_t o;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          uint32_t w = vv[pr];
          if constexpr (RACC) {
            const uint32_t rm = q.r[i][pr] & pair_mask(q.rbits[i], pr);
            w = pack_bf16x2(unpack_bf16x2(w) + unpack_bf16x2(rm));
          }
          const uint32_t wm = w & pair_mask(q.xbits[i], pr);
          o[pr] = FOLD ? wm : w;              // the gradient leaves already masked under kIgemmBnFold
          const f32x2_t d = unpack_bf16x2(wm);      // the values as stored (rounded to T), masked
          s1p[pr] += d;
          if constexpr (!FOLD) {
            const f32x2_t xh = unpack_bf16x2(q.x[i][pr]) * bisp[pr] + nbp[pr];
            s2p[pr] += d * xh;
          }
        }
        if (ABL == 4) asm volatile("" ::"v"(o));
        else if (nt_store) __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t*>(dst));
        else *reinterpret_cast<u32x4_t*>(dst) = o;
        continue;
      }
      Vec16<T> v;
      v.load(se + row * LDE + ce * 16);
      if constexpr (RACC) {
        Vec16<T> r;
        r.raw = make_uint4(q.r[i].x, q.r[i].y, q.r[i].z, q.r[i].w);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v.set(j, v.get(j) + (((q.rbits[i] >> j) & 1u) ? r.get(j) : 0.f));
      }
      if constexpr (BNB) {
        if (FOLD && MODE == 3) {       // the gradient leaves already masked (kIgemmBnFold)
#pragma unroll
          for (int j = 0; j < VEC; ++j)
            if (!((q.xbits[i] >> j) & 1u)) v.set(j, 0.f);
        }
      }
      if (ABL == 4) asm volatile("" ::"v"(v.raw.x), "v"(v.raw.y), "v"(v.raw.z), "v"(v.raw.w));
      else if (nt_store) v.store_nt(dst);
      else v.store(dst);
      if constexpr (STATS) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float f = v.get(j) - kshift[j];
          s1[j] += f;
          s2[j] += f * f;
        }
      }
      if constexpr (BNB) {
        // v holds the values as stored (rounded to T): exactly what BatchNorm backward will read back
        Vec16<T> xb;
        xb.raw = FOLD ? make_uint4(0u, 0u, 0u, 0u) : make_uint4(q.x[i].x, q.x[i].y, q.x[i].z, q.x[i].w);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float d = v.get(j);
          const float xv = xb.get(j);
          if (MODE == 3 && !((q.xbits[i] >> j) & 1u)) d = 0.f;
          if (MODE == 2) {
            const float yv = to_float(from_float<T>(bsc[j] * xv + bsh[j]));
            if (!(yv > 0.f) || (bn_clip && !(yv < 6.f))) d = 0.f;
          }
          s1[j] += d;
          s2[j] += d * ((xv - bmu[j]) * bis[j]);      // (fold: x = 0 here, the row is ignored by the consumer)
        }
      }
    }
  };
  using std::integral_constant;
  if (bn_mode == 3) {
    process(0, q0, integral_constant<int, 3>());
    process(1, q1, integral_constant<int, 3>());
  } else if (bn_mode == 2) {
    process(0, q0, integral_constant<int, 2>());
    process(1, q1, integral_constant<int, 2>());
  } else {
    process(0, q0, integral_constant<int, 0>());
    process(1, q1, integral_constant<int, 0>());
  }
#pragma unroll
  for (int j = 0; j < VEC / 2; ++j) {
    s1[2 * j] += s1p[j].x; s1[2 * j + 1] += s1p[j].y;
    s2[2 * j] += s2p[j].x; s2[2 * j + 1] += s2p[j].y;
  }
  if constexpr ((STATS || BNB) && ABL != 8) {
    if (ABL == 10) {                    // ablation: the sums are accumulated but never reduced
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) t += s1[j] + s2[j];
      if (t == 12345.678f) p.stats[0] = t;
      return;
    }
    // threads sharing a chunk column: tid = ce (mod CE): reduce inside the wave, then over waves
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
#pragma unroll
      for (int o = 32; o >= CE; o >>= 1) {
        s1[j] += __shfl_xor(s1[j], o, 64);
        if (!bn_fold) s2[j] += __shfl_xor(s2[j], o, 64);       // (fold: the second row is all zeros)
      }
    }
    if (ABL == 11) {                    // ablation: reduced inside the wave, not over the waves
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) t += s1[j] + s2[j];
      if (t == 12345.678f) p.stats[0] = t;
      return;
    }
    // LDS-only barrier: __syncthreads() would first drain the tile's global stores (vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][BN] + shifts [BN]
    if (lane < CE) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        red[(wave * 2 + 0) * BN + ce * VEC + j] = s1[j];
        red[(wave * 2 + 1) * BN + ce * VEC + j] = s2[j];
        if (STATS && wave == 0) red[8 * BN + ce * VEC + j] = kshift[j];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int sld = p.stats_ld;
    if constexpr (BNB) {
      // BatchNorm-backward partials of this tile: (sum dz_m, sum dz_m * xhat)
      for (int c = tid; c < (bn_fold ? BN : 2 * BN); c += 256) {       // (fold: the consumer never reads the second row)
        const int which = c / BN, col = c - which * BN;
        const float s = red[(0 * 2 + which) * BN + col] + red[(1 * 2 + which) * BN + col] +
                        red[(2 * 2 + which) * BN + col] + red[(3 * 2 + which) * BN + col];
        if (ABL == 9) asm volatile("" ::"v"(s));
        else p.stats[((size_t)mt * 2 + which) * sld + n0 + col] = s;
      }
    } else {
      const float nt_rows = (float)rows;
      if (tid < BN) {
        const int col = tid;
        const float a = red[0 * BN + col] + red[2 * BN + col] + red[4 * BN + col] + red[6 * BN + col];
        const float b = red[1 * BN + col] + red[3 * BN + col] + red[5 * BN + col] + red[7 * BN + col];
        const float k = red[8 * BN + col];
        p.stats[((size_t)mt * 2 + 0) * sld + n0 + col] = a + nt_rows * k;
        p.stats[((size_t)mt * 2 + 1) * sld + n0 + col] = fmaxf(b - a * a / nt_rows, 0.f);
      }
      // rows of this part, behind the slab: [tiles_m][2][stats_ld] floats, then [tiles_m] counts
      if (tid == 0 && n0 == 0) {
        const uint32_t tiles_m = gridDim.x / tiles_n;
        p.stats[(size_t)tiles_m * 2 * sld + mt] = nt_rows;
      }
    }
  }
}

template <typename T, int BN, int BKB, int MINB, int PF, bool STATS, bool RACC, bool BNB, bool STRIDED, int ABL = 0,
          bool BIAS = false, bool AXF = false, bool CAT = false, bool FOLD = false, bool RELUM = false>
void launch_lean(const IgemmParams& p, hipStream_t stream) {
  constexpr int LDR = BKB + 16;
  constexpr int stage_bytes = (128 + BN) * LDR;
  constexpr int epi_bytes = 128 * (BN * 2 + 16);
  constexpr int red_bytes = 9 * BN * 4;
  constexpr int max_lds = (2 * stage_bytes > epi_bytes ? 2 * stage_bytes : epi_bytes) + (AXF ? 8 * 2048 : 0);
  int lds = ((p.Cred + p.Cred2) * 2 > BKB ? 2 : 1) * stage_bytes;     // one K tile: no second staging buffer
  if (AXF) lds = 2 * stage_bytes + 8 * p.Cred;            // (scale, shift) table behind both stages
  if (lds < epi_bytes) lds = epi_bytes;
  if (lds < red_bytes) lds = red_bytes;
  auto kern = gemm_lean_kernel<T, BN, BKB, MINB, PF, STATS, RACC, BNB, STRIDED, ABL, BIAS, AXF, CAT, FOLD, RELUM>;
  static bool attr_done = false;
  if (!attr_done && max_lds > 64 * 1024) {
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    attr_done = true;
  }
  const long tiles = (long)ceil_div(p.M, 128) * (p.N / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
}

template <typename T, int BN, int BKB, int MINB, int PF>
void launch_lean_epi(const IgemmParams& p, bool strided, hipStream_t stream) {
  const bool bnb = p.flags & kIgemmBnBwd;
  const bool stats = p.stats != nullptr && !bnb, racc = p.R != nullptr;
  if (p.flags & kIgemmAxf) {           // forward with BN + ReLU applied on load: always wants the next BN's statistics
    launch_lean<T, BN, BKB, MINB, PF, true, false, false, false, 0, false, true>(p, stream);
    return;
  }
  if (p.flags & kIgemmReluMask) {      // bottleneck tail: relu(acc + shift + identity) + bit mask; two branches as two operands
    if (p.A2 != nullptr) launch_lean<T, BN, BKB, MINB, PF, false, false, false, false, 0, true, false, true, false, true>(p, stream);
    else launch_lean<T, BN, BKB, MINB, PF, false, true, false, false, 0, true, false, false, false, true>(p, stream);
    return;
  }
  if (p.A2 != nullptr) {               // two operands back to back + bias ("bn fold" data gradient), plain or with the reduction
    if (bnb) launch_lean<T, BN, BKB, (BN == 128 && BKB == 64) ? 4 : MINB, PF, false, false, true, false, 0, true, false, true>(p, stream);
    else launch_lean<T, BN, BKB, MINB, PF, false, false, false, false, 0, true, false, true>(p, stream);
    return;
  }
  if (p.bias != nullptr) {             // Linear forward (gemm_lean_try admits a bias only on the plain variant)
    launch_lean<T, BN, BKB, MINB, PF, false, false, false, false, 0, true>(p, stream);
    return;
  }
  if (strided) {                       // strided gathers: forward only (statistics or nothing)
    if (stats) launch_lean<T, BN, BKB, MINB, PF, true, false, false, true>(p, stream);
    else launch_lean<T, BN, BKB, MINB, PF, false, false, false, true>(p, stream);
  } else if (bnb) {
#ifdef DPA_LEAN_ABLATE
    if constexpr (BN == 128 && sizeof(T) == 2) {
      if (racc) switch (g_lean_abl) {
        case 3: launch_lean<T, BN, BKB, MINB, PF, false, true, true, false, 3>(p, stream); return;
        case 4: launch_lean<T, BN, BKB, MINB, PF, false, true, true, false, 4>(p, stream); return;
        case 5: launch_lean<T, BN, BKB, MINB, PF, false, true, true, false, 5>(p, stream); return;
        case 7: launch_lean<T, BN, BKB, MINB, PF, false, true, true, false, 7>(p, stream); return;
        case 8: launch_lean<T, BN, BKB, MINB, PF, false, true, true, false, 8>(p, stream); return;
        case 9: launch_lean<T, BN, BKB, MINB, PF, false, true, true, false, 9>(p, stream); return;
        case 10: launch_lean<T, BN, BKB, MINB, PF, false, true, true, false, 10>(p, stream); return;
        case 11: launch_lean<T, BN, BKB, MINB, PF, false, true, true, false, 11>(p, stream); return;
        default: break;
      }
    }
#endif
    // (register budget of 4 workgroups per CU where the tile leaves the LDS for them: 128 x 128 tiles with 64-byte K
    // tiles; the fold variants fit it, the others are 1-2 registers over without the bound)
    constexpr int MB = (BN == 128 && BKB == 64) ? 4 : MINB;
    if (p.flags & kIgemmBnFold) {
      if (racc) launch_lean<T, BN, BKB, MB, PF, false, true, true, false, 0, false, false, false, true>(p, stream);
      else launch_lean<T, BN, BKB, MB, PF, false, false, true, false, 0, false, false, false, true>(p, stream);
    } else if (racc) {
      launch_lean<T, BN, BKB, MINB, PF, false, true, true, false>(p, stream);
    } else {
      launch_lean<T, BN, BKB, MB, PF, false, false, true, false>(p, stream);
    }
  } else if (stats) {
    launch_lean<T, BN, BKB, MINB, PF, true, false, false, false>(p, stream);
  } else if (racc) {
    launch_lean<T, BN, BKB, MINB, PF, false, true, false, false>(p, stream);
  } else {
#ifdef DPA_LEAN_ABLATE
    if constexpr (BN == 128 && sizeof(T) == 2) {
      switch (g_lean_abl) {
        case 1: launch_lean<T, BN, BKB, MINB, PF, false, false, false, false, 1>(p, stream); return;
        case 2: launch_lean<T, BN, BKB, MINB, PF, false, false, false, false, 2>(p, stream); return;
        case 3: launch_lean<T, BN, BKB, MINB, PF, false, false, false, false, 3>(p, stream); return;
        case 4: launch_lean<T, BN, BKB, MINB, PF, false, false, false, false, 4>(p, stream); return;
        case 5: launch_lean<T, BN, BKB, MINB, PF, false, false, false, false, 5>(p, stream); return;
        case 6: launch_lean<T, BN, BKB, MINB, PF, false, false, false, false, 6>(p, stream); return;
        default: break;
      }
    }
#endif
    launch_lean<T, BN, BKB, MINB, PF, false, false, false, false>(p, stream);
  }
}


template <typename T>
void launch_lean_tiles(const IgemmParams& p, bool strided, hipStream_t stream) {
  // A 256 x 256 tile (8 waves of 128 x 64, half the staged bytes per MFMA) in this same loop structure measured
  // +3 .. 15 % on the deep layers and 896 vs 822 TFLOP/s on a 200k x 2048 x 2048 GEMM
  // (profiles/r2/lean_gemm_256x256_tile.txt): not worth a second set of epilogues; the structure, not the tile, caps it.
  const bool wide = p.N % 128 == 0 && g_lean_mode != 3;      // (mode 3: tuning, 64-channel tiles everywhere)
  const long tiles = (long)ceil_div(p.M, 128) * (p.N / (wide ? 128 : 64));
  // 128-byte K tiles (half the barriers) for long reductions and launches that barely fill the chip, as in
  // igemm.hip's choose_cfg(); 64-byte K tiles (more workgroups per CU) otherwise
  bool k128 = p.Cred % 64 == 0 && p.Cred2 % 64 == 0 && ((long)(p.Cred + p.Cred2) * 2 >= 4096 || tiles <= 512);
  if (g_lean_mode == 1) k128 = false;
  if (g_lean_mode == 2) k128 = p.Cred % 64 == 0 && p.Cred2 % 64 == 0;
  // PF = 2 (two K tiles in flight in registers) measured the same as PF = 1 on every ResNet-50 layer
  // (profiles/r2/lean_gemm_prefetch_depth.txt): the K loop is not what these launches wait for
  if (wide) {
    if (k128) launch_lean_epi<T, 128, 128, 2, 1>(p, strided, stream);
    else launch_lean_epi<T, 128, 64, 3, 1>(p, strided, stream);
  } else {
    if (k128) launch_lean_epi<T, 64, 128, 3, 1>(p, strided, stream);
    else launch_lean_epi<T, 64, 64, 4, 1>(p, strided, stream);
  }
}

}  // namespace

void igemm_set_lean(int mode) { g_lean_mode = mode; }
void igemm_set_lean_ablate(int a) { g_lean_abl = a; }

// Takes the launch when the problem is one the lean kernel covers; false = the caller runs the general kernel.
bool gemm_lean_try(const IgemmParams& p, int dtype, int out_dtype, hipStream_t stream) {
  if (g_lean_mode == 0 && !(p.flags & kIgemmRStrided)) return false;    // (a compact accumulate has no other kernel)
  if (dtype != out_dtype || (dtype != kBF16 && dtype != kF16)) return false;
  if (!(p.flags & kIgemmDense1) || p.ntaps != 1) return false;
  if (p.flags & kIgemmScatter) return false;
  if (p.groups > 1) return false;
  const bool bnb = p.flags & kIgemmBnBwd;
  // bias (+ ReLU): Linear layers, plain epilogue only
  if ((p.flags & kIgemmRelu) && p.bias == nullptr) return false;
  const bool cat = p.A2 != nullptr;
  if (cat && (p.bias == nullptr || p.R != nullptr || (p.stats != nullptr && !bnb) || p.Cred2 <= 0 || p.Cred2 % 32 ||
              p.lda2 % 8 || (p.B2 != nullptr && p.ldb2 % 8) || (p.flags & (kIgemmAxf | kIgemmRStrided))))
    return false;
  const bool relum = p.flags & kIgemmReluMask;
  if (relum && (bnb || p.bias == nullptr || (p.R == nullptr) == !cat || p.relu_mask == nullptr || p.stats != nullptr ||
                p.ldc != p.N || (p.flags & (kIgemmAxf | kIgemmRStrided | kIgemmRelu))))
    return false;
  if (!cat && !relum && p.bias != nullptr && (p.stats != nullptr || p.R != nullptr || bnb)) return false;
  if (p.stats != nullptr && ((p.R != nullptr && !bnb) || (p.stats_ld != 0 && p.stats_ld != p.N))) return false;
  // fused BN-backward reduction: dense rows; the ReLU mask as bits (3), recomputed (2) or absent (0) -- a mask read
  // from the full BN output (1) stays with the general kernel
  if ((p.flags & kIgemmBnFold) && (!bnb || cat || !(p.bn_relu == 0 || p.bn_relu == 3))) return false;
  if (bnb && (p.stats == nullptr || p.bn_x == nullptr || p.bn_coef == nullptr || p.ldc != p.N ||
              !(p.bn_relu == 0 || p.bn_relu == 2 || p.bn_relu == 6 || (p.bn_relu == 3 && p.bn_z != nullptr))))
    return false;
  if (p.M <= 0 || p.N % 64 || p.Cred % 32 || p.lda % 8 || p.ldb % 8 || p.ldc % 8) return false;
  const long ohow = p.div_ohow.d, ow = p.div_ow.d, oh = ohow / (ow > 0 ? ow : 1);
  // every output pixel reads pixel (oy * SY, ox * SX): must exist
  if ((oh - 1) * p.SY >= p.IH || (ow - 1) * p.SX >= p.IW) return false;
  const bool strided = !(p.SY == 1 && p.SX == 1 && (long)p.IH * p.IW == ohow && p.IW == ow);
  if (strided && (p.R != nullptr || bnb || p.bias != nullptr || cat)) return false;
  if ((p.flags & kIgemmAxf) && (p.stats == nullptr || p.R != nullptr || bnb || p.bias != nullptr || strided ||
                                p.a_scale == nullptr || p.a_shift == nullptr || p.Cred > 2048))
    return false;
  if (p.flags & kIgemmRStrided) {      // compact R of a strided 1x1 branch (see the kernel's prefetch)
    auto pow2 = [](int v) { return v >= 1 && (v & (v - 1)) == 0; };
    if (p.R == nullptr || p.r_mask != nullptr || strided || !pow2(p.OSY) || !pow2(p.OSX) ||
        p.OHF != (oh + p.OSY - 1) / p.OSY || p.OWF != (ow + p.OSX - 1) / p.OSX)
      return false;
  }
  // 32-bit byte offsets inside one tile: the images 128 output rows can touch, the weight panel, the output rows
  const double images = 130.0 / (double)(ohow > 0 ? ohow : 1) + 2.0;
  if (images * p.IH * p.IW * p.lda * 2.0 >= 4294967296.0) return false;
  if ((double)p.N * p.ldb * 2.0 >= 4294967296.0 || 128.0 * p.ldc * 2.0 >= 4294967296.0) return false;
  if ((double)p.M * (strided ? (double)p.SY * p.SX : 1.0) >= 2147483648.0) return false;
  IgemmParams q = p;
  if (q.stats_ld == 0) q.stats_ld = q.N;
  if (dtype == kBF16) launch_lean_tiles<bf16_t>(q, strided, stream);
  else launch_lean_tiles<f16_t>(q, strided, stream);
  return true;
}

}  // namespace dpa
