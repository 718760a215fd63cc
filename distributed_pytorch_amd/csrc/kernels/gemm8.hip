// 256 x 256 multi-phase MFMA GEMM for gfx950:  C[m][n] = sum_k A[m][k] * B[n][k]  (+ bias[n]) (+ ReLU)
//
// The 128 x 128 one-barrier-per-K-tile kernels (igemm.hip, gemm_lean.hip) plateau at 0.8-0.9 PFLOP/s: every K tile
// they stop at a barrier with nothing in flight that the barrier does not also wait for.  This kernel follows the CDNA
// guide's deep-pipelined structure instead (cdna_hip_programming.md, "The 256^2 8-phase template"), re-derived for
// 32x32x16 MFMAs and this project's operand layouts:
//
//  * 8 waves (2 along M x 4 along N), one workgroup per CU, a wave owns 128 x 64 of the tile in four 64 x 32 quadrants;
//  * operands go global -> LDS with global_load_lds_dwordx4 (no registers), in HALF-tiles of 128 rows x 64 k (16 KiB,
//    two wave-instructions per wave); the LDS is a ring of 8 half-tile slots = two K tiles, ONE __shared__ object;
//  * a K tile is four phases, one quadrant each: [fragment reads (inline-asm ds_read_b128) | request of the half-tile
//    5 phases ahead] barrier [lgkmcnt(0), 8 MFMAs] barrier.  The four waves with wr = 1 run one barrier behind the
//    other four, so on every SIMD one wave reads LDS while its partner issues MFMAs;
//  * the only VMEM wait is a COUNTED s_waitcnt vmcnt(2) in the last phase of a K tile (never 0 in steady state): the
//    half-tile requested in that phase stays in flight across the barriers.
//
// Ordering argument (phases p = 4t + q; half-tile g is requested in phase g - 5 into slot g mod 8, order inside a K
// tile: B cols 0..127, B cols 128..255, A rows 0..127, A rows 128..255):
//  * read after write: each wave waits for its own requests of K tile t+1 before the first barrier of phase 4t+3;
//    the staggered group passes that point one barrier later; the first read of tile t+1 comes two barriers later.
//  * write after read: the request issued in phase p overwrites half-tile p - 3.  B half-tiles are read in phases 4t
//    and 4t+1 (both 32-column fragments of the second quadrant pair stay in registers), A half-tiles in 4t and 4t+2:
//    half-tile g is last read in phase <= g + 1 <= p - 2, retired by lgkmcnt(0) and followed by at least two
//    barriers on either wave group before the overwriting request is issued.
//
// Operand rows are stored in LDS with the 16-byte k-chunk XORed by (row >> 1) & 7 (the source address is permuted, the
// DMA writes linearly, the fragment read applies the same involution): 16 consecutive fragment rows touch 16 distinct
// 16-byte columns of the 256-byte bank row.
#include "../common.h"
#include "kernels.h"
#include <stdexcept>

namespace dpa {
namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr8;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr8;

__device__ uint4 g_gemm8_zero[4];      // source of rows past M / N (never written: zero-initialised)

constexpr int kHalf = 128 * 128;       // bytes of a half-tile: 128 rows x 64 k x 2 B
constexpr int kRing = 8 * kHalf;       // 128 KiB

template <int OFF> __device__ __forceinline__ uint4 lds_read16(uint32_t addr) {
  uint4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

template <typename T> struct Mma8;
template <> struct Mma8<bf16_t> {
  static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    bf16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
};
template <> struct Mma8<f16_t> {
  static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    f16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  }
};

struct Gemm8Params {
  const void* A; const void* B; void* C; const float* bias;
  int M, N, K; long lda, ldb, ldc; int relu;
};

template <typename T>
__global__ __launch_bounds__(512, 1) void gemm8_kernel(const Gemm8Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform by construction: keep it in an SGPR
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, lh = lane >> 5;

  const uint32_t tiles_n = (p.N + 255) / 256;
  const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
  // grouped order: 8 M tiles x every N tile run back to back on one XCD (xcd_remap hands each XCD a contiguous range of
  // lid), so the 8 A panels stay in that XCD's L2 while the B panels stream past once per group
  const uint32_t tiles_m = (p.M + 255) / 256;
  const uint32_t per_group = 8 * tiles_n;
  const uint32_t grp = lid / per_group, in_grp = lid - grp * per_group;
  const uint32_t first_m = grp * 8;
  const uint32_t gm = min(8u, tiles_m - first_m);
  const int mt0 = (int)(first_m + in_grp % gm), nt0 = (int)(in_grp / gm);
  const int m0 = mt0 * 256, n0 = nt0 * 256;
  const int nk = p.K / 64;
  const int nhalf = 4 * nk;

  // ---- staging: wave-instruction i (0, 1) of this wave covers rows (wave * 2 + i) * 8 + lane / 8 of a half-tile; the
  // lane's LDS slot is chunk lane % 8 of that row and holds source chunk (lane % 8) ^ ((row >> 1) & 7)
  const char* zero = reinterpret_cast<const char*>(g_gemm8_zero);
  const char* src[4][2];          // [half-tile kind: B0, B1, A0, A1][instruction]
  int step[4][2];                 // bytes per K tile (0 for rows that read the zero page)
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave * 2 + i) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const bool isA = h >= 2;
      const long grow = (isA ? m0 : n0) + (h & 1) * 128 + row;
      const bool ok = grow < (isA ? p.M : p.N);
      const char* base = reinterpret_cast<const char*>(isA ? p.A : p.B);
      src[h][i] = ok ? base + (grow * (isA ? p.lda : p.ldb) + chunk * 8) * (long)sizeof(T) : zero;
      step[h][i] = ok ? 128 : 0;
    }
  // request half-tile g (kind g & 3 of K tile g >> 2) into slot g & 7
  auto stage = [&](int g, auto kind_c) {
    constexpr int h = decltype(kind_c)::value;
    unsigned char* dst = smem + (g & 7) * kHalf + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((gbl_void_ptr8)src[h][i], (lds_void_ptr8)(dst + i * 1024), 16, 0, 0);
      src[h][i] += step[h][i];
    }
  };
  constexpr std::integral_constant<int, 0> kB0{};
  constexpr std::integral_constant<int, 1> kB1{};
  constexpr std::integral_constant<int, 2> kA0{};
  constexpr std::integral_constant<int, 3> kA1{};

  // ---- fragment addresses (bytes from the start of the ring, K-tile parity added per tile)
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
  // A: half-tile kind 2 + wr, rows mq * 64 + mt * 32 + l31; B: kind wc >> 1, rows (wc & 1) * 64 + nq * 32 + l31.  The
  // swizzle term (row >> 1) & 7 has period 16 rows, so it only depends on l31 and the quadrant / tile offsets are
  // immediates of the ds_read: A + mq * 8192 + mt * 4096, B + nq * 4096.
  uint32_t cA[4], cB[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const uint32_t c = (uint32_t)(((ks * 2 + lh) ^ ((l31 >> 1) & 7)) * 16) + (uint32_t)l31 * 128;
    cA[ks] = lds0 + (2 + wr) * kHalf + c;
    cB[ks] = lds0 + (wc >> 1) * kHalf + (wc & 1) * 8192 + c;
  }

  f32x16_t acc[2][2][2];                  // [mq][mt][nq]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;

  // ---- prologue: half-tiles 0..4, K tile 0 complete (one half-tile may stay in flight)
  stage(0, kB0);
  if (1 < nhalf) stage(1, kB1);
  if (2 < nhalf) stage(2, kA0);
  if (3 < nhalf) stage(3, kA1);
  if (4 < nhalf) {
    stage(4, kB0);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();       // this group runs one barrier behind

  uint4 fa[2][4], fb[2][4];               // A fragments of the current mq [mt][ks]; B fragments [nq][ks]
  for (int t = 0; t < nk; ++t) {
    const uint32_t tb = (uint32_t)(t & 1) * (4 * kHalf);
    const int g = 4 * t + 5;              // half-tile requested in phase 0 of this K tile
    // ---------------- phase 0: quadrant (m 0, n 0)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[0][ks] = lds_read16<0>(cB[ks] + tb);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fa[0][ks] = lds_read16<0>(cA[ks] + tb);
      fa[1][ks] = lds_read16<4096>(cA[ks] + tb);
    }
    if (g < nhalf) stage(g, kB1);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Mma8<T>::run(acc[0][0][0], fb[0][ks], fa[0][ks]);
      Mma8<T>::run(acc[0][1][0], fb[0][ks], fa[1][ks]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 1: quadrant (m 0, n 1)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[1][ks] = lds_read16<4096>(cB[ks] + tb);
    if (g + 1 < nhalf) stage(g + 1, kA0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Mma8<T>::run(acc[0][0][1], fb[1][ks], fa[0][ks]);
      Mma8<T>::run(acc[0][1][1], fb[1][ks], fa[1][ks]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 2: quadrant (m 1, n 1)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fa[0][ks] = lds_read16<8192>(cA[ks] + tb);
      fa[1][ks] = lds_read16<12288>(cA[ks] + tb);
    }
    if (g + 2 < nhalf) stage(g + 2, kA1);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Mma8<T>::run(acc[1][0][1], fb[1][ks], fa[0][ks]);
      Mma8<T>::run(acc[1][1][1], fb[1][ks], fa[1][ks]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 3: quadrant (m 1, n 0): operands are in registers; K tile t + 1 must have landed
    if (g + 3 < nhalf) {
      stage(g + 3, kB0);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Mma8<T>::run(acc[1][0][0], fb[0][ks], fa[0][ks]);
      Mma8<T>::run(acc[1][1][0], fb[0][ks], fa[1][ks]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();       // balance the stagger

  // ---- epilogue: acc[mq][mt][nq][r] is C[m][n] with m = wr*128 + mq*64 + mt*32 + l31,
  //      n = wc*64 + nq*32 + (r & 3) + 8 * (r >> 2) + 4 * lh.  The ring is free now (every wave is past its last read:
  //      the barriers above); a wave parks its 128 x 64 outputs in ITS OWN 16 KiB of it ([row][8 chunks of 16 bytes],
  //      chunk XOR row & 7 against the 32-rows-one-column write pattern) and stores whole 128-byte row segments.
  unsigned char* my = smem + wave * 16384;
  float bv[2][4][4];
#pragma unroll
  for (int nq = 0; nq < 2; ++nq)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
      {
        // (one 16-byte load where the four channels exist: per-element selects made these 32 dword loads per thread)
        const int n = n0 + wc * 64 + nq * 32 + 8 * gq + 4 * lh;
        if (p.bias != nullptr && n + 3 < p.N) {
          const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
          bv[nq][gq][0] = b4.x; bv[nq][gq][1] = b4.y; bv[nq][gq][2] = b4.z; bv[nq][gq][3] = b4.w;
        } else {
      #pragma unroll
          for (int j = 0; j < 4; ++j) bv[nq][gq][j] = (p.bias != nullptr && n + j < p.N) ? p.bias[n + j] : 0.f;
        }
      }
#pragma unroll
  for (int mq = 0; mq < 2; ++mq)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = mq * 64 + mt * 32 + l31;
#pragma unroll
      for (int nq = 0; nq < 2; ++nq)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = acc[mq][mt][nq][gq * 4 + j] + bv[nq][gq][j];
            if (p.relu) v[j] = fmaxf(v[j], 0.f);
          }
          const T t0 = from_float<T>(v[0]), t1 = from_float<T>(v[1]), t2 = from_float<T>(v[2]), t3 = from_float<T>(v[3]);
          uint16_t b0, b1, b2, b3;
          __builtin_memcpy(&b0, &t0, 2); __builtin_memcpy(&b1, &t1, 2);
          __builtin_memcpy(&b2, &t2, 2); __builtin_memcpy(&b3, &t3, 2);
          uint2 o;
          o.x = (uint32_t)b0 | ((uint32_t)b1 << 16);
          o.y = (uint32_t)b2 | ((uint32_t)b3 << 16);
          const int col = nq * 32 + 8 * gq + 4 * lh;              // first of 4 columns: 8 bytes inside chunk col / 8
          *reinterpret_cast<uint2*>(my + row * 128 + (((col >> 3) ^ (row & 7)) * 16) + (col & 7) * 2) = o;
        }
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  T* Cg = reinterpret_cast<T*>(p.C);
  const int nbase = n0 + wc * 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = i * 8 + (lane >> 3), ck = lane & 7;
    const long m = (long)m0 + wr * 128 + row;
    const int n = nbase + ck * 8;
    const uint4 v = *reinterpret_cast<const uint4*>(my + row * 128 + ((ck ^ (row & 7)) * 16));
    if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(Cg + m * p.ldc + n) = v;      // N % 8 == 0: whole chunks
  }
}

}  // namespace

// C[M][N] = A[M][K] * B[N][K]^T (+ bias) (+ ReLU); 16-bit types, K a multiple of 64, N of 8, 16-byte aligned rows
void gemm8_launch(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda, long ldb,
                  long ldc, int relu, int dtype, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return;
  if (dtype == kF32) throw std::invalid_argument("gemm8: 16-bit types only");
  if (K % 64 != 0 || N % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0)
    throw std::invalid_argument("gemm8: K % 64, N % 8 and 16-byte aligned operand rows required");
  Gemm8Params p{A, B, C, bias, M, N, K, lda, ldb, ldc, relu};
  const long tiles = (long)ceil_div(M, 256) * ceil_div(N, 256);
  static bool attr_done = false;
  if (!attr_done) {
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_kernel<bf16_t>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kRing));
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_kernel<f16_t>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kRing));
    attr_done = true;
  }
  if (dtype == kBF16)
    hipLaunchKernelGGL(gemm8_kernel<bf16_t>, dim3((unsigned)tiles), dim3(512), kRing, stream, p);
  else
    hipLaunchKernelGGL(gemm8_kernel<f16_t>, dim3((unsigned)tiles), dim3(512), kRing, stream, p);
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
