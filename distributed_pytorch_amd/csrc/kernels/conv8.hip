// Convolutions on the 256 x 256 multi-phase MFMA pipeline of gemm8.hip: an implicit-GEMM front end, the fused
// epilogues the training step needs, and a stream-K schedule.
//
//     C[m][n] = sum_{tap t, channel c} A[pix(m) + (dy_t, dx_t)][c] * B[n][wtap_t * Cred + c]
//
// for stride-1 "same" problems (the output grid IS the gathered image: 1x1 convolutions, 3x3 / 5x5 / 7x7 with
// padding, their stride-1 data gradients, Linear layers), 16-bit operands, fp32 accumulation.
//
//  * K loop: gemm8.hip's structure unchanged -- 8 waves, a K tile of 64 channels in four phases of one 128 x 64
//    quadrant each, operands DMA'd into an 8-slot LDS ring five phases ahead, ONE counted vmcnt wait per K tile, the
//    two wave groups one barrier apart.  The ordering argument is in gemm8.hip.
//  * Operand addressing: `buffer_load_dwordx4 ... lds`.  A lane keeps ONE 32-bit offset per staged row for the whole
//    tile; the K position (tap and channel block) is a scalar offset shared by the workgroup, so advancing the K loop
//    costs no vector instruction.  Convolution padding, rows past M and weight rows past N are lanes whose offset
//    lies outside the buffer resource: the hardware range check returns zeros for them (no zero page, no select on
//    a 64-bit pointer).  Which taps a staged pixel can see is a per-row bit mask computed once per tile.
//  * Epilogues (a wave parks its 128 x 64 outputs in ITS OWN 16 KiB of the free ring and streams them out as whole
//    128-byte row pieces -- no barrier between the waves): bias / ReLU; the Chan partials of the following
//    BatchNorm's statistics (slab rows of 128 pixels, the format of igemm.hip, so bn_fold_finalize is unchanged);
//    the fused BatchNorm-backward reduction of data-gradient launches (kIgemmBnBwd, ReLU mask as bits or recomputed);
//    the masked store of kIgemmXMask.  The extra operand of the last two is requested before the accumulators go
//    through LDS.
//  * Schedule: ResNet-50's deep layers at batch 1024 are 784 and 392 tiles -- 3.06 and 1.53 rounds of 256 CUs, a
//    quarter of the chip-time idle in the last round.  The stream-K launch cuts the tiles x K-tiles iteration space
//    into as many equal ranges as there are CUs, cuts the ranges at tile boundaries, and runs every resulting SEGMENT
//    as a workgroup of its own, longest first (the dispatcher hands the next segment to the first CU that frees up,
//    so every CU ends up with about one range of work).  A segment that is not a whole tile writes its fp32
//    accumulators to a slab and takes a ticket on the tile's counter; the LAST arriver adds the slabs of the others and
//    runs the epilogue -- release / acquire at agent scope, nobody ever waits for another workgroup, the counter is
//    reset by the last arriver (graph-replay safe).  (A persistent loop over segments inside one workgroup, the
//    textbook form, cost 300-500 spilled registers inside the K loop with this compiler and ran at half the speed.)
//
// Replaces what cudnn.benchmark picks for these layers in the reference (imagenet_ddp_apex.py:110, nd_imagenet.py:291,302)
// -- SURVEY.md N-K1, N-K2.
#include "../common.h"
#include "kernels.h"
#include <algorithm>
#include <map>
#include <mutex>
#include <stdexcept>
#include <tuple>
#include <vector>

namespace dpa {
namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr9;

constexpr int kHalf9 = 128 * 128;      // bytes of a half-tile: 128 rows x 64 k x 2 B
constexpr int kRing9 = 8 * kHalf9;     // 128 KiB
constexpr uint32_t kOob = 0xffffffffu; // a lane offset past every buffer resource: the load returns zeros
constexpr int kMaxTaps = 32;
constexpr int kReach = 3;              // |dy|, |dx| <= 3 (7 x 7 filters)

template <int OFF> __device__ __forceinline__ uint4 lds_read16_9(uint32_t addr) {
  uint4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

template <typename T> struct Mma9;
template <> struct Mma9<bf16_t> {
  static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    bf16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
};
template <> struct Mma9<f16_t> {
  static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    f16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  }
};

struct Conv8Seg {       // one workgroup of a stream-K launch
  int tile, kb, ke;     // K tiles [kb, ke) of output tile `tile`
  int first, pieces;    // slab slots first .. first + pieces - 1 hold the tile's partial segments (pieces == 1: whole tile)
  int mine;             // this segment's slot among them
};

struct Conv8Extra {
  int nk;            // K tiles per output tile = ntaps * kpt
  int kpt;           // K tiles per tap = Cred / 64
  int tiles_m, tiles_n;
  int parts;         // ceil(M / 128): rows of the statistics slab
  int shift_pix;     // kReach * IW + kReach: the activation resource starts this many pixels before the tile
  const Conv8Seg* segs;  // stream-K: one entry per workgroup (nullptr: workgroup b computes tile b)
  float* slab;       // partial accumulators [slot][32][512] float4
  int* counters;     // [tiles], zero between launches
};

enum Conv8Epi : int { kEpiPlain = 0, kEpiStats = 1, kEpiBnb = 2, kEpiXMask = 3 };

// SK: stream-K launch (a workgroup walks several segments and hands partial tiles over); !SK: one whole tile per
// workgroup, nothing of the hand-off compiled in (the allocator then fits the kernel in 256 registers without spills).
template <typename T, int EPI, bool SK>
__global__ __launch_bounds__(512, 1) void conv8_kernel(const IgemmParams p, const Conv8Extra e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const int ntaps = min(p.ntaps, kMaxTaps);
  int tile, kb, ke, seg_first = 0, seg_pieces = 1, seg_mine = 0;
  if constexpr (SK) {
    const Conv8Seg sg = e.segs[blockIdx.x];
    tile = __builtin_amdgcn_readfirstlane(sg.tile);
    kb = __builtin_amdgcn_readfirstlane(sg.kb);
    ke = __builtin_amdgcn_readfirstlane(sg.ke);
    seg_first = __builtin_amdgcn_readfirstlane(sg.first);
    seg_pieces = __builtin_amdgcn_readfirstlane(sg.pieces);
    seg_mine = __builtin_amdgcn_readfirstlane(sg.mine);
  } else {
    tile = (int)xcd_remap(blockIdx.x, gridDim.x);
    kb = 0;
    ke = e.nk;
  }
  {
    // ---- fragment addresses (bytes from the start of the ring; gemm8.hip)
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
    uint32_t cA[4], cB[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint32_t c = (uint32_t)(((ks * 2 + lh) ^ ((l31 >> 1) & 7)) * 16) + (uint32_t)l31 * 128;
      cA[ks] = lds0 + (2 + wr) * kHalf9 + c;
      cB[ks] = lds0 + (wc >> 1) * kHalf9 + (wc & 1) * 8192 + c;
    }
    const int srow0 = (wave * 2) * 8 + (lane >> 3);            // staged row of instruction 0 (instruction 1: + 8)
    const int mt = tile / e.tiles_n, nt = tile - mt * e.tiles_n;
    const int m0 = mt * 256, n0 = nt * 256;
    const int nks = ke - kb;
    const int nhalf = 4 * nks;

    // ---- buffer resources of this tile: 2 GiB windows, lanes address them with 32-bit offsets
    const char* abase = reinterpret_cast<const char*>(p.A) + ((long)m0 - e.shift_pix) * p.lda * (long)sizeof(T);
    const char* bbase = reinterpret_cast<const char*>(p.B) + (long)n0 * p.ldb * (long)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(abase), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(bbase), 0, 0x7fffffff, 0x00020000);

    // ---- per-lane staging state: offsets of the 2 x 2 rows of either operand, tap visibility of the A rows
    uint32_t voffA[2][2], vmaskA[2][2], voffB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = srow0 + i * 8;
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int lr = h * 128 + row;
        voffA[h][i] = (uint32_t)((lr * p.lda + chunk * 8) * (int)sizeof(T));
        const int n = n0 + lr;
        voffB[h][i] = n < p.N ? (uint32_t)((lr * p.ldb + chunk * 8) * (int)sizeof(T)) : kOob;
        vmaskA[h][i] = 0u;
      }
    {
      int ry[2][2], rx[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint32_t m = (uint32_t)(m0 + h * 128 + srow0 + i * 8);
          if (m < (uint32_t)p.M) {
            const uint32_t img = fdiv(m, p.div_ohow);
            const uint32_t rem = m - img * p.div_ohow.d;
            const uint32_t y = fdiv(rem, p.div_ow);
            ry[h][i] = (int)y;
            rx[h][i] = (int)(rem - y * p.div_ow.d);
          } else {
            ry[h][i] = -1000;            // no tap reaches an image from here
            rx[h][i] = -1000;
          }
        }
      for (int t = 0; t < ntaps; ++t) {
        const int dy = __builtin_amdgcn_readfirstlane(p.taps[3 * t]);
        const int dx = __builtin_amdgcn_readfirstlane(p.taps[3 * t + 1]);
        const bool reach = dy >= -kReach && dy <= kReach && dx >= -kReach && dx <= kReach;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const bool ok = reach && (unsigned)(ry[h][i] + dy) < (unsigned)p.IH && (unsigned)(rx[h][i] + dx) < (unsigned)p.IW;
            vmaskA[h][i] |= ok ? (1u << t) : 0u;
          }
      }
    }

    // ---- scalar K cursors: tap index, channel block inside the tap, scalar byte offset of the K tile
    int a_ti = kb / e.kpt, a_kc = kb - a_ti * e.kpt;
    int b_ti = a_ti, b_kc = a_kc;
    auto tap_a = [&](int t) -> uint32_t {
      const int tt = t < ntaps ? t : 0;
      const int dy = __builtin_amdgcn_readfirstlane(p.taps[3 * tt]);
      const int dx = __builtin_amdgcn_readfirstlane(p.taps[3 * tt + 1]);
      return (uint32_t)((dy * p.IW + dx + e.shift_pix) * p.lda * (int)sizeof(T));
    };
    auto tap_b = [&](int t) -> uint32_t {
      const int tt = t < ntaps ? t : 0;
      return (uint32_t)(__builtin_amdgcn_readfirstlane(p.taps[3 * tt + 2]) * p.Cred * (int)sizeof(T));
    };
    uint32_t soffA = tap_a(a_ti) + (uint32_t)a_kc * 128u;
    uint32_t soffB = tap_b(b_ti) + (uint32_t)b_kc * 128u;
    uint32_t vselA[2][2];
    auto select_a = [&]() {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) vselA[h][i] = ((vmaskA[h][i] >> a_ti) & 1u) ? voffA[h][i] : kOob;
    };
    select_a();
    auto advance_a = [&]() {
      if (++a_kc == e.kpt) {
        a_kc = 0;
        ++a_ti;
        soffA = tap_a(a_ti);
      } else {
        soffA += 128u;
      }
      select_a();
    };
    auto advance_b = [&]() {
      if (++b_kc == e.kpt) {
        b_kc = 0;
        ++b_ti;
        soffB = tap_b(b_ti);
      } else {
        soffB += 128u;
      }
    };
    // request half-tile g (kind g & 3: B cols 0..127, B cols 128..255, A rows 0..127, A rows 128..255) into slot g & 7
    auto stage = [&](int g, auto kind_c) {
      constexpr int kind = decltype(kind_c)::value;
      unsigned char* dst = smem + (g & 7) * kHalf9 + wave * 2048;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (kind < 2)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_ptr9)(dst + i * 1024), 16, voffB[kind][i], soffB, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_ptr9)(dst + i * 1024), 16, vselA[kind - 2][i], soffA, 0, 0);
      }
    };
    constexpr std::integral_constant<int, 0> kB0{};
    constexpr std::integral_constant<int, 1> kB1{};
    constexpr std::integral_constant<int, 2> kA0{};
    constexpr std::integral_constant<int, 3> kA1{};

    f32x16_t acc[2][2][2];                  // [mq][mt][nq]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;

    // ---- prologue: half-tiles 0..4 (K tile 0 complete, one half-tile may stay in flight)
    stage(0, kB0);
    stage(1, kB1);
    advance_b();
    stage(2, kA0);
    stage(3, kA1);
    advance_a();
    if (4 < nhalf) {
      stage(4, kB0);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();       // this group runs one barrier behind

    uint4 fa[2][4], fb[2][4];
    for (int t = 0; t < nks; ++t) {
      const uint32_t tb = (uint32_t)(t & 1) * (4 * kHalf9);
      const int g = 4 * t + 5;
      // ---------------- phase 0: quadrant (m 0, n 0)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fb[0][ks] = lds_read16_9<0>(cB[ks] + tb);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fa[0][ks] = lds_read16_9<0>(cA[ks] + tb);
        fa[1][ks] = lds_read16_9<4096>(cA[ks] + tb);
      }
      if (g < nhalf) stage(g, kB1);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      advance_b();                          // K tile t + 2 for the B requests of phase 3 / next phase 0
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Mma9<T>::run(acc[0][0][0], fb[0][ks], fa[0][ks]);
        Mma9<T>::run(acc[0][1][0], fb[0][ks], fa[1][ks]);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      // ---------------- phase 1: quadrant (m 0, n 1)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fb[1][ks] = lds_read16_9<4096>(cB[ks] + tb);
      if (g + 1 < nhalf) stage(g + 1, kA0);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Mma9<T>::run(acc[0][0][1], fb[1][ks], fa[0][ks]);
        Mma9<T>::run(acc[0][1][1], fb[1][ks], fa[1][ks]);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      // ---------------- phase 2: quadrant (m 1, n 1)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fa[0][ks] = lds_read16_9<8192>(cA[ks] + tb);
        fa[1][ks] = lds_read16_9<12288>(cA[ks] + tb);
      }
      if (g + 2 < nhalf) stage(g + 2, kA1);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      advance_a();                          // K tile t + 2 for the A requests of the next phases 1 / 2
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Mma9<T>::run(acc[1][0][1], fb[1][ks], fa[0][ks]);
        Mma9<T>::run(acc[1][1][1], fb[1][ks], fa[1][ks]);
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
        Mma9<T>::run(acc[1][0][0], fb[0][ks], fa[0][ks]);
        Mma9<T>::run(acc[1][1][0], fb[0][ks], fa[1][ks]);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();       // balance the stagger: every wave is past its last ring read

    // ---- stream-K hand-off -------------------------------------------------------------------------------------
    // acc registers travel in a lane-private layout: float4 q of thread tid at slab[slot][q][tid] (coalesced; a
    // uniform base per access + one 32-bit lane offset: no per-access address registers)
    if (SK && seg_pieces > 1) {
      const uint32_t toff = (uint32_t)tid * 16u;
      {
        char* const out = reinterpret_cast<char*>(e.slab) + (size_t)(seg_first + seg_mine) * (32 * 512 * 16);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x16_t& v = acc[a][b][c];
                *reinterpret_cast<float4*>(out + (size_t)(((a * 2 + b) * 2 + c) * 4 + q) * 8192 + toff) =
                    make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
              }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int* const ticket_box = reinterpret_cast<int*>(smem);      // the ring is free (one __shared__ object: gemm8.hip)
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = __hip_atomic_fetch_add(e.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == seg_pieces - 1) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(e.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
        }
        *ticket_box = ticket;
      }
      __syncthreads();
      const int ticket = __builtin_amdgcn_readfirstlane(*ticket_box);
      if (ticket != seg_pieces - 1) return;          // (uniform) somebody else finishes this tile
      __syncthreads();                               // every wave has read the box before the ring is reused
      // fixed summation order whoever arrives last (its own part is read back like the others): bit-identical runs
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;
      for (int j = 0; j < seg_pieces; ++j) {
        const char* const in = reinterpret_cast<const char*>(e.slab) + (size_t)(seg_first + j) * (32 * 512 * 16);
        // two batches of 16 loads (64 registers, the dead fragment registers): everything at once would spill
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          float4 v[2][2][4];
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int q = 0; q < 4; ++q)
                v[b][c][q] = *reinterpret_cast<const float4*>(in + (size_t)(((a * 2 + b) * 2 + c) * 4 + q) * 8192 + toff);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                acc[a][b][c][4 * q] += v[b][c][q].x;
                acc[a][b][c][4 * q + 1] += v[b][c][q].y;
                acc[a][b][c][4 * q + 2] += v[b][c][q].z;
                acc[a][b][c][4 * q + 3] += v[b][c][q].w;
              }
          asm volatile("" ::: "memory");
        }
      }
    }

    // ---- epilogue: acc[mq][mt][nq][r] is C[m][n] with m = wr*128 + mq*64 + mt*32 + l31,
    //      n = wc*64 + nq*32 + (r & 3) + 8 * (r >> 2) + 4 * lh; parked in the wave's own 16 KiB as [row][8 chunks],
    //      chunk XOR row & 7; then lane (row = i*8 + lane/8, chunk = lane%8) streams 16-byte pieces, i = 0..15
    unsigned char* my = smem + wave * 16384;
    const int mbase = m0 + wr * 128;                 // first row of this wave's part
    const int rows = min(128, p.M - mbase);           // <= 0: nothing of this part exists
    const int ck = lane & 7, rsub = lane >> 3;
    const int ncol = n0 + wc * 64 + ck * 8;           // first of this lane's 8 channels
    const bool col_ok = ncol < p.N;                   // N % 8 == 0: whole chunks
    // global addresses = a wave-uniform base (+ a uniform row-group step) + ONE 32-bit lane offset
    // (a part that does not exist -- rows past M, or a 64-column block past N -- requests its epilogue operands at offset
    // 0 of their tensors: its lanes are all "not ok" and would otherwise read up to 255 rows past the end)
    const bool part_ok = rows > 0 && n0 + wc * 64 < p.N;      // wave uniform
    const size_t cb = part_ok ? ((size_t)mbase * p.ldc + (size_t)(n0 + wc * 64)) * sizeof(T) : (size_t)0;
    const uint32_t loff = (uint32_t)((rsub * p.ldc + ck * 8) * (int)sizeof(T));
    const uint32_t rstep = 8u * (uint32_t)p.ldc * (uint32_t)sizeof(T);
    uint4 px[16];
    uint32_t pbits[16];
    // epilogue operands (BN input / ReLU source, mask bits) are requested in two halves: rows 0..63 before the accumulators
    // go through LDS, rows 64..127 behind that -- all 16 row pieces next to the live accumulators spilled 43 VGPRs
    auto request = [&](int first, int last) __attribute__((always_inline)) {
      if constexpr (EPI == kEpiBnb || EPI == kEpiXMask) {
        const char* xg = reinterpret_cast<const char*>(p.bn_x) + cb;
        const uint8_t* bits = reinterpret_cast<const uint8_t*>(p.bn_z) + (cb >> 4);
        const bool want_bits = EPI == kEpiBnb && (p.bn_relu & 3) == 3;
        const bool fold = EPI == kEpiBnb && (p.flags & kIgemmBnFold);     // bn_x is not read (kernels.h)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i < first || i >= last) continue;
          const bool ok = col_ok && i * 8 + rsub < rows;
          const uint32_t lo = ok ? loff : 0u;              // (row 0 of an existing part always exists)
          const size_t ro = ok ? (size_t)i * rstep : 0;
          pbits[i] = want_bits ? bits[(ro >> 4) + (lo >> 4)] : 0xffu;
        }
        if (!fold) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (i < first || i >= last) continue;
            const bool ok = col_ok && i * 8 + rsub < rows;
            px[i] = *reinterpret_cast<const uint4*>(xg + (ok ? (size_t)i * rstep : 0) + (ok ? loff : 0u));
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (i < first || i >= last) continue;
            px[i] = make_uint4(0u, 0u, 0u, 0u);
          }
        }
      }
    };
    // per-channel BN coefficients FIRST: vmcnt retires in order, requested behind the row operands they would make row 0
    // wait for the last request (profiles/r3/lean_epilogue_vmcnt.md)
    float bmu[8], bis[8], bsc[8], bsh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bmu[j] = bis[j] = bsc[j] = bsh[j] = 0.f;
    int bn_mode = 0;
    bool bn_clip = false;
    if constexpr (EPI == kEpiBnb) {
      bn_mode = p.bn_relu & 3;
      bn_clip = p.bn_relu & 4;
      if (!(p.flags & kIgemmBnFold)) {
        const float* cf = p.bn_coef + (col_ok ? ncol : 0);      // [4][stats_ld]: mean, invstd, scale, shift
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bmu[j] = cf[j];
          bis[j] = cf[p.stats_ld + j];
          bsc[j] = bn_mode == 2 ? cf[2 * (long)p.stats_ld + j] : 0.f;
          bsh[j] = bn_mode == 2 ? cf[3 * (long)p.stats_ld + j] : 0.f;
        }
      }
    }
    request(0, 8);
    float bv[2][4][4];
    if constexpr (EPI == kEpiPlain) {
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
    }
    const bool relu = EPI == kEpiPlain && (p.flags & kIgemmRelu);
#pragma unroll
    for (int mq = 0; mq < 2; ++mq)
#pragma unroll
      for (int mt2 = 0; mt2 < 2; ++mt2) {
        const int row = mq * 64 + mt2 * 32 + l31;
#pragma unroll
        for (int nq = 0; nq < 2; ++nq)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = acc[mq][mt2][nq][gq * 4 + j];
              if constexpr (EPI == kEpiPlain) {
                v[j] += bv[nq][gq][j];
                if (relu) v[j] = fmaxf(v[j], 0.f);
              }
            }
            const T t0 = from_float<T>(v[0]), t1 = from_float<T>(v[1]), t2 = from_float<T>(v[2]), t3 = from_float<T>(v[3]);
            uint16_t b0, b1, b2, b3;
            __builtin_memcpy(&b0, &t0, 2); __builtin_memcpy(&b1, &t1, 2);
            __builtin_memcpy(&b2, &t2, 2); __builtin_memcpy(&b3, &t3, 2);
            uint2 o;
            o.x = (uint32_t)b0 | ((uint32_t)b1 << 16);
            o.y = (uint32_t)b2 | ((uint32_t)b3 << 16);
            const int col = nq * 32 + 8 * gq + 4 * lh;
            *reinterpret_cast<uint2*>(my + row * 128 + (((col >> 3) ^ (row & 7)) * 16) + (col & 7) * 2) = o;
          }
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    request(8, 16);

    float s1[8], s2[8], kshift[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = kshift[j] = 0.f;
    if constexpr (EPI == kEpiStats) {
      Vec16<T> k0;                                  // the part's first row: a sample, |v - k| ~ std (igemm.hip)
      k0.load(my + ck * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) kshift[j] = k0.get(j);
    }
    char* cg = reinterpret_cast<char*>(p.C) + cb;
    const bool nt_store = p.flags & kIgemmNtStore;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = i * 8 + rsub;
      if (row >= rows || !col_ok) continue;
      Vec16<T> v;
      v.load(my + row * 128 + ((ck ^ (row & 7)) * 16));
      if constexpr (EPI == kEpiXMask) {
        Vec16<T> xb;
        xb.raw = px[i];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (!(xb.get(j) > 0.f)) v.set(j, 0.f);
      }
      if constexpr (EPI == kEpiBnb) {
        if ((p.flags & kIgemmBnFold) && bn_mode == 3) {       // the gradient leaves already masked
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (!((pbits[i] >> j) & 1u)) v.set(j, 0.f);
        }
      }
      char* dst = cg + (size_t)i * rstep + loff;
      if (nt_store) v.store_nt(dst);
      else v.store(dst);
      if constexpr (EPI == kEpiStats) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = v.get(j) - kshift[j];
          s1[j] += f;
          s2[j] += f * f;
        }
      }
      if constexpr (EPI == kEpiBnb) {
        Vec16<T> xb;
        xb.raw = px[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float d = v.get(j);                       // as stored: what BatchNorm backward reads back
          const float xv = xb.get(j);
          if (bn_mode == 3 && !((pbits[i] >> j) & 1u)) d = 0.f;
          if (bn_mode == 2) {
            const float yv = to_float(from_float<T>(bsc[j] * xv + bsh[j]));
            if (!(yv > 0.f) || (bn_clip && !(yv < 6.f))) d = 0.f;
          }
          s1[j] += d;
          s2[j] += d * ((xv - bmu[j]) * bis[j]);
        }
      }
    }
    if constexpr (EPI == kEpiStats || EPI == kEpiBnb) {
      // lanes with the same chunk column (lane % 8) hold the same 8 channels: fold the 8 row groups
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int o = 32; o >= 8; o >>= 1) {
          s1[j] += __shfl_xor(s1[j], o, 64);
          s2[j] += __shfl_xor(s2[j], o, 64);
        }
      }
      const int part = mt * 2 + wr;                 // 128-row part of the slab
      if (lane < 8 && col_ok && rows > 0) {
        float* o0 = p.stats + ((size_t)part * 2 + 0) * p.stats_ld + ncol;
        float* o1 = p.stats + ((size_t)part * 2 + 1) * p.stats_ld + ncol;
        const float nr = (float)rows;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if constexpr (EPI == kEpiBnb) {
            o0[j] = s1[j];
            o1[j] = s2[j];
          } else {
            o0[j] = s1[j] + nr * kshift[j];
            o1[j] = fmaxf(s2[j] - s1[j] * s1[j] / nr, 0.f);
          }
        }
      }
      if (EPI == kEpiStats && lane == 0 && wc == 0 && nt == 0 && rows > 0)
        p.stats[(size_t)e.parts * 2 * p.stats_ld + part] = (float)rows;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------
int g_conv8_mode = -1;        // -1 automatic, 0 never, 1 / 2 whenever covered (one tile per workgroup / equal-range stream-K)
int g_conv8_min_k = 512;
int g_conv8_tail_split = 4;    // automatic mode: cut the tiles of a mostly empty last round into at most this many K pieces (0: never)
long g_conv8_launches = 0;
int g_conv8_last[4] = {0, 0, 0, 0};   // (diagnostics) grid, ranges / -pieces, slab slots, CUs of the last launch  // (tests: proof that the launch really took this kernel)      // shortest reduction (elements) the pipeline is worth filling for

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// A stream-K plan: the segment table of one (tiles, K tiles per tile, ranges) combination, built once and kept on the
// device.  `ranges` equal pieces of the iteration space, cut at tile boundaries, longest first.
struct Conv8Plan {
  Conv8Seg* segs = nullptr;
  int grid = 0;          // segments = workgroups
  int slots = 0;         // partial segments = slab slots
};

// segments arrive in iteration order (the pieces of a tile are consecutive): number the slab slots, longest first
std::vector<Conv8Seg> finish_plan(std::vector<Conv8Seg> segs, int* slots_out) {
  int slots = 0;
  for (size_t a = 0; a < segs.size();) {
    size_t b = a;
    while (b < segs.size() && segs[b].tile == segs[a].tile) ++b;
    const int pieces = (int)(b - a);
    for (size_t c = a; c < b; ++c) {
      segs[c].pieces = pieces;
      segs[c].first = pieces > 1 ? slots : 0;
      segs[c].mine = (int)(c - a);
    }
    if (pieces > 1) slots += pieces;
    a = b;
  }
  std::stable_sort(segs.begin(), segs.end(),
                   [](const Conv8Seg& x, const Conv8Seg& y) { return x.ke - x.kb > y.ke - y.kb; });
  *slots_out = slots;
  return segs;
}

std::vector<Conv8Seg> plan_segments(long tiles, int nk, int ranges, int* slots_out) {
  const long iters = tiles * nk;
  const long base = iters / ranges, rem = iters % ranges;
  std::vector<Conv8Seg> segs;
  long it = 0;
  for (int v = 0; v < ranges; ++v) {
    const long end = it + base + (v < rem ? 1 : 0);
    while (it < end) {
      const long tile = it / nk;
      const int kb = (int)(it - tile * nk);
      const int ke = (int)std::min<long>(nk, kb + (end - it));
      segs.push_back(Conv8Seg{(int)tile, kb, ke, 0, 1, 0});
      it += ke - kb;
    }
  }
  return finish_plan(std::move(segs), slots_out);
}

// Tail split: whole rounds of tiles run one tile per workgroup; the tiles of the last, mostly empty round are cut
// into `pieces` equal K ranges each, so that round keeps every CU busy for 1 / pieces of a tile time.  (Equal ranges
// over the whole launch -- plan_segments -- leave slivers of one or two K tiles at every range boundary, each paying
// the pipeline fill and a 256 KiB slab round trip: measured slower than one tile per workgroup on every layer,
// profiles/r3/conv8_layers.txt.)
std::vector<Conv8Seg> plan_tail_split(long tiles, int nk, long full, int pieces, int* slots_out) {
  std::vector<Conv8Seg> segs;
  for (long t = 0; t < full; ++t) segs.push_back(Conv8Seg{(int)t, 0, nk, 0, 1, 0});
  for (long t = full; t < tiles; ++t)
    for (int q = 0; q < pieces; ++q) {
      const int kb = (int)((long)nk * q / pieces), ke = (int)((long)nk * (q + 1) / pieces);
      segs.push_back(Conv8Seg{(int)t, kb, ke, 0, 1, 0});
    }
  return finish_plan(std::move(segs), slots_out);
}

// One workspace per DEVICE: launches of one stream are ordered, and a graph capture (which runs on a stream of its
// own and may not allocate) finds what its eager warm-up steps created.  Two stream-K launches running CONCURRENTLY
// on two streams of one device would share slab and counters -- nothing in this framework does that (the optional
// weight-gradient side stream never runs conv8).
struct Conv8Workspace {
  float* slab = nullptr;
  int slots = 0;
  int* counters = nullptr;
  long tiles = 0;
  std::map<std::tuple<long, int, int>, Conv8Plan> plans;
};
std::mutex g_ws_mu;
std::map<int, Conv8Workspace> g_ws;

// plan + workspace for a launch; allocates (synchronously) on first use of a shape -- never inside a graph capture
// ranges > 0: equal ranges (plan_segments); ranges < 0: tail split into -ranges pieces (plan_tail_split)
const Conv8Plan& plan_for(long tiles, int nk, int ranges, long full, Conv8Workspace** ws_out) {
  int dev = 0;
  DPA_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_ws_mu);
  Conv8Workspace& w = g_ws[dev];
  *ws_out = &w;
  Conv8Plan& pl = w.plans[std::make_tuple(tiles, nk, ranges)];
  if (pl.segs == nullptr) {
    int slots = 0;
    const std::vector<Conv8Seg> segs = ranges > 0 ? plan_segments(tiles, nk, ranges, &slots)
                                                  : plan_tail_split(tiles, nk, full, -ranges, &slots);
    DPA_CHECK_HIP(hipMalloc(&pl.segs, segs.size() * sizeof(Conv8Seg)));
    DPA_CHECK_HIP(hipMemcpy(pl.segs, segs.data(), segs.size() * sizeof(Conv8Seg), hipMemcpyHostToDevice));
    pl.grid = (int)segs.size();
    pl.slots = slots;
  }
  if (w.slots < pl.slots) {
    // (the old slab, if any, may still be in use by a launch in flight: it is left alone, never freed)
    const int want = std::max(pl.slots, 3 * num_cus());
    float* slab = nullptr;
    DPA_CHECK_HIP(hipMalloc(&slab, (size_t)want * 32 * 512 * sizeof(float4)));
    w.slab = slab;
    w.slots = want;
  }
  if (w.tiles < tiles) {
    const long want = std::max<long>(tiles, 1 << 16);
    int* counters = nullptr;
    DPA_CHECK_HIP(hipMalloc(&counters, (size_t)want * sizeof(int)));
    DPA_CHECK_HIP(hipMemset(counters, 0, (size_t)want * sizeof(int)));
    DPA_CHECK_HIP(hipDeviceSynchronize());
    w.counters = counters;
    w.tiles = want;
  }
  return pl;
}

template <typename T, int EPI, bool SK>
void launch_conv8_sk(const IgemmParams& p, const Conv8Extra& e, int grid, hipStream_t stream) {
  auto kern = conv8_kernel<T, EPI, SK>;
  static bool attr_done = false;
  if (!attr_done) {
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kRing9));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), kRing9, stream, p, e);
}
template <typename T, int EPI>
void launch_conv8(const IgemmParams& p, const Conv8Extra& e, int grid, bool sk, hipStream_t stream) {
  if (sk) launch_conv8_sk<T, EPI, true>(p, e, grid, stream);
  else launch_conv8_sk<T, EPI, false>(p, e, grid, stream);
}

}  // namespace

void conv8_set_mode(int mode) { g_conv8_mode = mode; }
void conv8_set_tail_split(int on) { g_conv8_tail_split = on; }
long conv8_launches() { return g_conv8_launches; }
int conv8_last(int i) { return g_conv8_last[i & 3]; }
void conv8_set_min_k(int k) { g_conv8_min_k = k > 0 ? k : 512; }

// Whether conv8 covers the launch (pure function of the problem: the statistics slab of 128-row parts is what
// igemm_stats_rows promises for these shapes).
bool conv8_covers(const IgemmParams& p, int dtype, int out_dtype) {
  if (dtype != out_dtype || (dtype != kBF16 && dtype != kF16)) return false;
  if (p.groups > 1 || p.ntaps < 1 || p.ntaps > kMaxTaps) return false;
  if (p.flags & (kIgemmScatter | kIgemmAxf | kIgemmRStrided)) return false;
  if (p.R != nullptr || p.r_mask != nullptr) return false;
  if (!(p.flags & (kIgemmDense1 | kIgemmRow3)) ) return false;          // tap reach is only promised for these
  if (p.SY != 1 || p.SX != 1 || (long)p.div_ohow.d != (long)p.IH * p.IW || (int)p.div_ow.d != p.IW) return false;
  if (p.M < 256 || p.N % 64 != 0 || p.Cred % 64 != 0 || p.lda % 8 || p.ldb % 8 || p.ldc % 8) return false;
  const bool bnb = p.flags & kIgemmBnBwd, xm = p.flags & kIgemmXMask;
  if (bnb && xm) return false;
  if ((p.flags & kIgemmBnFold) && (!bnb || !(p.bn_relu == 0 || p.bn_relu == 3))) return false;
  if (bnb && (p.stats == nullptr || p.bn_x == nullptr || p.bn_coef == nullptr || p.ldc != p.N ||
              (p.stats_ld != 0 && p.stats_ld != p.N) ||
              !(p.bn_relu == 0 || p.bn_relu == 2 || p.bn_relu == 6 || (p.bn_relu == 3 && p.bn_z != nullptr))))
    return false;
  if (xm && (p.bn_x == nullptr || p.ldc != p.N)) return false;
  if (!bnb && p.stats != nullptr && (p.bias != nullptr || (p.flags & kIgemmRelu) || (p.stats_ld != 0 && p.stats_ld != p.N)))
    return false;
  if ((bnb || xm || p.stats != nullptr) && (p.bias != nullptr || (p.flags & kIgemmRelu))) return false;
  // 32-bit lane offsets inside one tile's windows
  if ((256.0 + 2.0 * (kReach * p.IW + kReach)) * p.lda * 2.0 + (double)p.ntaps * p.Cred * 2.0 >= 2147483648.0) return false;
  if (256.0 * p.ldb * 2.0 >= 2147483648.0 || (double)kMaxTaps * p.Cred * 2.0 >= 2147483648.0) return false;
  if (128.0 * p.ldc * 2.0 >= 4294967296.0) return false;
  return true;
}

// Takes the launch when conv8 covers it and is expected to win; false = the caller runs its other kernels.
bool conv8_try(const IgemmParams& p_in, int dtype, int out_dtype, hipStream_t stream) {
  if (g_conv8_mode == 0 || !conv8_covers(p_in, dtype, out_dtype)) return false;
  IgemmParams p = p_in;
  if (p.stats_ld == 0) p.stats_ld = p.N;
  const long ktot = (long)p.ntaps * p.Cred;
  const int tiles_m = ceil_div(p.M, 256), tiles_n = ceil_div(p.N, 256);
  const long tiles = (long)tiles_m * tiles_n;
  const int nk = (int)(ktot / 64);
  if (tiles * nk >= 2147483647l) return false;
  const int cus = num_cus();
  if (g_conv8_mode < 0) {
    // the pipeline needs a long K loop (fill + epilogue ~ 4.5 us per segment), wide N tiles and enough work per CU
    if (ktot < g_conv8_min_k || p.N % 256 != 0 || tiles * nk < (long)cus * 8) return false;
  }
  Conv8Extra e{};
  e.nk = nk;
  e.kpt = p.Cred / 64;
  e.tiles_m = tiles_m;
  e.tiles_n = tiles_n;
  e.parts = ceil_div(p.M, 128);
  e.shift_pix = kReach * p.IW + kReach;
  // mode 2 (tests): equal ranges over the whole launch, two K tiles per range at least.  Automatic: tail split when the
  // last round of tiles is mostly empty (ResNet-50 layer3 at batch 1024: 784 tiles = 3 rounds + 16 tiles).
  const long iters_all = tiles * nk;
  int ranges = 0;
  long full = 0;
  if (g_conv8_mode == 2) {
    ranges = (int)std::min<long>(cus, std::max<long>(1, iters_all / 2));
    if (ranges < 2) ranges = 0;
  } else if (g_conv8_mode != 1 && g_conv8_tail_split) {
    full = tiles / cus * cus;
    const long tail = tiles - full;
    if (full > 0 && tail > 0 && tail * 3 <= cus && nk >= 4) {
      int pieces = (int)std::min<long>(cus / tail, nk / 2);      // >= 2 K tiles per piece, one sub-round of pieces
      pieces = std::min(pieces, g_conv8_tail_split);   // the last arriver pulls every slab through ONE CU (~100 GB/s)
      if (pieces >= 2) ranges = -pieces;
    }
  }
  int grid;
  const bool stream_k = ranges != 0;
  if (stream_k) {
    Conv8Workspace* w = nullptr;
    const Conv8Plan& pl = plan_for(tiles, nk, ranges, full, &w);
    grid = pl.grid;
    e.segs = pl.segs;
    e.slab = w->slab;
    e.counters = w->counters;
  } else {
    grid = (int)tiles;
  }
  const bool bnb = p.flags & kIgemmBnBwd, xm = p.flags & kIgemmXMask;
  const int epi = bnb ? kEpiBnb : (xm ? kEpiXMask : (p.stats != nullptr ? kEpiStats : kEpiPlain));
  if (dtype == kBF16) {
    switch (epi) {
      case kEpiBnb: launch_conv8<bf16_t, kEpiBnb>(p, e, grid, stream_k, stream); break;
      case kEpiXMask: launch_conv8<bf16_t, kEpiXMask>(p, e, grid, stream_k, stream); break;
      case kEpiStats: launch_conv8<bf16_t, kEpiStats>(p, e, grid, stream_k, stream); break;
      default: launch_conv8<bf16_t, kEpiPlain>(p, e, grid, stream_k, stream); break;
    }
  } else {
    switch (epi) {
      case kEpiBnb: launch_conv8<f16_t, kEpiBnb>(p, e, grid, stream_k, stream); break;
      case kEpiXMask: launch_conv8<f16_t, kEpiXMask>(p, e, grid, stream_k, stream); break;
      case kEpiStats: launch_conv8<f16_t, kEpiStats>(p, e, grid, stream_k, stream); break;
      default: launch_conv8<f16_t, kEpiPlain>(p, e, grid, stream_k, stream); break;
    }
  }
  DPA_CHECK_HIP(hipGetLastError());
  g_conv8_last[0] = grid;
  g_conv8_last[1] = ranges;
  g_conv8_last[3] = cus;
  ++g_conv8_launches;
  return true;
}

// (tests) nothing in the stream-K hand-off can time out any more (no workgroup waits for another): always 0
int conv8_error(hipStream_t /*stream*/) { return 0; }

}  // namespace dpa
