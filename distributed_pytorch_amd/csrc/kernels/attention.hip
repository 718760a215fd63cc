// Fused multi-head self-attention for the zoo's transformers (ViT, Swin, MaxViT) on gfx950 MFMA.
//
//     O[b,h] = softmax(scale[h] * Q K^T + bias[b % nbias, h]) V        S <= a few hundred tokens, D = 32 / 64 / 80 / ...
//
// The input is the packed projection [B][S][3][H][D] exactly as the qkv Linear writes it (no head transposes), the
// output [B][S][H][D] is what the output projection reads.  Scores never touch memory.
//
// Everything is built from ONE MFMA shape, v_mfma_f32_32x32x16 (bf16 / f16), in the orientation the convolution
// kernels use:  C[i][j] += sum_k A[i][k] * B[j][k]  with lane (l31 = lane & 31, lh = lane >> 5) holding
// C[(r&3) + 8*(r>>2) + 4*lh][l31] in accumulator register r, operand A rows <-> the register axis, operand B rows
// <-> l31, and each lane supplying 8 consecutive k at 8*lh.
//   * contraction over D (Q.K, dO.V): both operands are read row-major from LDS / global with one b128 per lane.
//   * contraction over tokens (P.V, dS.K, P^T.dO, dS^T.Q): the probabilities stay in their accumulator registers and
//     become the B operand directly -- register r of lane (l31, lh) is token (r&3) + 8*(r>>2) + 4*lh, so the 16-wide
//     MFMA step s consumes registers 8s..8s+7, i.e. a PERMUTED token order; the other operand is fetched with the
//     gfx950 transposed LDS read (ds_read_b64_tr_b16) from rows in the same permuted order.  No LDS round trip of P.
//   * softmax rows live in two lanes (l31, lh = 0 / 1): the row max / sum need one cross-lane exchange (xor 32).
// Forward is flash-style (online softmax over 32-key blocks) and writes the log-sum-exp per row.  Backward is two
// kernels without atomics: dQ (a wave owns 32 queries, loops over key blocks) and dK / dV (a wave owns 32 keys,
// loops over query blocks); each recomputes P from the saved log-sum-exp.  The relative-position-bias gradient
// (Swin / MaxViT, S <= 64) is accumulated in registers over a strip of batch entries and leaves as one partial slab
// per workgroup (folded by slab_reduce: deterministic).
//
// Replaces what torchvision's nn.MultiheadAttention / ShiftedWindowAttention / RelativePositionalMultiHeadAttention
// run through ATen bmm + softmax (reference model zoo, nd_imagenet.py:22-24,133-138).
#include "../common.h"
#include "kernels.h"
#include <stdexcept>

namespace dpa {
namespace {

typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

__device__ __forceinline__ uint2 attn_tr_read(const unsigned char* p) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  uint2 r;
  __builtin_memcpy(&r, &v, 8);
  return r;
}

template <typename T> struct AMma;
template <> struct AMma<bf16_t> {
  static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    bf16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
};
template <> struct AMma<f16_t> {
  static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
    f16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  }
};

// token index inside a 32-block of accumulator register r for lane half lh
__device__ __forceinline__ int reg_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

template <typename T> __device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  Vec16<T> o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o.set(j, v[j]);
  return o.raw;
}

// LDS tile [rows][DP] of T with row stride LD bytes; rows beyond `nrows_valid` and columns beyond D are zero.
// Eight chunks per thread are requested before the first is stored (one memory latency per trip instead of one per
// chunk: the per-chunk load -> store loop made the staging the longest phase of a 197-token workgroup), and the
// (row, chunk) of a thread advances incrementally (no division per chunk).
template <typename T>
__device__ __forceinline__ void stage_rows(unsigned char* dst, int LD, const T* __restrict__ src, long src_ld, int row0,
                                           int nrows, int nrows_valid, int D, int DP, int tid, int nthreads) {
  constexpr int U = 8;
  const int cpr = DP / 8;            // 16-byte chunks per staged row
  const int total = nrows * cpr;
  const int dr = nthreads / cpr, dc = nthreads - dr * cpr;
  int r = tid / cpr, c = tid - r * cpr;
  for (int i0 = tid; i0 < total; i0 += U * nthreads) {
    uint4 v[U];
    int off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = make_uint4(0u, 0u, 0u, 0u);
      off[u] = -1;
      if (i0 + u * nthreads < total) {
        off[u] = r * LD + c * 16;
        if (row0 + r < nrows_valid && c * 8 < D) v[u] = *reinterpret_cast<const uint4*>(src + (long)(row0 + r) * src_ld + c * 8);
      }
      r += dr; c += dc;
      if (c >= cpr) { c -= cpr; ++r; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (off[u] >= 0) *reinterpret_cast<uint4*>(dst + off[u]) = v[u];
  }
}

// B-operand fragment of a row-major LDS tile: row = row0 + l31, k = 16-element step ks (contraction axis contiguous)
__device__ __forceinline__ uint4 frag_direct(const unsigned char* tile, int LD, int row, int ks, int lh) {
  return *reinterpret_cast<const uint4*>(tile + row * LD + ks * 32 + lh * 16);
}

// A-operand fragment with the CONTRACTION along the LDS rows: rows (tokens) tok0 + {4*lh + 0..3, 8 + 4*lh + 0..3} (the
// permuted order of a probability accumulator), columns col0 + 0..31 <-> the accumulator's register axis.
__device__ __forceinline__ uint4 frag_tr(const unsigned char* tile, int LD, int tok0, int col0, int lane) {
  const int g = lane >> 4, gi = lane & 15;
  const int trow = 4 * (g >> 1) + (gi >> 2);
  const int tcol = 16 * (g & 1) + 4 * (gi & 3);
  const unsigned char* base = tile + (tok0 + trow) * LD + (col0 + tcol) * 2;
  const uint2 lo = attn_tr_read(base), hi = attn_tr_read(base + 8 * LD);
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

struct AttnArgs {
  const void* qkv;       // [B][S][3][H][D]
  void* out;             // [B][S][H][D]                     (forward)
  float* lse;            // [B][H][S]
  const float* bias;     // optional [nbias][H][S][S], batch entry b uses b % nbias
  const float* scale;    // [H]
  const void* dout;      // [B][S][H][D]                     (backward)
  const float* delta;    // [B][H][S] = rowsum(dO * O)
  void* dqkv;            // [B][S][3][H][D]
  float* dbias;          // optional partial slabs [nsplit][H][S][S]
  int B, H, S, D, nbias, nsplit;
};

constexpr int kAttnMaxDB = 4;      // D <= 128.  The kernels are instantiated for MDB = 1, 2, 4 blocks of 32 head
                                   // channels (D <= 32 / 64 / 128): accumulators and operand fragments are sized by MDB, and
                                   // sizing them for 128 cost D = 64 heads half their occupancy (419 registers, 1 wave per SIMD)

// ------------------------------------------------------------------------------------------------ forward / dQ
// BWD = false: O and log-sum-exp.  BWD = true: dQ (same loop + the dP product).  Workgroup = NW waves = 32 * NW queries of
// one (b, h); K and V (all S tokens) are staged in LDS once per workgroup (NW = 4 for S > 64: half the staging traffic
// and twice the waves per LDS byte; NW = 2 for single-window problems, where 4 waves would idle).
template <typename T, bool BWD, int NW, int MDB>
__global__ __launch_bounds__(64 * NW) void attn_q_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int S = a.S, D = a.D, H = a.H;
  const int b = blockIdx.y / H, h = blockIdx.y - b * H;
  const int DP = (D + 31) / 32 * 32, NDB = DP / 32, KS = (D + 15) / 16;
  const int SP = (S + 31) / 32 * 32;
  const int LD = DP * 2 + 16;
  unsigned char* sk = smem;
  unsigned char* sv = smem + SP * LD;
  const long row_ld = 3l * H * D;
  const T* qkv = reinterpret_cast<const T*>(a.qkv) + (long)b * S * row_ld + (long)h * D;
  stage_rows<T>(sk, LD, qkv + (long)H * D, row_ld, 0, SP, S, D, DP, tid, 64 * NW);
  stage_rows<T>(sv, LD, qkv + 2l * H * D, row_ld, 0, SP, S, D, DP, tid, 64 * NW);

  const int q = blockIdx.x * (32 * NW) + wave * 32 + l31;       // this lane's query row
  const bool qok = q < S;
  // Q (and dO) fragments straight from global memory: row q, 8 elements at 16*ks + 8*lh
  uint4 qf[2 * MDB], gf[2 * MDB];
#pragma unroll
  for (int ks = 0; ks < 2 * MDB; ++ks) {
    qf[ks] = gf[ks] = make_uint4(0u, 0u, 0u, 0u);
    const int c = ks * 16 + lh * 8;
    if (ks < KS && qok && c < D) {
      qf[ks] = *reinterpret_cast<const uint4*>(qkv + (long)q * row_ld + c);
      if (BWD) gf[ks] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.dout) +
                                                        ((long)b * S + q) * H * D + (long)h * D + c);
    }
  }
  const float scale = a.scale[h];
  const float* bias = a.bias ? a.bias + ((long)(b % a.nbias) * H + h) * S * S + (long)(qok ? q : 0) * S : nullptr;
  float lse_q = 0.f, delta_q = 0.f;
  if (BWD && qok) {
    lse_q = a.lse[((long)b * H + h) * S + q];
    delta_q = a.delta[((long)b * H + h) * S + q];
  }
  __syncthreads();

  f32x16_t acc[MDB];
#pragma unroll
  for (int d = 0; d < MDB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int kb = 0; kb < SP / 32; ++kb) {
    // S^T block: keys <-> registers, queries <-> l31
    f32x16_t sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2 * MDB; ++ks)
      if (ks < KS) AMma<T>::run(sc, frag_direct(sk, LD, kb * 32 + l31, ks, lh), qf[ks]);
    float p[16];
    float mblk = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + reg_row(r, lh);
      float s = sc[r] * scale;
      if (bias != nullptr && key < S) s += bias[key];
      if (key >= S) s = -INFINITY;
      p[r] = s;
      mblk = fmaxf(mblk, s);
    }
    if (!BWD) {
      mblk = fmaxf(mblk, __shfl_xor(mblk, 32, 64));
      const float m_new = fmaxf(m_run, mblk);
      const float alpha = m_run == -INFINITY ? 0.f : __expf(m_run - m_new);
      float lsum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = m_new == -INFINITY ? 0.f : __expf(p[r] - m_new);
        lsum += p[r];
      }
      l_run = l_run * alpha + lsum;
      m_run = m_new;
#pragma unroll
      for (int d = 0; d < MDB; ++d)
        if (d < NDB)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[d][r] *= alpha;
    } else {
      // P from the saved log-sum-exp, dP^T = V dO^T, dS^T = P (dP - delta)
      f32x16_t dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2 * MDB; ++ks)
        if (ks < KS) AMma<T>::run(dp, frag_direct(sv, LD, kb * 32 + l31, ks, lh), gf[ks]);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = (qok && p[r] != -INFINITY) ? __expf(p[r] - lse_q) : 0.f;
        p[r] = pr * (dp[r] - delta_q) * scale;           // gradient of the raw Q.K product
      }
    }
    // acc[d][query] += sum_key p[key] * (V | K)[key][d]: the probabilities are the B operand as they sit
    const unsigned char* tile = BWD ? sk : sv;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float half[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) half[j] = p[8 * s2 + j];
      const uint4 pf = pack8<T>(half);
#pragma unroll
      for (int d = 0; d < MDB; ++d)
        if (d < NDB) AMma<T>::run(acc[d], frag_tr(tile, LD, kb * 32 + 16 * s2, d * 32, lane), pf);
    }
  }

  if (!BWD) {
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (qok) {
      T* o = reinterpret_cast<T*>(a.out) + ((long)b * S + q) * H * D + (long)h * D;
#pragma unroll
      for (int d = 0; d < MDB; ++d)
        if (d < NDB)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c = d * 32 + 8 * g + 4 * lh;
            if (c < D) {       // D is a multiple of 8: 4 consecutive channels are all in or all out
              T v4[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) v4[j] = from_float<T>(acc[d][4 * g + j] * inv);
              uint2 w;
              __builtin_memcpy(&w, v4, 8);
              *reinterpret_cast<uint2*>(o + c) = w;
            }
          }
      if (lh == 0) a.lse[((long)b * H + h) * S + q] = m_run + __logf(l_tot);
    }
  } else if (qok) {
    T* o = reinterpret_cast<T*>(a.dqkv) + ((long)b * S + q) * row_ld + (long)h * D;
#pragma unroll
    for (int d = 0; d < MDB; ++d)
      if (d < NDB)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = d * 32 + 8 * g + 4 * lh;
          if (c < D) {
            T v4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v4[j] = from_float<T>(acc[d][4 * g + j]);
            uint2 w;
            __builtin_memcpy(&w, v4, 8);
            *reinterpret_cast<uint2*>(o + c) = w;
          }
        }
  }
}

// ------------------------------------------------------------------------------------------------ dK, dV (, dbias)
// Workgroup = 2 waves = 64 keys of head h; it walks batch entries b = split, split + nsplit, ...  Q and dO of the
// current (b, h) are staged in LDS (all S tokens), the wave's own K / V rows are read as fragments from global memory.
// DBIAS (S <= 64: one workgroup covers every key): dS accumulates in registers over the strip of batch entries.
template <typename T, bool DBIAS, int NW, int MDB>
__global__ __launch_bounds__(64 * NW, MDB <= 2 ? 2 : 1) void attn_kv_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int S = a.S, D = a.D, H = a.H;
  const int split = blockIdx.y / H, h = blockIdx.y - split * H;
  const int DP = (D + 31) / 32 * 32, NDB = DP / 32, KS = (D + 15) / 16;
  const int SP = (S + 31) / 32 * 32;
  const int LD = DP * 2 + 16;
  unsigned char* sq = smem;
  unsigned char* sg = smem + SP * LD;
  float* s_lse = reinterpret_cast<float*>(smem + 2 * SP * LD);
  float* s_delta = s_lse + SP;
  const long row_ld = 3l * H * D;
  const int key = blockIdx.x * (32 * NW) + wave * 32 + l31;     // this lane's key row
  const bool kok = key < S;
  const float scale = a.scale[h];

  f32x16_t dsacc[DBIAS ? 2 : 1];
#pragma unroll
  for (int i = 0; i < (DBIAS ? 2 : 1); ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dsacc[i][r] = 0.f;

  for (int b = split; b < a.B; b += a.nsplit) {
    const T* qkv = reinterpret_cast<const T*>(a.qkv) + (long)b * S * row_ld + (long)h * D;
    const T* dout = reinterpret_cast<const T*>(a.dout) + (long)b * S * H * D + (long)h * D;
    __syncthreads();                       // previous batch entry fully consumed
    stage_rows<T>(sq, LD, qkv, row_ld, 0, SP, S, D, DP, tid, 64 * NW);
    stage_rows<T>(sg, LD, dout, (long)H * D, 0, SP, S, D, DP, tid, 64 * NW);
    for (int i = tid; i < SP; i += 64 * NW) {
      s_lse[i] = i < S ? a.lse[((long)b * H + h) * S + i] : 0.f;
      s_delta[i] = i < S ? a.delta[((long)b * H + h) * S + i] : 0.f;
    }
    uint4 kf[2 * MDB], vf[2 * MDB];
#pragma unroll
    for (int ks = 0; ks < 2 * MDB; ++ks) {
      kf[ks] = vf[ks] = make_uint4(0u, 0u, 0u, 0u);
      const int c = ks * 16 + lh * 8;
      if (ks < KS && kok && c < D) {
        kf[ks] = *reinterpret_cast<const uint4*>(qkv + (long)key * row_ld + (long)H * D + c);
        vf[ks] = *reinterpret_cast<const uint4*>(qkv + (long)key * row_ld + 2l * H * D + c);
      }
    }
    const float* bias = a.bias ? a.bias + ((long)(b % a.nbias) * H + h) * S * S : nullptr;
    __syncthreads();

    f32x16_t dk[MDB], dv[MDB];
#pragma unroll
    for (int d = 0; d < MDB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) dk[d][r] = dv[d][r] = 0.f;

    for (int qb = 0; qb < SP / 32; ++qb) {
      // S block: queries <-> registers, keys <-> l31
      f32x16_t sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = dp[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2 * MDB; ++ks)
        if (ks < KS) {
          AMma<T>::run(sc, frag_direct(sq, LD, qb * 32 + l31, ks, lh), kf[ks]);
          AMma<T>::run(dp, frag_direct(sg, LD, qb * 32 + l31, ks, lh), vf[ks]);
        }
      float p[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qrow = qb * 32 + reg_row(r, lh);
        float s = sc[r] * scale;
        if (bias != nullptr && qrow < S && kok) s += bias[(long)qrow * S + key];
        const float pr = (qrow < S && kok) ? __expf(s - s_lse[qrow]) : 0.f;
        p[r] = pr;
        ds[r] = pr * (dp[r] - s_delta[qrow]);
        if (DBIAS) dsacc[qb & 1][r] += ds[r];
        ds[r] *= scale;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float hp[8], hd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          hp[j] = p[8 * s2 + j];
          hd[j] = ds[8 * s2 + j];
        }
        const uint4 pf = pack8<T>(hp), df = pack8<T>(hd);
#pragma unroll
        for (int d = 0; d < MDB; ++d)
          if (d < NDB) {
            AMma<T>::run(dv[d], frag_tr(sg, LD, qb * 32 + 16 * s2, d * 32, lane), pf);   // dV += P^T dO
            AMma<T>::run(dk[d], frag_tr(sq, LD, qb * 32 + 16 * s2, d * 32, lane), df);   // dK += dS^T Q
          }
      }
    }
    if (kok) {
      T* o = reinterpret_cast<T*>(a.dqkv) + ((long)b * S + key) * row_ld + (long)h * D;
#pragma unroll
      for (int d = 0; d < MDB; ++d)
        if (d < NDB)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c = d * 32 + 8 * g + 4 * lh;
            if (c < D) {
              T k4[4], v4[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                k4[j] = from_float<T>(dk[d][4 * g + j]);
                v4[j] = from_float<T>(dv[d][4 * g + j]);
              }
              uint2 w;
              __builtin_memcpy(&w, k4, 8);
              *reinterpret_cast<uint2*>(o + (long)H * D + c) = w;
              __builtin_memcpy(&w, v4, 8);
              *reinterpret_cast<uint2*>(o + 2l * H * D + c) = w;
            }
          }
    }
  }
  if (DBIAS && kok) {
    float* o = a.dbias + ((long)split * H + h) * S * S;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qrow = qb * 32 + reg_row(r, lh);
        if (qrow < S) o[(long)qrow * S + key] = dsacc[qb][r];
      }
  }
}

// delta[b][h][s] = sum_d dO * O  (one wave per 64 / (D/8) rows: a row of D elements is D/8 lanes of 16 bytes)
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout,
                                                         float* __restrict__ delta, long rows_bs, int H, int S, int D) {
  // one thread per (b, s, h): D <= 128 elements, 16-byte loads
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_bs * H) return;
  const long bs = i / H;
  const int h = (int)(i - bs * H);
  const T* po = o + bs * H * D + (long)h * D;
  const T* pg = dout + bs * H * D + (long)h * D;
  float acc = 0.f;
  for (int c = 0; c < D; c += 8) {
    Vec16<T> vo, vg;
    vo.load(po + c);
    vg.load(pg + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += vo.get(j) * vg.get(j);
  }
  const long b = bs / S;
  const int s = (int)(bs - b * S);
  delta[(b * H + h) * S + s] = acc;
}

// ---- window / grid partition of NHWC token maps (Swin, MaxViT) ------------------------------------------------------
// x [B][H][W][C] <-> windows [B * nh * nw][wh * ww][C].  Token (i, j) of window (gh, gw) is the pixel
//   window mode: (gh * wh + i + shift_h, gw * ww + j + shift_w) mod (Hp, Wp)      (Swin: cyclic shift + partition in one)
//   grid mode  : (i * nh + gh, j * nw + gw)                                        (MaxViT grid attention)
// of the image zero-padded to Hp = nh * wh, Wp = nw * ww.  REVERSE writes the image from the windows (the padding is
// dropped).  Both directions are gathers of 16-byte chunks, so each is the other's gradient.
struct WinGeom { int B, H, W, C16, wh, ww, nh, nw, sh, sw, grid; };

template <bool REVERSE>
__global__ __launch_bounds__(256) void window_partition_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                               WinGeom g) {
  const int Hp = g.nh * g.wh, Wp = g.nw * g.ww;
  const long total = REVERSE ? (long)g.B * g.H * g.W * g.C16 : (long)g.B * Hp * Wp * g.C16;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int c = (int)(idx % g.C16);
    long t = idx / g.C16;
    if (!REVERSE) {
      // t enumerates (b, gh, gw, i, j)
      const int j = (int)(t % g.ww); t /= g.ww;
      const int i = (int)(t % g.wh); t /= g.wh;
      const int gw = (int)(t % g.nw); t /= g.nw;
      const int gh = (int)(t % g.nh);
      const long b = t / g.nh;
      int y, x;
      if (g.grid) {
        y = i * g.nh + gh;
        x = j * g.nw + gw;
      } else {
        y = gh * g.wh + i + g.sh;
        x = gw * g.ww + j + g.sw;
        if (y >= Hp) y -= Hp;
        if (x >= Wp) x -= Wp;
      }
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (y < g.H && x < g.W) v = src[((b * g.H + y) * g.W + x) * g.C16 + c];
      dst[idx] = v;
    } else {
      // t enumerates (b, y, x) of the unpadded image
      const int x = (int)(t % g.W); t /= g.W;
      const int y = (int)(t % g.H);
      const long b = t / g.H;
      int gh, gw, i, j;
      if (g.grid) {
        i = y / g.nh; gh = y - i * g.nh;
        j = x / g.nw; gw = x - j * g.nw;
      } else {
        int yy = y - g.sh, xx = x - g.sw;
        if (yy < 0) yy += Hp;
        if (xx < 0) xx += Wp;
        gh = yy / g.wh; i = yy - gh * g.wh;
        gw = xx / g.ww; j = xx - gw * g.ww;
      }
      dst[idx] = src[((((b * g.nh + gh) * g.nw + gw) * g.wh + i) * g.ww + j) * g.C16 + c];
    }
  }
}

// ---- cosine attention pre-pass (Swin V2): q <- mult[h] * q / max(|q|, eps), k <- k / max(|k|, eps), v copied -----------
// One thread per (token, which in {q, k, v}, head).  inv[token][2][H] keeps 1 / max(norm, eps) for backward.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void qk_normalize_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           T* __restrict__ out, const float* __restrict__ mult,
                                                           float* __restrict__ inv, float* __restrict__ dmult_part,
                                                           long tokens, int H, int D) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tokens * 3 * H) return;
  const int h = (int)(i % H);
  const int which = (int)((i / H) % 3);
  const long tok = i / (3l * H);
  const long off = (tok * 3 + which) * H * D + (long)h * D;
  if (which == 2) {          // v: identity
    for (int c = 0; c < D; c += 8) *reinterpret_cast<uint4*>(out + off + c) = *reinterpret_cast<const uint4*>((BWD ? dy : x) + off + c);
    return;
  }
  const float m = which == 0 ? mult[h] : 1.f;
  if (!BWD) {
    float ss = 0.f;
    for (int c = 0; c < D; c += 8) {
      Vec16<T> v;
      v.load(x + off + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v.get(j) * v.get(j);
    }
    const float r = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    inv[(tok * 2 + which) * H + h] = r;
    for (int c = 0; c < D; c += 8) {
      Vec16<T> v, o;
      v.load(x + off + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) o.set(j, v.get(j) * r * m);
      o.store(out + off + c);
    }
  } else {
    // y = m * x * r;  dx = m * r * (dy - xhat (xhat . dy)),  dm += dy . xhat
    const float r = inv[(tok * 2 + which) * H + h];
    float dot = 0.f;
    for (int c = 0; c < D; c += 8) {
      Vec16<T> v, g;
      v.load(x + off + c);
      g.load(dy + off + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) dot += g.get(j) * v.get(j) * r;
    }
    if (which == 0) dmult_part[tok * H + h] = dot;
    for (int c = 0; c < D; c += 8) {
      Vec16<T> v, g, o;
      v.load(x + off + c);
      g.load(dy + off + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // A zero vector (Swin V2's padding tokens: no key bias) has r = 1 / eps = 1e12: its gradient is finite in fp32 /
        // bf16 and multiplies exact zeros downstream, but it would be inf in fp16 (inf * 0 = NaN): saturate.
        float d = m * r * (g.get(j) - v.get(j) * r * dot);
        if (TypeInfo<T>::kCode == kF16) d = fminf(fmaxf(d, -60000.f), 60000.f);
        o.set(j, d);
      }
      o.store(out + off + c);
    }
  }
}

void check_args(int B, int H, int S, int D, int dtype) {
  if (dtype != kBF16 && dtype != kF16) throw std::invalid_argument("attention: bf16 / f16 operands");
  if (B <= 0 || H <= 0 || S <= 0) throw std::invalid_argument("attention: empty problem");
  if (D % 8 != 0 || D <= 0 || D > 32 * kAttnMaxDB) throw std::invalid_argument("attention: head dim must be a multiple of 8, <= 128");
  if ((long)B * H > 2147483647l / 64) throw std::invalid_argument("attention: too many (batch, head) pairs for one launch");
}

int lds_bytes(int S, int D, bool with_rows) {
  const int DP = (D + 31) / 32 * 32, SP = (S + 31) / 32 * 32;
  return 2 * SP * (DP * 2 + 16) + (with_rows ? 2 * SP * 4 : 0);
}

template <typename K> void set_lds(K kern, int lds) {
  if (lds > 64 * 1024)
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
}

}  // namespace

void window_partition_launch(const void* src, void* dst, int reverse, int B, int H, int W, int C, int wh, int ww, int nh,
                             int nw, int sh, int sw, int grid_mode, int dtype, hipStream_t s) {
  const int vec = dtype == kF32 ? 4 : 8;
  if (C % vec) throw std::invalid_argument("window_partition: C must fill 16-byte vectors");
  if (B <= 0 || H <= 0 || W <= 0) return;
  if (nh * wh < H || nw * ww < W || sh < 0 || sw < 0 || sh >= nh * wh || sw >= nw * ww)
    throw std::invalid_argument("window_partition: windows do not cover the image / bad shift");
  if (grid_mode && (sh || sw)) throw std::invalid_argument("window_partition: grid mode has no shift");
  WinGeom g{B, H, W, C / vec, wh, ww, nh, nw, sh, sw, grid_mode};
  const long total = reverse ? (long)B * H * W * g.C16 : (long)B * nh * wh * nw * ww * g.C16;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (reverse)
    hipLaunchKernelGGL(window_partition_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint4*)src,
                       (uint4*)dst, g);
  else
    hipLaunchKernelGGL(window_partition_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint4*)src,
                       (uint4*)dst, g);
  DPA_CHECK_HIP(hipGetLastError());
}

void qk_normalize_launch(const void* x, const void* dy, void* out, const float* mult, float* inv, float* dmult_part,
                         long tokens, int H, int D, int backward, int dtype, hipStream_t s) {
  if (dtype != kBF16 && dtype != kF16) throw std::invalid_argument("qk_normalize: bf16 / f16");
  if (D % 8) throw std::invalid_argument("qk_normalize: head dim must be a multiple of 8");
  if (tokens <= 0) return;
  const long n = tokens * 3 * H;
  const unsigned blocks = (unsigned)((n + 255) / 256);
#define DPA_QKN(T)                                                                                                     \
  if (backward)                                                                                                        \
    hipLaunchKernelGGL((qk_normalize_kernel<T, true>), dim3(blocks), dim3(256), 0, s, (const T*)x, (const T*)dy, (T*)out, \
                       mult, inv, dmult_part, tokens, H, D);                                                           \
  else                                                                                                                 \
    hipLaunchKernelGGL((qk_normalize_kernel<T, false>), dim3(blocks), dim3(256), 0, s, (const T*)x, (const T*)dy, (T*)out, \
                       mult, inv, dmult_part, tokens, H, D);
  if (dtype == kBF16) { DPA_QKN(bf16_t) } else { DPA_QKN(f16_t) }
#undef DPA_QKN
  DPA_CHECK_HIP(hipGetLastError());
}

int attn_max_tokens(int D) {
  // K and V (forward) or Q and dO + two fp32 rows (backward) of all tokens must fit the 160 KB LDS of a CU
  const int DP = (D + 31) / 32 * 32;
  return (160 * 1024 - 1024) / (2 * (DP * 2 + 16) + 8) / 32 * 32;
}

void attn_fwd_launch(const void* qkv, void* out, float* lse, const float* bias, int nbias, const float* scale, int B,
                     int H, int S, int D, int dtype, hipStream_t s) {
  check_args(B, H, S, D, dtype);
  if (S > attn_max_tokens(D)) throw std::invalid_argument("attention: sequence too long for the single-pass kernel");
  AttnArgs a{};
  a.qkv = qkv; a.out = out; a.lse = lse; a.bias = bias; a.nbias = nbias > 0 ? nbias : 1; a.scale = scale;
  a.B = B; a.H = H; a.S = S; a.D = D;
  const int lds = lds_bytes(S, D, false);
  const int nw = S > 64 ? 4 : 2;
  dim3 grid(ceil_div(S, 32 * nw), B * H);
#define DPA_ATTN_FWD2(T, NW, MDB)                                                            \
  set_lds(attn_q_kernel<T, false, NW, MDB>, lds);                                            \
  hipLaunchKernelGGL((attn_q_kernel<T, false, NW, MDB>), grid, dim3(64 * NW), lds, s, a);
#define DPA_ATTN_FWD(T, NW)                                                                  \
  if (D <= 32) { DPA_ATTN_FWD2(T, NW, 1) } else if (D <= 64) { DPA_ATTN_FWD2(T, NW, 2) } else { DPA_ATTN_FWD2(T, NW, 4) }
  if (dtype == kBF16) {
    if (nw == 4) { DPA_ATTN_FWD(bf16_t, 4) } else { DPA_ATTN_FWD(bf16_t, 2) }
  } else {
    if (nw == 4) { DPA_ATTN_FWD(f16_t, 4) } else { DPA_ATTN_FWD(f16_t, 2) }
  }
#undef DPA_ATTN_FWD2
#undef DPA_ATTN_FWD
  DPA_CHECK_HIP(hipGetLastError());
}

int attn_dbias_splits(int B, int H) {
  int n = 1024 / (H > 0 ? H : 1);
  if (n < 1) n = 1;
  return n < B ? n : B;
}

void attn_bwd_launch(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                     const float* bias, int nbias, const float* scale, void* dqkv, float* dbias, int nsplit, int B, int H,
                     int S, int D, int dtype, hipStream_t s) {
  check_args(B, H, S, D, dtype);
  if (S > attn_max_tokens(D)) throw std::invalid_argument("attention: sequence too long for the single-pass kernel");
  if (dbias != nullptr && (S > 64 || nsplit < 1 || nsplit > B))
    throw std::invalid_argument("attention: the bias gradient needs S <= 64 and 1 <= nsplit <= B");
  AttnArgs a{};
  a.qkv = qkv; a.out = const_cast<void*>(out); a.lse = const_cast<float*>(lse); a.bias = bias;
  a.nbias = nbias > 0 ? nbias : 1; a.scale = scale; a.dout = dout; a.delta = delta; a.dqkv = dqkv; a.dbias = dbias;
  a.B = B; a.H = H; a.S = S; a.D = D; a.nsplit = dbias != nullptr ? nsplit : B;
  const long rows = (long)B * S;
  const int dgrid = ceil_div(rows * H, 256);
  const int lds_q = lds_bytes(S, D, false), lds_kv = lds_bytes(S, D, true);
  const int nw = S > 64 ? 4 : 2;
  dim3 gq(ceil_div(S, 32 * nw), B * H), gkv(ceil_div(S, 32 * nw), a.nsplit * H);
#define DPA_ATTN_BWD2(T, NW, MDB)                                                                                    \
  set_lds(attn_q_kernel<T, true, NW, MDB>, lds_q);                                                                   \
  hipLaunchKernelGGL((attn_q_kernel<T, true, NW, MDB>), gq, dim3(64 * NW), lds_q, s, a);                             \
  if (dbias != nullptr) {                                                                                            \
    set_lds(attn_kv_kernel<T, true, 2, MDB>, lds_kv);                                                                \
    hipLaunchKernelGGL((attn_kv_kernel<T, true, 2, MDB>), gkv, dim3(128), lds_kv, s, a);                             \
  } else {                                                                                                           \
    set_lds(attn_kv_kernel<T, false, NW, MDB>, lds_kv);                                                              \
    hipLaunchKernelGGL((attn_kv_kernel<T, false, NW, MDB>), gkv, dim3(64 * NW), lds_kv, s, a);                       \
  }
#define DPA_ATTN_BWD(T, NW)                                                                                          \
  hipLaunchKernelGGL(attn_delta_kernel<T>, dim3(dgrid), dim3(256), 0, s, (const T*)out, (const T*)dout, delta, rows, H, \
                     S, D);                                                                                          \
  if (D <= 32) { DPA_ATTN_BWD2(T, NW, 1) } else if (D <= 64) { DPA_ATTN_BWD2(T, NW, 2) } else { DPA_ATTN_BWD2(T, NW, 4) }
  if (dtype == kBF16) {
    if (nw == 4) { DPA_ATTN_BWD(bf16_t, 4) } else { DPA_ATTN_BWD(bf16_t, 2) }
  } else {
    if (nw == 4) { DPA_ATTN_BWD(f16_t, 4) } else { DPA_ATTN_BWD(f16_t, 2) }
  }
#undef DPA_ATTN_BWD2
#undef DPA_ATTN_BWD
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
