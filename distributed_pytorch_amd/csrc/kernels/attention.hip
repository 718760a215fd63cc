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

constexpr int kAttnMaxDB = 4;      // D <= 128.  The kernels are instantiated for NDB = 1 .. 4 blocks of 32 head channels
                                   // (D <= 32 / 64 / 96 / 128): accumulators and operand fragments are sized by NDB (sizing
                                   // them for 128 cost D = 64 heads half their occupancy: 419 registers, 1 wave per SIMD)

// ------------------------------------------------------------------------------------------------ forward / dQ
// BWD = false: O and log-sum-exp.  BWD = true: dQ (same loop + the dP product).  Workgroup = NW waves = 32 * NW queries of
// one (b, h); K and V (all S tokens) are staged in LDS once per workgroup.  NW (the launch's block size) is the number of
// 32-query blocks when that is <= 8 -- ONE workgroup per (b, h): S = 197 runs 7 waves and stages K / V once, where 2
// workgroups of 4 waves staged them twice and idled a wave -- otherwise the blocks are split evenly over workgroups.
//
// The loop is VALU bound (rocprofv3 --pmc, profiles/r3/pmc_attn.md: 7 % MFMA busy, 70 % VALU), so everything in it is
// compile-time shaped: NDB = exact number of 32-channel blocks of the head (D padded with zeros; runtime `ks < KS` /
// `d < NDB` guards made every accumulator a PHI that the compiler shuttled between AGPRs and VGPRs: 208 of 496 VALU
// instructions per key block), BIAS a template parameter, key-padding masks in the LAST key block only, softmax in the
// log2 domain (one fma + v_exp_f32 per score), the running-max rescale of the output accumulators only when some
// lane's maximum moved (a wave-uniform branch: after the first blocks it rarely does).
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

template <typename T> __device__ __forceinline__ uint4 pack8_fast(const float (&v)[8]) { return pack8<T>(v); }
template <> __device__ __forceinline__ uint4 pack8_fast<bf16_t>(const float (&v)[8]) {
  f32x2_t a, b, c, d;
  a.x = v[0]; a.y = v[1]; b.x = v[2]; b.y = v[3]; c.x = v[4]; c.y = v[5]; d.x = v[6]; d.y = v[7];
  return make_uint4(pack_bf16x2(a), pack_bf16x2(b), pack_bf16x2(c), pack_bf16x2(d));
}

template <typename T, bool BWD, int NDB, bool BIAS>
__global__ __launch_bounds__(512) void attn_q_kernel(const AttnArgs a) {
  constexpr int DP = 32 * NDB, KS = 2 * NDB, LD = DP * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NW = (int)(blockDim.x >> 6);          // 1 .. 8 waves: attn_waves()
  const int l31 = lane & 31, lh = lane >> 5;
  const int S = a.S, D = a.D, H = a.H;
  const int b = blockIdx.y / H, h = blockIdx.y - b * H;
  const int SP = (S + 31) / 32 * 32;
  unsigned char* sk = smem;
  unsigned char* sv = smem + SP * LD;
  const long row_ld = 3l * H * D;
  const T* qkv = reinterpret_cast<const T*>(a.qkv) + (long)b * S * row_ld + (long)h * D;
  stage_rows<T>(sk, LD, qkv + (long)H * D, row_ld, 0, SP, S, D, DP, tid, 64 * NW);
  stage_rows<T>(sv, LD, qkv + 2l * H * D, row_ld, 0, SP, S, D, DP, tid, 64 * NW);

  const int q = blockIdx.x * (32 * NW) + wave * 32 + l31;       // this lane's query row
  const bool qok = q < S;
  // Q (and dO) fragments straight from global memory: row q, 8 elements at 16*ks + 8*lh
  uint4 qf[KS], gf[BWD ? KS : 1];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = make_uint4(0u, 0u, 0u, 0u);
    if (BWD) gf[ks] = make_uint4(0u, 0u, 0u, 0u);
    const int c = ks * 16 + lh * 8;
    if (qok && c < D) {
      qf[ks] = *reinterpret_cast<const uint4*>(qkv + (long)q * row_ld + c);
      if (BWD) gf[ks] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.dout) +
                                                        ((long)b * S + q) * H * D + (long)h * D + c);
    }
  }
  const float scale = a.scale[h];
  const float sl2 = scale * kLog2e;
  const float* bias = BIAS ? a.bias + ((long)(b % a.nbias) * H + h) * S * S + (long)(qok ? q : 0) * S : nullptr;
  float lse2 = 0.f, delta_q = 0.f;
  if (BWD && qok) {
    lse2 = a.lse[((long)b * H + h) * S + q] * kLog2e;
    delta_q = a.delta[((long)b * H + h) * S + q];
  }
  // per-lane fragment bases: direct reads address row l31 (+ 32 per key block), transposed reads the 4 x 16 sub-block
  const unsigned char* kdir = sk + l31 * LD + lh * 16;
  const unsigned char* vdir = sv + l31 * LD + lh * 16;
  const int g4 = lane >> 4, gi = lane & 15;
  const int troff = (4 * (g4 >> 1) + (gi >> 2)) * LD + (16 * (g4 & 1) + 4 * (gi & 3)) * 2;
  const unsigned char* ttr = (BWD ? sk : sv) + troff;
  __syncthreads();

  f32x16_t acc[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int nkb = SP / 32;
  const bool ragged = (S & 31) != 0;

  for (int kb = 0; kb < nkb; ++kb) {
    const int boff = kb * 32 * LD;
    // S^T block: keys <-> registers, queries <-> l31
    f32x16_t sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      AMma<T>::run(sc, *reinterpret_cast<const uint4*>(kdir + boff + ks * 32), qf[ks]);
    float p[16];
    if (BIAS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + reg_row(r, lh);
        p[r] = fmaf(sc[r], sl2, (key < S ? bias[key] : 0.f) * kLog2e);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) p[r] = sc[r] * sl2;
    }
    const bool tail = ragged && kb == nkb - 1;        // workgroup-uniform
    if (!BWD) {
      if (tail) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + reg_row(r, lh) >= S) p[r] = -INFINITY;
      }
      float mblk = p[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mblk = fmaxf(mblk, p[r]);
      mblk = fmaxf(mblk, __shfl_xor(mblk, 32, 64));
      // a finite floor: a fully masked block (never the first) then gives exp2(-inf) = 0 without a select
      const float m_new = fmaxf(fmaxf(m_run, mblk), -3.0e38f);
      if (__any(m_new != m_run)) {
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[d][r] *= alpha;
        m_run = m_new;
      }
      float lsum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = __builtin_amdgcn_exp2f(p[r] - m_run);
        lsum += p[r];
      }
      l_run += lsum;
    } else {
      // P from the saved log-sum-exp, dP^T = V dO^T, dS^T = P (dP - delta)
      f32x16_t dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        AMma<T>::run(dp, *reinterpret_cast<const uint4*>(vdir + boff + ks * 32), gf[ks]);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(p[r] - lse2);
        p[r] = pr * (dp[r] - delta_q) * scale;           // gradient of the raw Q.K product
      }
      // lanes without a query have zero operands, lse2 = delta = 0: p = 0 already.  Padding keys (zero K rows) would
      // contribute exp2(-lse2) * dp: masked here, in the last block only
      if (tail) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + reg_row(r, lh) >= S) p[r] = 0.f;
      }
    }
    // acc[d][query] += sum_key p[key] * (V | K)[key][d]: the probabilities are the B operand as they sit
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float half[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) half[j] = p[8 * s2 + j];
      const uint4 pf = pack8_fast<T>(half);
#pragma unroll
      for (int d = 0; d < NDB; ++d) {
        const unsigned char* base = ttr + boff + 16 * s2 * LD + d * 64;
        const uint2 lo = attn_tr_read(base), hi = attn_tr_read(base + 8 * LD);
        AMma<T>::run(acc[d], make_uint4(lo.x, lo.y, hi.x, hi.y), pf);
      }
    }
  }

  if (!BWD) {
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (qok) {
      T* o = reinterpret_cast<T*>(a.out) + ((long)b * S + q) * H * D + (long)h * D;
#pragma unroll
      for (int d = 0; d < NDB; ++d)
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
      if (lh == 0) a.lse[((long)b * H + h) * S + q] = m_run * kLn2 + __logf(l_tot);
    }
  } else if (qok) {
    T* o = reinterpret_cast<T*>(a.dqkv) + ((long)b * S + q) * row_ld + (long)h * D;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
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
// Workgroup = NW waves = 32 * NW keys of head h; it walks batch entries b = split, split + nsplit, ...  Q and dO of the
// current (b, h) are staged in LDS (all S tokens), the wave's own K / V rows are read as fragments from global memory.
// DBIAS (S <= 64: one workgroup covers every key): dS accumulates in registers over the strip of batch entries.
// No masks: padded query rows are zero rows of Q and dO with lse = delta = 0 (P = 1 meets dO = 0, dS = 0), lanes without
// a key compute values that are never stored.  lse is staged pre-multiplied by log2(e).
template <typename T, bool DBIAS, int NDB, bool BIAS>
__global__ __launch_bounds__(DBIAS ? 128 : (NDB <= 2 ? 512 : 256), DBIAS && NDB <= 2 ? 2 : 1) void attn_kv_kernel(const AttnArgs a) {
  constexpr int DP = 32 * NDB, KS = 2 * NDB, LD = DP * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NW = (int)(blockDim.x >> 6);          // DBIAS: 2 waves (S <= 64); otherwise attn_waves()
  const int l31 = lane & 31, lh = lane >> 5;
  const int S = a.S, D = a.D, H = a.H;
  const int split = blockIdx.y / H, h = blockIdx.y - split * H;
  const int SP = (S + 31) / 32 * 32;
  unsigned char* sq = smem;
  unsigned char* sg = smem + SP * LD;
  float* s_lse = reinterpret_cast<float*>(smem + 2 * SP * LD);
  float* s_delta = s_lse + SP;
  const long row_ld = 3l * H * D;
  const int key = blockIdx.x * (32 * NW) + wave * 32 + l31;     // this lane's key row
  const bool kok = key < S;
  const float scale = a.scale[h];
  const float sl2 = scale * kLog2e;
  const unsigned char* qdir = sq + l31 * LD + lh * 16;
  const unsigned char* gdir = sg + l31 * LD + lh * 16;
  const int g4 = lane >> 4, gi = lane & 15;
  const int troff = (4 * (g4 >> 1) + (gi >> 2)) * LD + (16 * (g4 & 1) + 4 * (gi & 3)) * 2;

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
      s_lse[i] = i < S ? a.lse[((long)b * H + h) * S + i] * kLog2e : 0.f;
      s_delta[i] = i < S ? a.delta[((long)b * H + h) * S + i] : 0.f;
    }
    uint4 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = vf[ks] = make_uint4(0u, 0u, 0u, 0u);
      const int c = ks * 16 + lh * 8;
      if (kok && c < D) {
        kf[ks] = *reinterpret_cast<const uint4*>(qkv + (long)key * row_ld + (long)H * D + c);
        vf[ks] = *reinterpret_cast<const uint4*>(qkv + (long)key * row_ld + 2l * H * D + c);
      }
    }
    const float* bias = BIAS ? a.bias + ((long)(b % a.nbias) * H + h) * S * S : nullptr;
    __syncthreads();

    f32x16_t dk[NDB], dv[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) dk[d][r] = dv[d][r] = 0.f;

    for (int qb = 0; qb < SP / 32; ++qb) {
      const int boff = qb * 32 * LD;
      // S block: queries <-> registers, keys <-> l31
      f32x16_t sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = dp[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        AMma<T>::run(sc, *reinterpret_cast<const uint4*>(qdir + boff + ks * 32), kf[ks]);
        AMma<T>::run(dp, *reinterpret_cast<const uint4*>(gdir + boff + ks * 32), vf[ks]);
      }
      float p[16], ds[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // registers 4g .. 4g+3 are query rows qb*32 + 8g + 4lh + 0..3: one 16-byte read of each per-row table
        const float4 l4 = *reinterpret_cast<const float4*>(s_lse + qb * 32 + 8 * g + 4 * lh);
        const float4 d4 = *reinterpret_cast<const float4*>(s_delta + qb * 32 + 8 * g + 4 * lh);
        const float lr[4] = {l4.x, l4.y, l4.z, l4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          float s = fmaf(sc[r], sl2, -lr[j]);
          if (BIAS) {
            const int qrow = qb * 32 + 8 * g + 4 * lh + j;
            s = fmaf((qrow < S && kok) ? bias[(long)qrow * S + key] : 0.f, kLog2e, s);
          }
          const float pr = __builtin_amdgcn_exp2f(s);
          p[r] = pr;
          ds[r] = pr * (dp[r] - dr[j]);
          if (DBIAS) dsacc[qb & 1][r] += ds[r];
          ds[r] *= scale;
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float hp[8], hd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          hp[j] = p[8 * s2 + j];
          hd[j] = ds[8 * s2 + j];
        }
        const uint4 pf = pack8_fast<T>(hp), df = pack8_fast<T>(hd);
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
          const unsigned char* bg = sg + troff + boff + 16 * s2 * LD + d * 64;
          const unsigned char* bq = sq + troff + boff + 16 * s2 * LD + d * 64;
          const uint2 glo = attn_tr_read(bg), ghi = attn_tr_read(bg + 8 * LD);
          const uint2 qlo = attn_tr_read(bq), qhi = attn_tr_read(bq + 8 * LD);
          AMma<T>::run(dv[d], make_uint4(glo.x, glo.y, ghi.x, ghi.y), pf);   // dV += P^T dO
          AMma<T>::run(dk[d], make_uint4(qlo.x, qlo.y, qhi.x, qhi.y), df);   // dK += dS^T Q
        }
      }
    }
    if (kok) {
      T* o = reinterpret_cast<T*>(a.dqkv) + ((long)b * S + key) * row_ld + (long)h * D;
#pragma unroll
      for (int d = 0; d < NDB; ++d)
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

// delta[b][h][s] = sum_d dO * O.  One lane per 16-byte chunk: the D / 8 lanes of a row read one contiguous row segment
// (a lane per row made every load instruction touch 64 different 128-byte lines: 80 us for 78 MB), partial dot
// products meet in LDS and the first lanes of the workgroup add the D / 8 partials of their row.
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout,
                                                         float* __restrict__ delta, long rows_bs, int H, int S, int D) {
  __shared__ float part[256];
  const int lpr = D >> 3;                        // lanes per row (D is a multiple of 8, <= 128)
  const int rpw = 256 / lpr;                     // rows per workgroup
  const int t = threadIdx.x;
  const int rl = t / lpr, cl = t - rl * lpr;
  const long row = (long)blockIdx.x * rpw + rl;  // (b, s, h) flattened: [B*S][H] rows of D elements, contiguous
  float acc = 0.f;
  if (rl < rpw && row < rows_bs * H) {
    Vec16<T> vo, vg;
    vo.load(o + row * D + cl * 8);
    vg.load(dout + row * D + cl * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += vo.get(j) * vg.get(j);
  }
  part[t] = acc;
  __syncthreads();
  if (t < rpw) {
    const long r = (long)blockIdx.x * rpw + t;
    if (r < rows_bs * H) {
      float sum = 0.f;
      for (int j = 0; j < lpr; ++j) sum += part[t * lpr + j];
      const long bs = r / H;
      const int h = (int)(r - bs * H);
      const long b = bs / S;
      const int sidx = (int)(bs - b * S);
      delta[(b * H + h) * S + sidx] = sum;
    }
  }
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

// The same with one lane per 16-byte chunk (LPR = D / 8 lanes per vector, a power of two): the lanes of a vector read one
// contiguous run and meet through LPR-wide butterflies; every element is loaded once.  The thread-per-vector kernel
// above touched 64 different lines per load instruction: 190-210 us per Swin-V2 layer, 4x the bandwidth bound.
template <typename T, bool BWD, int LPR>
__global__ __launch_bounds__(256) void qk_normalize_rows_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                T* __restrict__ out, const float* __restrict__ mult,
                                                                float* __restrict__ inv, float* __restrict__ dmult_part,
                                                                long tokens, int H) {
  constexpr int D = LPR * 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long rows = tokens * 3 * H;
  const long row = i / LPR;
  const int cl = (int)(i & (LPR - 1));
  const bool valid = row < rows;                  // whole groups of LPR lanes are valid or not
  const long rr = valid ? row : 0;
  const int h = (int)(rr % H);
  const int which = (int)((rr / H) % 3);
  const long tok = rr / (3l * H);
  const long off = rr * D + cl * 8;
  Vec16<T> v, g, o;
  v.raw = g.raw = make_uint4(0u, 0u, 0u, 0u);
  if (valid) {
    if (!BWD || which != 2) v.load(x + off);
    if (BWD) g.load(dy + off);
  }
  const float m = which == 0 ? mult[h] : 1.f;
  if (!BWD) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += v.get(j) * v.get(j);
#pragma unroll
    for (int sft = 1; sft < LPR; sft <<= 1) ss += __shfl_xor(ss, sft, 64);
    const float r = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    if (!valid) return;
    if (which == 2) { v.store(out + off); return; }
    if (cl == 0) inv[(tok * 2 + which) * H + h] = r;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.set(j, v.get(j) * r * m);
    o.store(out + off);
  } else {
    const float r = (valid && which != 2) ? inv[(tok * 2 + which) * H + h] : 0.f;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dot += g.get(j) * v.get(j) * r;
#pragma unroll
    for (int sft = 1; sft < LPR; sft <<= 1) dot += __shfl_xor(dot, sft, 64);
    if (!valid) return;
    if (which == 2) { g.store(out + off); return; }
    if (which == 0 && cl == 0) dmult_part[tok * H + h] = dot;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float d = m * r * (g.get(j) - v.get(j) * r * dot);
      if (TypeInfo<T>::kCode == kF16) d = fminf(fmaxf(d, -60000.f), 60000.f);     // see qk_normalize_kernel
      o.set(j, d);
    }
    o.store(out + off);
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

// waves per workgroup: the 32-token blocks of a sequence in one workgroup when there are at most 8, else split evenly
int attn_waves(int S, int maxw = 8) {
  const int nblk = (S + 31) / 32, nwg = (nblk + maxw - 1) / maxw;
  return (nblk + nwg - 1) / nwg;
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
  if (D == 32 || D == 64 || D == 128) {          // one lane per 16-byte chunk
    const long nb = (n * (D / 8) + 255) / 256;
    if (nb < 2147483647l) {
#define DPA_QKN_ROWS(T, LPR)                                                                                          \
  if (backward)                                                                                                       \
    hipLaunchKernelGGL((qk_normalize_rows_kernel<T, true, LPR>), dim3((unsigned)nb), dim3(256), 0, s, (const T*)x,    \
                       (const T*)dy, (T*)out, mult, inv, dmult_part, tokens, H);                                      \
  else                                                                                                                \
    hipLaunchKernelGGL((qk_normalize_rows_kernel<T, false, LPR>), dim3((unsigned)nb), dim3(256), 0, s, (const T*)x,   \
                       (const T*)dy, (T*)out, mult, inv, dmult_part, tokens, H);
#define DPA_QKN_ROWS_D(T)                                                                                             \
  if (D == 32) { DPA_QKN_ROWS(T, 4) } else if (D == 64) { DPA_QKN_ROWS(T, 8) } else { DPA_QKN_ROWS(T, 16) }
      if (dtype == kBF16) { DPA_QKN_ROWS_D(bf16_t) } else { DPA_QKN_ROWS_D(f16_t) }
#undef DPA_QKN_ROWS_D
#undef DPA_QKN_ROWS
      DPA_CHECK_HIP(hipGetLastError());
      return;
    }
  }
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
  const int nw = attn_waves(S);
  dim3 grid(ceil_div(S, 32 * nw), B * H);
#define DPA_ATTN_FWD3(T, NDB, BIAS)                                                            \
  set_lds(attn_q_kernel<T, false, NDB, BIAS>, lds);                                            \
  hipLaunchKernelGGL((attn_q_kernel<T, false, NDB, BIAS>), grid, dim3(64 * nw), lds, s, a);
#define DPA_ATTN_FWD2(T, NDB)                                                                  \
  if (bias != nullptr) { DPA_ATTN_FWD3(T, NDB, true) } else { DPA_ATTN_FWD3(T, NDB, false) }
#define DPA_ATTN_FWD(T)                                                                        \
  if (D <= 32) { DPA_ATTN_FWD2(T, 1) } else if (D <= 64) { DPA_ATTN_FWD2(T, 2) }               \
  else if (D <= 96) { DPA_ATTN_FWD2(T, 3) } else { DPA_ATTN_FWD2(T, 4) }
  if (dtype == kBF16) { DPA_ATTN_FWD(bf16_t) } else { DPA_ATTN_FWD(f16_t) }
#undef DPA_ATTN_FWD3
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
  const int dgrid = (int)ceil_div(rows * H, (long)(256 / (D / 8)));
  const int lds_q = lds_bytes(S, D, false), lds_kv = lds_bytes(S, D, true);
  // dK / dV with D > 64 needs more than 256 registers: at most 4 waves per workgroup there (one per SIMD)
  const int nw = attn_waves(S), nwk = attn_waves(S, D <= 64 ? 8 : 4);
  dim3 gq(ceil_div(S, 32 * nw), B * H), gkv(ceil_div(S, 32 * nwk), a.nsplit * H);
#define DPA_ATTN_BWD3(T, NDB, BIAS)                                                                                  \
  set_lds(attn_q_kernel<T, true, NDB, BIAS>, lds_q);                                                                 \
  hipLaunchKernelGGL((attn_q_kernel<T, true, NDB, BIAS>), gq, dim3(64 * nw), lds_q, s, a);                           \
  if (dbias != nullptr) {                                                                                            \
    set_lds(attn_kv_kernel<T, true, NDB, true>, lds_kv);                                                             \
    hipLaunchKernelGGL((attn_kv_kernel<T, true, NDB, true>), dim3(1, a.nsplit * H), dim3(128), lds_kv, s, a);        \
  } else {                                                                                                           \
    set_lds(attn_kv_kernel<T, false, NDB, BIAS>, lds_kv);                                                            \
    hipLaunchKernelGGL((attn_kv_kernel<T, false, NDB, BIAS>), gkv, dim3(64 * nwk), lds_kv, s, a);                    \
  }
#define DPA_ATTN_BWD2(T, NDB)                                                                                        \
  if (bias != nullptr) { DPA_ATTN_BWD3(T, NDB, true) } else { DPA_ATTN_BWD3(T, NDB, false) }
#define DPA_ATTN_BWD(T)                                                                                              \
  hipLaunchKernelGGL(attn_delta_kernel<T>, dim3(dgrid), dim3(256), 0, s, (const T*)out, (const T*)dout, delta, rows, H, \
                     S, D);                                                                                          \
  if (D <= 32) { DPA_ATTN_BWD2(T, 1) } else if (D <= 64) { DPA_ATTN_BWD2(T, 2) }                                     \
  else if (D <= 96) { DPA_ATTN_BWD2(T, 3) } else { DPA_ATTN_BWD2(T, 4) }
  if (dtype == kBF16) { DPA_ATTN_BWD(bf16_t) } else { DPA_ATTN_BWD(f16_t) }
#undef DPA_ATTN_BWD3
#undef DPA_ATTN_BWD2
#undef DPA_ATTN_BWD
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
