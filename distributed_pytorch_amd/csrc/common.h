// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Everything here assumes wave64 and the gfx950 MFMA operand layouts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdexcept>
#include <string>

namespace dpa {

enum DType : int { kF32 = 0, kBF16 = 1, kF16 = 2 };

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct bf16_t { uint16_t v; };
struct f16_t { _Float16 v; };

// ---- scalar conversions ------------------------------------------------------------------
__device__ __forceinline__ float to_float(float x) { return x; }
__device__ __forceinline__ float to_float(bf16_t x) { return __uint_as_float(((uint32_t)x.v) << 16); }
__device__ __forceinline__ float to_float(f16_t x) { return (float)x.v; }

template <typename T> __device__ __forceinline__ T from_float(float x);
template <> __device__ __forceinline__ float from_float<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_float<bf16_t>(float x) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
  __bf16 b = (__bf16)x;
  bf16_t r;
  __builtin_memcpy(&r.v, &b, 2);
  return r;
}
template <> __device__ __forceinline__ f16_t from_float<f16_t>(float x) {
  f16_t r;
  r.v = (_Float16)x;
  return r;
}

template <typename T> struct TypeInfo;
template <> struct TypeInfo<float> { static constexpr int kVec = 4; static constexpr int kCode = kF32; };
template <> struct TypeInfo<bf16_t> { static constexpr int kVec = 8; static constexpr int kCode = kBF16; };
template <> struct TypeInfo<f16_t> { static constexpr int kVec = 8; static constexpr int kCode = kF16; };

// 16-byte vector <-> fp32 lanes
template <typename T> struct Vec16 {
  static constexpr int N = 16 / sizeof(T);
  uint4 raw;
  __device__ __forceinline__ void load(const void* p) { raw = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void store(void* p) const { *reinterpret_cast<uint4*>(p) = raw; }
  // streaming variants (nt): data that is touched once and is far larger than the caches
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  __device__ __forceinline__ void load_nt(const void* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    raw = make_uint4(v.x, v.y, v.z, v.w);
  }
  __device__ __forceinline__ void store_nt(void* p) const {
    u32x4_t v = {raw.x, raw.y, raw.z, raw.w};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p));
  }
  __device__ __forceinline__ float get(int i) const {
    T t;
    __builtin_memcpy(&t, reinterpret_cast<const char*>(&raw) + i * sizeof(T), sizeof(T));
    return to_float(t);
  }
  __device__ __forceinline__ void set(int i, float f) {
    T t = from_float<T>(f);
    __builtin_memcpy(reinterpret_cast<char*>(&raw) + i * sizeof(T), &t, sizeof(T));
  }
};

// ---- relu(scale * x + shift) on one 16-byte chunk (BatchNorm + ReLU applied while an operand tile is staged) -----
// Bit-identical to the stand-alone bn_apply pass: fp32 fma, one rounding to T, ReLU.  The bf16 version is written
// for the VALU budget of the GEMM loops that host it: packed fp32 fma (v_pk_fma_f32), packed convert
// (v_cvt_pk_bf16_f32) and the ReLU as a packed signed-16-bit max with zero on the ROUNDED pair (a negative bf16 has
// its sign bit set, i.e. is a negative int16; -0.0 becomes +0.0 like fmaxf(x, 0) does) -- 5 VALU ops per two elements.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t bn_relu_pair_bf16(uint32_t w, f32x2_t sc, f32x2_t sh) {
  f32x2_t x;
  x.x = __uint_as_float(w << 16);
  x.y = __uint_as_float(w & 0xffff0000u);
  const f32x2_t y = x * sc + sh;
  const bf16x2_t p = __builtin_convertvector(y, bf16x2_t);
  s16x2_t v;
  __builtin_memcpy(&v, &p, 4);
  const s16x2_t zero = {0, 0};
  v = __builtin_elementwise_max(v, zero);
  uint32_t r;
  __builtin_memcpy(&r, &v, 4);
  return r;
}

// ---- packed bf16 pair arithmetic for the streaming epilogues (v_pk_add_f32 / v_pk_fma_f32 / v_cvt_pk_bf16_f32): the
// epilogue of a fused accumulate + BN-backward launch was VALU bound at ~22 instructions per element when every element
// went through scalar get / select / set (bench/experiments/lean_racc_ablate.py); pairs take ~8.
__device__ __forceinline__ f32x2_t unpack_bf16x2(uint32_t w) {
  f32x2_t x;
  x.x = __uint_as_float(w << 16);
  x.y = __uint_as_float(w & 0xffff0000u);
  return x;
}
__device__ __forceinline__ uint32_t pack_bf16x2(f32x2_t y) {
  const bf16x2_t p = __builtin_convertvector(y, bf16x2_t);
  uint32_t r;
  __builtin_memcpy(&r, &p, 4);
  return r;
}
// bits 2j, 2j+1 of a mask byte -> an AND mask for the bf16 pair j (low half = element 2j)
__device__ __forceinline__ uint32_t pair_mask(uint32_t bits, int j) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 2 * j, 1);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 2 * j + 1, 1);
  return (lo & 0x0000ffffu) | (hi & 0xffff0000u);        // v_bfi_b32
}

// (int16 view of a bf16 pair) > 0 per half -> AND mask 0xffff / 0: +0, -0, negatives and NaNs with the sign bit give 0
__device__ __forceinline__ uint32_t pair_positive_mask(uint32_t w) {
  s16x2_t v;
  __builtin_memcpy(&v, &w, 4);
  const s16x2_t zero = {0, 0}, one = {1, 1};
  v = __builtin_elementwise_min(__builtin_elementwise_max(v, zero), one);      // 1 where > 0
  typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
  u16x2_t u;
  __builtin_memcpy(&u, &v, 4);
  const u16x2_t ff = {0xffff, 0xffff};
  u = u * ff;                                                                   // v_pk_mul_lo_u16
  uint32_t r;
  __builtin_memcpy(&r, &u, 4);
  return r;
}

template <typename T>
__device__ __forceinline__ uint4 bn_relu_chunk(const uint4& raw, const float (&sc)[16 / sizeof(T)],
                                               const float (&sh)[16 / sizeof(T)]) {
  if constexpr (sizeof(T) == 2 && TypeInfo<T>::kCode == kBF16) {
    uint4 r;
    r.x = bn_relu_pair_bf16(raw.x, f32x2_t{sc[0], sc[1]}, f32x2_t{sh[0], sh[1]});
    r.y = bn_relu_pair_bf16(raw.y, f32x2_t{sc[2], sc[3]}, f32x2_t{sh[2], sh[3]});
    r.z = bn_relu_pair_bf16(raw.z, f32x2_t{sc[4], sc[5]}, f32x2_t{sh[4], sh[5]});
    r.w = bn_relu_pair_bf16(raw.w, f32x2_t{sc[6], sc[7]}, f32x2_t{sh[6], sh[7]});
    return r;
  } else {
    Vec16<T> v;
    v.raw = raw;
#pragma unroll
    for (int j = 0; j < (int)(16 / sizeof(T)); ++j) v.set(j, fmaxf(v.get(j) * sc[j] + sh[j], 0.f));
    return v.raw;
  }
}

// ---- fast unsigned division by a runtime constant (host computes the magic) ---------------
struct FastDiv {
  uint32_t d, magic, shift;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  if (d == 0) throw std::invalid_argument("fastdiv: zero divisor (empty output grid?)");
  if (d == 1) {
    f.magic = 0;
    f.shift = 0;
    return f;
  }
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;  // ceil(log2 d)
  uint64_t m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
  f.magic = (uint32_t)m;
  f.shift = l;
  return f;
}
// exact for all n < 2^31
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  if (f.d == 1) return n;
  uint32_t t = __umulhi(n, f.magic);
  return (t + ((n - t) >> 1)) >> (f.shift - 1);
}

// ---- wave / block reductions ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD (round-robin
// dispatch), so give each XCD one contiguous chunk of the logical tile space (L2 reuse of the
// operand panels shared by neighbouring tiles).  Speed only, never correctness.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblk) {
  const uint32_t q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  const uint32_t base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + i;
}

// ---- smooth activations: 1 sigmoid, 2 SiLU, 3 hard-sigmoid, 4 hard-swish, 5 GELU (erf) -------------------------
// sigmoid through v_rcp_f32 (1 ulp): the IEEE division costs more VALU than the rest of the activation together
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float act_value(int act, float x) {
  switch (act) {
    case 1: return fast_sigmoid(x);
    case 2: return x * fast_sigmoid(x);
    case 3: return fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
    case 5: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
    default: return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
  }
}
__device__ __forceinline__ float act_slope(int act, float x) {
  switch (act) {
    case 1: { const float s = fast_sigmoid(x); return s * (1.f - s); }
    case 2: { const float s = fast_sigmoid(x); return s * (1.f + x * (1.f - s)); }
    case 3: return (x > -3.f && x < 3.f) ? (1.f / 6.f) : 0.f;
    case 5: return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
    default: return x <= -3.f ? 0.f : (x >= 3.f ? 1.f : (2.f * x + 3.f) * (1.f / 6.f));   // ATen's convention at +-3
  }
}

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace dpa

// dtype dispatch: binds `T` inside the statement block
#define DPA_DISPATCH_DTYPE(dtype, ...)                                   \
  switch (dtype) {                                                       \
    case ::dpa::kF32: { using T = float; __VA_ARGS__; break; }           \
    case ::dpa::kBF16: { using T = ::dpa::bf16_t; __VA_ARGS__; break; }  \
    case ::dpa::kF16: { using T = ::dpa::f16_t; __VA_ARGS__; break; }    \
    default: throw std::invalid_argument("unsupported dtype");           \
  }

#define DPA_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess)                                                                \
      throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e));      \
  } while (0)
