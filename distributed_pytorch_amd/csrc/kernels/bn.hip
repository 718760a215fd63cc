// BatchNorm (+ residual add, + ReLU) forward / backward for NHWC activations on gfx950.
//
// NHWC makes the channel the fast axis, so per-channel statistics are COLUMN reductions: every
// thread owns one 16-byte channel chunk and walks rows; partial sums leave each workgroup as an
// fp32 slab [block][2][C] that a tiny finalize kernel folds in a fixed order (deterministic, no
// float atomics).  Statistics are always fp32 (double in the final fold).
//
// forward  (train): stats -> finalize (mean, invstd, scale/shift, running stats) -> apply
// forward  (eval) : eval_coeffs -> apply
// backward        : bwd_reduce (sum dy, sum dy*xhat with the ReLU mask recomputed from y)
//                   -> bwd_finalize -> bwd_apply (dx, optional masked copy for the residual branch)
// SyncBatchNorm reuses the same kernels with an all-reduce between the reduce and finalize steps
// (parallel/sync_batchnorm.py).
//
// Replaces cuDNN BatchNorm + ATen ReLU/add kernels and Apex's syncbn extension
// (reference imagenet_ddp_apex.py:146-148) -- SURVEY.md N-K4..N-K7.
#include "../common.h"
#include "kernels.h"
#include <stdexcept>
#include <type_traits>

namespace dpa {
namespace {

constexpr int kBnThreads = 256;

// MODE 0: (sum x, sum x^2).  MODE 1: (sum dy_m, sum dy_m * xhat) with dy_m = dy * (y > 0) if relu.
// SMOOTH (MODE 1 only): the BN is followed by a smooth activation ((relu >> 4) & 7, see common.h act_slope) fused into
// the apply pass; dy is multiplied by its slope at z = scale*x + shift, recomputed here.  A separate instantiation, so
// the ReLU / no-activation kernels of the ResNet hot path are not touched.
// RM (the ReLU-mask source, relu & 3 below) and NT (streaming loads) are compile-time: as run-time branches they doubled
// the load paths and left the bf16 kernel at 194 VGPRs = 2 waves per SIMD, 3.1 TB/s; sized per variant it holds 4.
template <typename T, int MODE, bool SMOOTH = false, int RM = 0, bool NT = false>
__global__ __launch_bounds__(kBnThreads, 4) void col_reduce_kernel(const T* __restrict__ a, const T* __restrict__ x,
                                                                const T* __restrict__ y, long M, int C,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int relu,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                float* __restrict__ partial, long rows_per_block) {
  constexpr int VEC = 16 / (int)sizeof(T);
  __shared__ float red[2][kBnThreads * VEC];
  const int cc_total = C / VEC;
  const int cc0 = blockIdx.y * kBnThreads;
  const int ccb = min(cc_total - cc0, kBnThreads);        // chunk columns in this block
  const int rp = kBnThreads / ccb;                          // row lanes
  const int tid = threadIdx.x;
  const int col = tid % ccb, rl = tid / ccb;
  const bool active = rl < rp;
  const int c0 = (cc0 + col) * VEC;

  // relu & 3: 0 none, 1 mask = (y > 0) read from the saved output, 2 mask = (scale*x + shift > 0) recomputed
  // from x (no residual): the output tensor is not read at all, 3 mask = one BIT per element written by the
  // forward apply pass (`y` then points at bytes, one per 16-byte chunk): 1/16 of the output's bytes.
  // relu & 4: ReLU6, the mask also needs y < 6.
  constexpr int rm = RM;
  constexpr bool ntl = NT;                     // streaming loads (tuning hook)
  const bool clip = (relu & 4) != 0;
  constexpr bool kCoef = MODE == 1 && (RM == 2 || SMOOTH);
  float s1[VEC], s2[VEC], mu[VEC], is[VEC], sc[kCoef ? VEC : 1], sh[kCoef ? VEC : 1];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    s1[j] = s2[j] = 0.f;
    mu[j] = 0.f;
    is[j] = 1.f;
    if constexpr (kCoef) sc[j] = sh[j] = 0.f;
  }
  if (MODE == 1 && active) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      mu[j] = mean[c0 + j];
      is[j] = invstd[c0 + j];
      if constexpr (kCoef) {
        sc[j] = scale[c0 + j];
        sh[j] = shift[c0 + j];
      }
    }
  }
  const long r_begin = (long)blockIdx.x * rows_per_block;
  const long r_end = min(M, r_begin + rows_per_block);
  // MODE 0 leaves (sum, M2 about the block mean, rows) -- Chan's partials -- not a raw sum of squares (which cancels
  // catastrophically in fp32 when |mean| >> std): sums of (v - k), (v - k)^2 with k = the block's first row.
  float ksh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) ksh[j] = 0.f;
  if (MODE == 0 && active && r_begin < r_end) {
    Vec16<T> k0;
    k0.load(a + (size_t)r_begin * C + c0);
#pragma unroll
    for (int j = 0; j < VEC; ++j) ksh[j] = k0.get(j);
  }
  auto accumulate = [&](const Vec16<T>& va, const Vec16<T>& vx, const Vec16<T>& vy, uint32_t bits) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float f = va.get(j) - ksh[j];
        s1[j] += f;
        s2[j] += f * f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float d = va.get(j);
        const float xv = vx.get(j);
        if constexpr (rm == 3) {
          if (!((bits >> j) & 1u)) d = 0.f;
        } else if constexpr (rm == 1) {
          const float yv = vy.get(j);
          if (!(yv > 0.f) || (clip && !(yv < 6.f))) d = 0.f;
        } else if constexpr (rm == 2) {
          const float yv = to_float(from_float<T>(sc[kCoef ? j : 0] * xv + sh[kCoef ? j : 0]));
          if (!(yv > 0.f) || (clip && !(yv < 6.f))) d = 0.f;
        }
        if constexpr (SMOOTH) d *= act_slope((relu >> 4) & 7, sc[kCoef ? j : 0] * xv + sh[kCoef ? j : 0]);
        const float xh = (xv - mu[j]) * is[j];
        s1[j] += d;
        s2[j] += d * xh;
      }
    }
  };
  if (active) {
    long r = r_begin + rl;
    // UR rows per trip, every load issued before the first use (the summation order stays fixed); the smooth-activation
    // variants evaluate an exp + rcp per element and spill at 4 rows under the 128-VGPR cap: 2 rows there
    constexpr int UR = SMOOTH ? 2 : 4;
    for (; r + (UR - 1) * rp < r_end; r += UR * rp) {
      Vec16<T> va[UR], vx[MODE == 1 ? UR : 1], vy[MODE == 1 && rm == 1 ? UR : 1];
      uint32_t mb[UR] = {};
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const size_t off = (size_t)(r + u * rp) * C + c0;
        if constexpr (MODE == 1 && rm == 3) mb[u] = reinterpret_cast<const uint8_t*>(y)[off / VEC];
        if constexpr (ntl) {
          va[u].load_nt(a + off);
          if constexpr (MODE == 1) vx[u].load_nt(x + off);
          if constexpr (MODE == 1 && rm == 1) vy[u].load_nt(y + off);
        } else {
          va[u].load(a + off);
          if constexpr (MODE == 1) vx[u].load(x + off);
          if constexpr (MODE == 1 && rm == 1) vy[u].load(y + off);
        }
      }
#pragma unroll
      for (int u = 0; u < UR; ++u) accumulate(va[u], vx[MODE == 1 ? u : 0], vy[MODE == 1 && rm == 1 ? u : 0], mb[u]);
    }
    for (; r < r_end; r += rp) {
      const size_t off = (size_t)r * C + c0;
      Vec16<T> va, vx, vy;
      va.load(a + off);
      if constexpr (MODE == 1) vx.load(x + off);
      if constexpr (MODE == 1 && rm == 1) vy.load(y + off);
      uint32_t mb = 0u;
      if constexpr (MODE == 1 && rm == 3) mb = reinterpret_cast<const uint8_t*>(y)[off / VEC];
      accumulate(va, vx, vy, mb);
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    red[0][tid * VEC + j] = s1[j];
    red[1][tid * VEC + j] = s2[j];
  }
  __syncthreads();
  const float nb = r_end > r_begin ? (float)(r_end - r_begin) : 0.f;
  // thread t < ccb*VEC sums channel (t) over row lanes
  for (int t = tid; t < ccb * VEC; t += kBnThreads) {
    const int cchunk = t / VEC, j = t - cchunk * VEC;
    float a1 = 0.f, a2 = 0.f;
    for (int q = 0; q < rp; ++q) {
      a1 += red[0][(q * ccb + cchunk) * VEC + j];
      a2 += red[1][(q * ccb + cchunk) * VEC + j];
    }
    const int c = cc0 * VEC + t;
    if (MODE == 0) {
      const float k = nb > 0.f ? to_float(a[(size_t)r_begin * C + c]) : 0.f;
      partial[((size_t)blockIdx.x * 2 + 0) * C + c] = a1 + nb * k;
      partial[((size_t)blockIdx.x * 2 + 1) * C + c] = nb > 0.f ? fmaxf(a2 - a1 * a1 / nb, 0.f) : 0.f;
    } else {
      partial[((size_t)blockIdx.x * 2 + 0) * C + c] = a1;
      partial[((size_t)blockIdx.x * 2 + 1) * C + c] = a2;
    }
  }
  // rows of this part, behind the slab ([gridDim.x][2][C] floats, then [gridDim.x] counts)
  if (MODE == 0 && tid == 0 && blockIdx.y == 0) partial[(size_t)gridDim.x * 2 * C + blockIdx.x] = nb;
}

// SyncBatchNorm: Chan / Welford parallel merge of the ranks' local moments (Apex `welford_parallel`, reference
// imagenet_ddp_apex.py:146-148).  gathered: [W][2C + 1] = (mean_r[C], M2_r[C], n_r) per rank.
//   n = sum n_r;  mean = sum n_r mean_r / n;  M2 = sum M2_r + sum n_r (mean_r - mean)^2
// in double, with the deviations taken from the MERGED mean (no E[x^2] - mean^2 anywhere).
__global__ void bn_sync_merge_kernel(const float* __restrict__ gathered, int W, int C,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                     float momentum, float* running_mean, float* running_var,
                                     long* num_batches_tracked, float* mean, float* invstd, float* scale,
                                     float* shift, float* count_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;
  if (c >= C) return;
  const int ld = 2 * C + 1;
  double n = 0.0, sm = 0.0;
  for (int r = 0; r < W; ++r) {
    const double nr = (double)gathered[(size_t)r * ld + 2 * C];
    n += nr;
    sm += nr * (double)gathered[(size_t)r * ld + c];
  }
  const double m = n > 0.0 ? sm / n : 0.0;
  double m2 = 0.0;
  for (int r = 0; r < W; ++r) {
    const double nr = (double)gathered[(size_t)r * ld + 2 * C];
    const double d = (double)gathered[(size_t)r * ld + c] - m;
    m2 += (double)gathered[(size_t)r * ld + C + c] + nr * d * d;
  }
  const double var = n > 0.0 ? m2 / n : 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)m;
  invstd[c] = is;
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * is;
  shift[c] = b - (float)m * g * is;
  if (running_mean != nullptr) {
    const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
  if (c == 0 && count_out != nullptr) count_out[0] = (float)n;
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.f / sqrtf(rv[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * is;
  shift[c] = b - rm[c] * g * is;
}

// [nparts][2][C] -> o1[C], o2[C].  16 channel lanes x 16 part lanes per workgroup, four rows in flight per lane, the part
// lanes meet in LDS in a fixed order.  (One thread per channel walking 1024 rows serially took 200 us per bias
// gradient -- 30 % of a ConvNeXt step.)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int nparts, int C, float* o1,
                                                           float* o2) {
  __shared__ double red[2][16][17];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    int k = pl;
    for (; k + 48 < nparts; k += 64) {
      const float a0 = partial[((size_t)k * 2 + 0) * C + c], b0 = partial[((size_t)k * 2 + 1) * C + c];
      const float a1 = partial[((size_t)(k + 16) * 2 + 0) * C + c], b1 = partial[((size_t)(k + 16) * 2 + 1) * C + c];
      const float a2 = partial[((size_t)(k + 32) * 2 + 0) * C + c], b2 = partial[((size_t)(k + 32) * 2 + 1) * C + c];
      const float a3 = partial[((size_t)(k + 48) * 2 + 0) * C + c], b3 = partial[((size_t)(k + 48) * 2 + 1) * C + c];
      s1 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      s2 += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; k < nparts; k += 16) {
      s1 += (double)partial[((size_t)k * 2 + 0) * C + c];
      s2 += (double)partial[((size_t)k * 2 + 1) * C + c];
    }
  }
  red[0][pl][cl] = s1;
  red[1][pl][cl] = s2;
  __syncthreads();
  if (pl != 0 || c >= C) return;
  double t1 = 0.0, t2 = 0.0;
  for (int q = 0; q < 16; ++q) {
    t1 += red[0][q][cl];
    t2 += red[1][q][cl];
  }
  o1[c] = (float)t1;
  o2[c] = (float)t2;
}

// [nparts][2][C] -> [nout][2][C]: block (cx, j) folds parts [j*chunk, (j+1)*chunk) of 64 channels,
// 4 part lanes x 64 channel lanes (256-byte coalesced rows), fixed order => deterministic.
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ partial, int nparts, int C,
                                                            float* __restrict__ out, int chunk) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int p0 = blockIdx.y * chunk, p1 = min(nparts, p0 + chunk);
  float s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (int k = p0 + pl; k < p1; k += 4) {
      s1 += partial[((size_t)k * 2 + 0) * C + c];
      s2 += partial[((size_t)k * 2 + 1) * C + c];
    }
  red[0][pl][cl] = s1;
  red[1][pl][cl] = s2;
  __syncthreads();
  if (pl == 0 && c < C) {
    out[((size_t)blockIdx.y * 2 + 0) * C + c] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    out[((size_t)blockIdx.y * 2 + 1) * C + c] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  }
}

// ------------------------------------------------------------------------------------------------
// One launch: fold a long partial slab [nparts][2][C] AND finalize.  Workgroup (g, j) folds its chunk of parts
// for 64 channels (4 part lanes x 64 channel lanes, fixed order) into tmp[j][2][C]; the LAST workgroup of a
// column group to finish (agent-scope release + ticket, acquire by the last arriver -- cdna guide "in-launch
// split-K reduction") sums the nchunk rows in index order and writes the coefficients.  The ticket counter is
// returned to zero by that workgroup.  Replaces fold_partials + bn_finalize (forward) and fold_partials +
// bn_bwd_finalize_fused (backward): 2 of the 4 tiny launches per BatchNorm layer and direction.
// MODE 0: forward statistics -> mean, invstd, scale, shift (+ running stats).  MODE 1: backward sums ->
// dgamma, dbeta, apply coefficients [5][C].
#ifndef DPA_BN_FOLD_ROWS
#define DPA_BN_FOLD_ROWS 256
#endif
struct BnFinalizeArgs {
  const float* partial; int nparts, C, chunk, nchunk;
  double* tmp; int* counter;
  double count; float eps, momentum;
  const float* gamma; const float* beta;
  float* running_mean; float* running_var; long* num_batches_tracked;
  float* mean; float* invstd; float* scale; float* shift;          // MODE 0 outputs / MODE 1 inputs
  float* dgamma; float* dbeta; float* coef;                          // MODE 1 outputs
};

template <int MODE>
__global__ __launch_bounds__(256) void bn_fold_finalize_kernel(const BnFinalizeArgs a) {
  // sums are carried in double from the first add: var = E[x^2] - mean^2 cancels badly in float when the
  // slab is long, and this kernel is latency bound, not ALU bound
  __shared__ double red[2][4][64];
  __shared__ int is_last;
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int g = blockIdx.x, j = blockIdx.y;
  const int c = g * 64 + cl, C = a.C;
  const int p0 = j * a.chunk, p1 = min(a.nparts, p0 + a.chunk);
  // Forward statistics (MODE 0 / 2): part k holds (sum_k, M2_k about its own mean) and its row count n_k sits behind
  // the slab.  The additive quantities of Chan's merge are  S = sum_k  and  Q = M2_k + sum_k^2 / n_k  (in double:
  // M2_k is small and exact, the big term is formed in 53 bits), total M2 = Q - S^2 / n.
  const float* __restrict__ cnt = a.partial + (size_t)a.nparts * 2 * C;
  float inv_n_of = -1.f;       // the parts of a slab have the same row count except the last: one division, reused
  double inv_n = 0.0;
  auto second = [&](float nk, float sum_k, float b_k) -> double {
    if (MODE == 1) return (double)b_k;
    if (nk != inv_n_of) {
      inv_n_of = nk;
      inv_n = nk > 0.f ? 1.0 / (double)nk : 0.0;
    }
    return (double)b_k + (double)sum_k * (double)sum_k * inv_n;
  };
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    int k = p0 + pl;
    for (; k + 28 < p1; k += 32) {          // 8 independent row pairs in flight per lane
      float av[8], bv[8], nv[8];         // the row counts ride with the batch (a load inside second() is a dependent trip)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        av[u] = a.partial[((size_t)(k + 4 * u) * 2 + 0) * C + c];
        bv[u] = a.partial[((size_t)(k + 4 * u) * 2 + 1) * C + c];
        nv[u] = MODE == 1 ? 0.f : cnt[k + 4 * u];
      }
      s1 += (((double)av[0] + (double)av[1]) + ((double)av[2] + (double)av[3])) +
            (((double)av[4] + (double)av[5]) + ((double)av[6] + (double)av[7]));
      s2 += ((second(nv[0], av[0], bv[0]) + second(nv[1], av[1], bv[1])) +
             (second(nv[2], av[2], bv[2]) + second(nv[3], av[3], bv[3]))) +
            ((second(nv[4], av[4], bv[4]) + second(nv[5], av[5], bv[5])) +
             (second(nv[6], av[6], bv[6]) + second(nv[7], av[7], bv[7])));
    }
    for (; k < p1; k += 4) {
      const float a0 = a.partial[((size_t)k * 2 + 0) * C + c], b0 = a.partial[((size_t)k * 2 + 1) * C + c];
      s1 += (double)a0;
      s2 += second(MODE == 1 ? 0.f : cnt[k], a0, b0);
    }
  }
  red[0][pl][cl] = s1;
  red[1][pl][cl] = s2;
  __syncthreads();
  if (a.nchunk > 1) {
    if (pl == 0 && c < C) {
      a.tmp[((size_t)j * 2 + 0) * C + c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
      a.tmp[((size_t)j * 2 + 1) * C + c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int ticket = __hip_atomic_fetch_add(&a.counter[g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      is_last = ticket == a.nchunk - 1;
      if (is_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&a.counter[g], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      }
    }
    __syncthreads();
    if (!is_last) return;
    // last arriver: fold the chunk sums, again 4 part lanes x 4 independent rows in flight (fixed order)
    s1 = s2 = 0.0;
    if (c < C) {
      int k = pl;
      for (; k + 28 < a.nchunk; k += 32) {
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          av[u] = a.tmp[((size_t)(k + 4 * u) * 2 + 0) * C + c];
          bv[u] = a.tmp[((size_t)(k + 4 * u) * 2 + 1) * C + c];
        }
        s1 += ((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7]));
        s2 += ((bv[0] + bv[1]) + (bv[2] + bv[3])) + ((bv[4] + bv[5]) + (bv[6] + bv[7]));
      }
      for (; k < a.nchunk; k += 4) {
        s1 += a.tmp[((size_t)k * 2 + 0) * C + c];
        s2 += a.tmp[((size_t)k * 2 + 1) * C + c];
      }
    }
    red[0][pl][cl] = s1;
    red[1][pl][cl] = s2;
    __syncthreads();
  }
  if (pl != 0 || c >= C) return;
  const double t1 = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
  const double t2 = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  if (MODE == 2) {           // local moments only (SyncBatchNorm: merged across ranks afterwards)
    const double m = t1 / a.count;
    double m2 = t2 - t1 * m;
    a.mean[c] = (float)m;
    a.invstd[c] = (float)(m2 > 0.0 ? m2 : 0.0);      // M2, not an inverse std
    if (c == 0 && a.scale != nullptr) a.scale[0] = (float)a.count;
  } else if (MODE == 0) {
    if (c == 0 && a.num_batches_tracked != nullptr) *a.num_batches_tracked += 1;
    const double m = t1 / a.count;
    double var = (t2 - t1 * m) / a.count;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)a.eps));
    a.mean[c] = (float)m;
    a.invstd[c] = is;
    const float gm = a.gamma ? a.gamma[c] : 1.f, bt = a.beta ? a.beta[c] : 0.f;
    a.scale[c] = gm * is;
    a.shift[c] = bt - (float)m * gm * is;
    if (a.running_mean != nullptr) {
      const double unbiased = a.count > 1.0 ? var * a.count / (a.count - 1.0) : var;
      a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)m;
      a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
    }
  } else {
    const float db = (float)t1, dg = (float)t2;
    a.dbeta[c] = db;
    a.dgamma[c] = dg;
    if (a.coef != nullptr) {
      const float inv_count = (float)(1.0 / a.count);
      const float is = a.invstd[c], gm = a.gamma ? a.gamma[c] : 1.f;
      const float A = gm * is;
      const float B = -A * is * dg * inv_count;
      a.coef[c] = A;
      a.coef[C + c] = B;
      a.coef[2 * C + c] = -A * db * inv_count - B * a.mean[c];
      a.coef[3 * C + c] = a.scale ? a.scale[c] : 0.f;
      a.coef[4 * C + c] = a.shift ? a.shift[c] : 0.f;
    }
  }
}


// Per-channel coefficient vectors are loop-invariant for a thread whenever the grid stride is a
// multiple of the chunk-column count (always true for power-of-two channel counts): they are then
// loaded ONCE into registers instead of per element.
template <int VEC>
__device__ __forceinline__ void load_coef(const float* __restrict__ p, int c0, float (&v)[VEC]) {
#pragma unroll
  for (int q = 0; q < VEC / 4; ++q) {
    const float4 a = reinterpret_cast<const float4*>(p + c0)[q];
    v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
  }
}

// Workgroups of the grid-stride streaming kernels.  NOT a power of two: a lane's successive chunks (and the chunks the
// chip reads at the same moment in different streams) are grid * 4 KiB apart, and with 2048 / 4096 workgroups that
// stride maps them onto the same HBM channels -- 1280 (5 per CU, one resident round) is 8-11 % faster than 2048 on
// the batch-1024 tensors, 1024 / 2048 / 4096 are the slow ones (profiles/r2/bn_stream_grid.txt).
int g_bn_grid = 256 * 5;     // tuning hooks (bench/bn_layers.py, DPA_BN_GRID): workgroups of the streaming kernels, unroll
int g_bn_unroll = 1;
long g_nt_min_bytes = 32l << 20;    // tensors at least this large are streamed with nontemporal loads / stores
                                    // (they cannot stay in the 256 MB Infinity Cache anyway): +1.5 % on the bench
int g_nt_mask = 15;                 // which kernels stream: 1 BN apply, 2 BN backward apply, 4 BN backward reduce, 8 igemm stores
inline bool nt_auto(long bytes, int who) {
  return (g_nt_mask & who) && g_nt_min_bytes >= 0 && bytes >= g_nt_min_bytes;
}
int g_bn_reverse = 0;        // 1: the apply kernels walk their tensors from the END (what the producer wrote last
                             // is still in the Infinity Cache / L2)

template <typename T, int U, bool SMOOTH = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                       T* __restrict__ y, long nchunks, int cc,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int relu, int rev,
                                                       uint8_t* __restrict__ mask,
                                                       const float* __restrict__ res_scale,
                                                       const float* __restrict__ res_shift) {
  // mask (optional): one byte per 16-byte chunk, bit j = (y_j > 0) -- all BatchNorm backward needs of the output
  // res_scale / res_shift (optional): the residual is the RAW output of another convolution whose (activation-free)
  // BatchNorm is applied here on the fly: y = act(scale*x + shift + res_scale*res + res_shift)
  constexpr int VEC = 16 / (int)sizeof(T);
  const bool nts = (rev & 2) != 0, ntl = (rev & 4) != 0;      // streaming stores / loads (tuning hook)
  rev &= 1;
  const long stride = (long)gridDim.x * blockDim.x;
  const bool fixed = (stride % cc) == 0;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float scv[VEC], shv[VEC], rsc[VEC], rsh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { rsc[j] = 1.f; rsh[j] = 0.f; }
  // rev: chunk q = nchunks-1-i, i.e. the tensor is walked from its end (the rows the producing kernel wrote
  // last are the ones still cached); q % cc is again constant per thread when the stride is a multiple of cc
  int c0 = (int)((rev ? nchunks - 1 - i : i) % cc) * VEC;
  if (i >= nchunks) return;
  load_coef<VEC>(scale, c0, scv);
  load_coef<VEC>(shift, c0, shv);
  if (res_scale != nullptr) {
    load_coef<VEC>(res_scale, c0, rsc);
    load_coef<VEC>(res_shift, c0, rsh);
  }
  if (U > 1 && fixed && !rev) {
    // U chunks per thread per trip, all loads issued before the first use: more bytes in flight per wave
    for (; i + (U - 1) * stride < nchunks; i += U * stride) {
      Vec16<T> vx[U], vr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ntl) vx[u].load_nt(x + (i + u * stride) * VEC);
        else vx[u].load(x + (i + u * stride) * VEC);
      }
      if (res != nullptr) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (ntl) vr[u].load_nt(res + (i + u * stride) * VEC);
          else vr[u].load(res + (i + u * stride) * VEC);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        Vec16<T> vo;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float v = vx[u].get(j) * scv[j] + shv[j];
          if (res != nullptr) v += vr[u].get(j) * rsc[j] + rsh[j];
          if constexpr (SMOOTH) {
            v = act_value((relu >> 4) & 7, v);
          } else {
            if (relu) v = fmaxf(v, 0.f);
            if (relu & 4) v = fminf(v, 6.f);
          }
          vo.set(j, v);
        }
        if (nts) vo.store_nt(y + (i + u * stride) * VEC);
        else vo.store(y + (i + u * stride) * VEC);
        if (mask != nullptr) {
          uint32_t m = 0;
#pragma unroll
          for (int j = 0; j < VEC; ++j) m |= (vo.get(j) > 0.f ? 1u : 0u) << j;
          mask[i + u * stride] = (uint8_t)m;
        }
      }
    }
  }
  for (; i < nchunks; i += stride) {
    const long q = rev ? nchunks - 1 - i : i;
    if (!fixed) {
      c0 = (int)(q % cc) * VEC;
      load_coef<VEC>(scale, c0, scv);
      load_coef<VEC>(shift, c0, shv);
      if (res_scale != nullptr) {
        load_coef<VEC>(res_scale, c0, rsc);
        load_coef<VEC>(res_shift, c0, rsh);
      }
    }
    Vec16<T> vx, vr, vo;
    if (ntl) vx.load_nt(x + q * VEC);
    else vx.load(x + q * VEC);
    if (res != nullptr) {
      if (ntl) vr.load_nt(res + q * VEC);
      else vr.load(res + q * VEC);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float v = vx.get(j) * scv[j] + shv[j];
      if (res != nullptr) v += vr.get(j) * rsc[j] + rsh[j];
      if constexpr (SMOOTH) {
        v = act_value((relu >> 4) & 7, v);
      } else {
        if (relu) v = fmaxf(v, 0.f);
        if (relu & 4) v = fminf(v, 6.f);
      }
      vo.set(j, v);
    }
    if (nts) vo.store_nt(y + q * VEC);
    else vo.store(y + q * VEC);
    if (mask != nullptr) {
      uint32_t m = 0;
#pragma unroll
      for (int j = 0; j < VEC; ++j) m |= (vo.get(j) > 0.f ? 1u : 0u) << j;
      mask[q] = (uint8_t)m;
    }
  }
}

// coef[0] = A, coef[1] = B, coef[2] = K with dx = A*dy_m + B*x + K  (dy_m = ReLU-masked dy):
//   A = gamma*invstd, B = -A*invstd*sum(dy*xhat)/count, K = -A*sum(dy)/count - B*mean
__global__ void bn_bwd_coeffs_kernel(const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                     const float* __restrict__ gamma, float inv_count,
                                     const float* __restrict__ count_dev, int C, float* __restrict__ coef,
                                     const float* __restrict__ scale, const float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  coef[3 * C + c] = scale ? scale[c] : 0.f;
  coef[4 * C + c] = shift ? shift[c] : 0.f;
  if (count_dev != nullptr) inv_count = 1.f / count_dev[0];
  const float is = invstd[c], g = gamma ? gamma[c] : 1.f;
  const float A = g * is;
  const float B = -A * is * dgamma[c] * inv_count;
  coef[c] = A;
  coef[C + c] = B;
  coef[2 * C + c] = -A * dbeta[c] * inv_count - B * mean[c];
}

// One launch: fold [nparts][2][C] partials (4 part lanes x 64 channels per workgroup, fixed order),
// write dgamma / dbeta and -- when the element count is known here (no SyncBN all-reduce in
// between) -- the apply-pass coefficients.  Replaces fold + sum + coeffs (3 tiny launches).
__global__ __launch_bounds__(256) void bn_bwd_finalize_fused_kernel(
    const float* __restrict__ partial, int nparts, int C, float* __restrict__ dgamma, float* __restrict__ dbeta,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    float inv_count, float* __restrict__ coef, const float* __restrict__ scale, const float* __restrict__ shift) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (int k = pl; k < nparts; k += 4) {
      s1 += partial[((size_t)k * 2 + 0) * C + c];
      s2 += partial[((size_t)k * 2 + 1) * C + c];
    }
  red[0][pl][cl] = s1;
  red[1][pl][cl] = s2;
  __syncthreads();
  if (pl != 0 || c >= C) return;
  const float db = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
  const float dg = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  dbeta[c] = db;
  dgamma[c] = dg;
  if (coef != nullptr) {
    const float is = invstd[c], g = gamma ? gamma[c] : 1.f;
    const float A = g * is;
    const float B = -A * is * dg * inv_count;
    coef[c] = A;
    coef[C + c] = B;
    coef[2 * C + c] = -A * db * inv_count - B * mean[c];
    coef[3 * C + c] = scale ? scale[c] : 0.f;
    coef[4 * C + c] = shift ? shift[c] : 0.f;
  }
}

template <typename T, bool SMOOTH = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ y, T* __restrict__ dx,
                                                           T* __restrict__ dres, long nchunks, int cc, int C,
                                                           const float* __restrict__ coef, int relu, int rev) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const long stride = (long)gridDim.x * blockDim.x;
  const bool fixed = (stride % cc) == 0;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float ca[VEC], cb[VEC], ck[VEC], sc[VEC], sh[VEC];
  const bool nts = (rev & 2) != 0, ntl = (rev & 4) != 0;      // streaming stores / loads (tuning hook)
  rev &= 1;
  const int rm = relu & 3;             // see col_reduce_kernel
  const bool clip = (relu & 4) != 0;
  if (i >= nchunks) return;
  int c0 = (int)((rev ? nchunks - 1 - i : i) % cc) * VEC;
  load_coef<VEC>(coef, c0, ca);
  load_coef<VEC>(coef + C, c0, cb);
  load_coef<VEC>(coef + 2 * C, c0, ck);
  load_coef<VEC>(coef + 3 * C, c0, sc);
  load_coef<VEC>(coef + 4 * C, c0, sh);
  for (; i < nchunks; i += stride) {
    const long q = rev ? nchunks - 1 - i : i;
    if (!fixed) {
      c0 = (int)(q % cc) * VEC;
      load_coef<VEC>(coef, c0, ca);
      load_coef<VEC>(coef + C, c0, cb);
      load_coef<VEC>(coef + 2 * C, c0, ck);
      load_coef<VEC>(coef + 3 * C, c0, sc);
      load_coef<VEC>(coef + 4 * C, c0, sh);
    }
    Vec16<T> vd, vx, vy, vo, vr;
    if (ntl) { vd.load_nt(dy + q * VEC); vx.load_nt(x + q * VEC); }
    else { vd.load(dy + q * VEC); vx.load(x + q * VEC); }
    if (rm == 1) {
      if (ntl) vy.load_nt(y + q * VEC);
      else vy.load(y + q * VEC);
    }
    const uint32_t mbits = rm == 3 ? reinterpret_cast<const uint8_t*>(y)[q] : 0u;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float d = vd.get(j);
      const float xv = vx.get(j);
      if (rm == 3) {
        if (!((mbits >> j) & 1u)) d = 0.f;
      } else if (rm != 0) {
        const float yv = rm == 1 ? vy.get(j) : to_float(from_float<T>(sc[j] * xv + sh[j]));
        if (!(yv > 0.f) || (clip && !(yv < 6.f))) d = 0.f;
      }
      if constexpr (SMOOTH) d *= act_slope((relu >> 4) & 7, sc[j] * xv + sh[j]);
      vo.set(j, fmaf(ca[j], d, fmaf(cb[j], xv, ck[j])));      // explicit: every instantiation contracts alike
      vr.set(j, d);
    }
    if (nts) vo.store_nt(dx + q * VEC);
    else vo.store(dx + q * VEC);
    if (dres != nullptr) {
      if (nts) vr.store_nt(dres + q * VEC);
      else vr.store(dres + q * VEC);
    }
  }
}

// ---- lean streaming variants of the two apply passes -----------------------------------------------------------------
// The generic kernels above decide the mask source, streaming hints and the residual at run time; every such branch
// doubles a load path and the bf16 instantiations ended at 74-100 VGPRs with ONE 16-byte chunk per stream in flight per
// lane: 4.4-4.8 TB/s on the batch-1024 tensors (scratch/bn_stream_bench.py).  Here everything that shapes the loop is a
// template parameter, U chunks per stream are requested before the first use, and the register budget is capped so
// that DPA_BN_OCC waves per SIMD stay resident.  Taken when a thread's channel chunk is loop invariant (grid stride a
// multiple of the chunk-column count) and the tensor is walked forwards.
#ifndef DPA_BN_U
#define DPA_BN_U 2
#endif
#ifndef DPA_BN_OCC
#define DPA_BN_OCC 6
#endif

// RES: 0 no residual, 1 y = act(bn(x) + res), 2 y = act(bn(x) + res_scale * res + res_shift)
template <typename T, int RES, bool MASK, bool NT, int U>
__global__ __launch_bounds__(256, DPA_BN_OCC) void bn_apply_fast_kernel(
    const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, long nchunks, int cc,
    const float* __restrict__ scale, const float* __restrict__ shift, int relu, uint8_t* __restrict__ mask,
    const float* __restrict__ res_scale, const float* __restrict__ res_shift) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nchunks) return;
  const int c0 = (int)(i % cc) * VEC;
  float scv[VEC], shv[VEC], rsc[RES == 2 ? VEC : 1], rsh[RES == 2 ? VEC : 1];
  load_coef<VEC>(scale, c0, scv);
  load_coef<VEC>(shift, c0, shv);
  if constexpr (RES == 2) {
    load_coef<VEC>(res_scale, c0, rsc);
    load_coef<VEC>(res_shift, c0, rsh);
  }
  const float lo = (relu & 3) ? 0.f : -INFINITY, hi = (relu & 4) ? 6.f : INFINITY;
  auto one = [&](const Vec16<T>& vx, const Vec16<T>& vr, long q) {
    Vec16<T> vo;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float v = vx.get(j) * scv[j] + shv[j];
      if constexpr (RES == 1) v += vr.get(j);
      if constexpr (RES == 2) v += vr.get(j) * rsc[j] + rsh[j];      // same association as the generic kernel
      vo.set(j, fminf(fmaxf(v, lo), hi));
    }
    if constexpr (NT) vo.store_nt(y + q * VEC);
    else vo.store(y + q * VEC);
    if constexpr (MASK) {
      uint32_t m = 0;
#pragma unroll
      for (int j = 0; j < VEC; ++j) m |= (vo.get(j) > 0.f ? 1u : 0u) << j;
      mask[q] = (uint8_t)m;
    }
  };
  for (; i + (U - 1) * stride < nchunks; i += U * stride) {
    Vec16<T> vx[U], vr[RES ? U : 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long q = i + u * stride;
      if constexpr (NT) {
        vx[u].load_nt(x + q * VEC);
        if constexpr (RES != 0) vr[u].load_nt(res + q * VEC);
      } else {
        vx[u].load(x + q * VEC);
        if constexpr (RES != 0) vr[u].load(res + q * VEC);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one(vx[u], vr[RES ? u : 0], i + u * stride);
  }
  for (; i < nchunks; i += stride) {
    Vec16<T> vx, vr;
    vx.load(x + i * VEC);
    if constexpr (RES != 0) vr.load(res + i * VEC);
    one(vx, vr, i);
  }
}

// relu(scale * x + shift) like bn_apply_fast_kernel<T, 0, false, NT, U>, plus the COLUMN SUMS of what it stores: every
// workgroup leaves one row of partial sums (slab [grid][2][C], second row zero) -- the colsum(a) that the "forward fold"
// needs (BatchNorm statistics of the next 1x1 convolution from a^T a and colsum(a), hip_ops._ConvBnResReluFn) without
// another pass over a.  256 % cc == 0: a thread keeps one channel chunk.
template <typename T, bool NT, int U>
__global__ __launch_bounds__(256, DPA_BN_OCC) void bn_apply_colsum_kernel(
    const T* __restrict__ x, T* __restrict__ y, long nchunks, int cc, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ part) {
  constexpr int VEC = 16 / (int)sizeof(T);
  __shared__ float red[256 * VEC];
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = (int)(threadIdx.x % cc) * VEC;
  float scv[VEC], shv[VEC], acc[VEC];
  load_coef<VEC>(scale, c0, scv);
  load_coef<VEC>(shift, c0, shv);
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  auto one = [&](const Vec16<T>& vx, long q) {
    Vec16<T> vo;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      vo.set(j, fmaxf(vx.get(j) * scv[j] + shv[j], 0.f));
      acc[j] += vo.get(j);                       // the value as stored
    }
    if constexpr (NT) vo.store_nt(y + q * VEC);
    else vo.store(y + q * VEC);
  };
  for (; i + (U - 1) * stride < nchunks; i += U * stride) {
    Vec16<T> vx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (NT) vx[u].load_nt(x + (i + u * stride) * VEC);
      else vx[u].load(x + (i + u * stride) * VEC);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one(vx[u], i + u * stride);
  }
  for (; i < nchunks; i += stride) {
    Vec16<T> vx;
    vx.load(x + i * VEC);
    one(vx, i);
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[threadIdx.x * VEC + j] = acc[j];
  __syncthreads();
  if ((int)threadIdx.x < cc) {
    const int C = cc * VEC;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float s = 0.f;
      for (int k = threadIdx.x; k < 256; k += cc) s += red[k * VEC + j];
      part[((size_t)blockIdx.x * 2 + 0) * C + c0 + j] = s;
      part[((size_t)blockIdx.x * 2 + 1) * C + c0 + j] = 0.f;
    }
  }
}

// dx = A * dy_m + B * x + K, optional second output dres = dy_m; RM = the mask source (relu & 3, see col_reduce_kernel)
template <typename T, int RM, bool DRES, bool NT, int U>
__global__ __launch_bounds__(256, DPA_BN_OCC) void bn_bwd_apply_fast_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ dx, T* __restrict__ dres,
    long nchunks, int cc, int C, const float* __restrict__ coef, int relu) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nchunks) return;
  const int c0 = (int)(i % cc) * VEC;
  float ca[VEC], cb[VEC], ck[VEC], sc[RM == 2 ? VEC : 1], sh[RM == 2 ? VEC : 1];
  load_coef<VEC>(coef, c0, ca);
  load_coef<VEC>(coef + C, c0, cb);
  load_coef<VEC>(coef + 2 * C, c0, ck);
  if constexpr (RM == 2) {
    load_coef<VEC>(coef + 3 * C, c0, sc);
    load_coef<VEC>(coef + 4 * C, c0, sh);
  }
  const bool clip = (relu & 4) != 0;
  auto one = [&](const Vec16<T>& vd, const Vec16<T>& vx, const Vec16<T>& vy, uint32_t mbits, long q) {
    Vec16<T> vo, vr;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float d = vd.get(j);
      const float xv = vx.get(j);
      if constexpr (RM == 3) {
        if (!((mbits >> j) & 1u)) d = 0.f;
      } else if constexpr (RM == 1) {
        const float yv = vy.get(j);
        if (!(yv > 0.f) || (clip && !(yv < 6.f))) d = 0.f;
      } else if constexpr (RM == 2) {
        const float yv = to_float(from_float<T>(sc[j] * xv + sh[j]));
        if (!(yv > 0.f) || (clip && !(yv < 6.f))) d = 0.f;
      }
      vo.set(j, fmaf(ca[j], d, fmaf(cb[j], xv, ck[j])));      // explicit: every instantiation contracts alike
      if constexpr (DRES) vr.set(j, d);
    }
    if constexpr (NT) vo.store_nt(dx + q * VEC);
    else vo.store(dx + q * VEC);
    if constexpr (DRES) {
      if constexpr (NT) vr.store_nt(dres + q * VEC);
      else vr.store(dres + q * VEC);
    }
  };
  for (; i + (U - 1) * stride < nchunks; i += U * stride) {
    Vec16<T> vd[U], vx[U], vy[RM == 1 ? U : 1];
    uint32_t mb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long q = i + u * stride;
      mb[u] = 0u;
      if constexpr (RM == 3) mb[u] = reinterpret_cast<const uint8_t*>(y)[q];
      if constexpr (NT) {
        vd[u].load_nt(dy + q * VEC);
        vx[u].load_nt(x + q * VEC);
        if constexpr (RM == 1) vy[u].load_nt(y + q * VEC);
      } else {
        vd[u].load(dy + q * VEC);
        vx[u].load(x + q * VEC);
        if constexpr (RM == 1) vy[u].load(y + q * VEC);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one(vd[u], vx[u], vy[RM == 1 ? u : 0], mb[u], i + u * stride);
  }
  for (; i < nchunks; i += stride) {
    Vec16<T> vd, vx, vy;
    vd.load(dy + i * VEC);
    vx.load(x + i * VEC);
    if constexpr (RM == 1) vy.load(y + i * VEC);
    uint32_t mb = 0u;
    if constexpr (RM == 3) mb = reinterpret_cast<const uint8_t*>(y)[i];
    one(vd, vx, vy, mb, i);
  }
}

template <typename T>
void bn_apply_fast(int grid, hipStream_t s, const void* x, const void* res, void* y, long nchunks, int cc,
                   const float* scale, const float* shift, int relu, uint8_t* mask, const float* res_scale,
                   const float* res_shift, bool nt) {
  auto go = [&](auto resc, auto maskc, auto ntc) {
    hipLaunchKernelGGL((bn_apply_fast_kernel<T, decltype(resc)::value, decltype(maskc)::value, decltype(ntc)::value,
                                             DPA_BN_U>), dim3(grid), dim3(256), 0, s, (const T*)x, (const T*)res, (T*)y,
                       nchunks, cc, scale, shift, relu, mask, res_scale, res_shift);
  };
  auto go2 = [&](auto resc, auto maskc) { nt ? go(resc, maskc, std::true_type{}) : go(resc, maskc, std::false_type{}); };
  auto go1 = [&](auto resc) { mask != nullptr ? go2(resc, std::true_type{}) : go2(resc, std::false_type{}); };
  if (res == nullptr) go1(std::integral_constant<int, 0>{});
  else if (res_scale == nullptr) go1(std::integral_constant<int, 1>{});
  else go1(std::integral_constant<int, 2>{});
}

template <typename T>
void bn_bwd_apply_fast(int grid, hipStream_t s, const void* dy, const void* x, const void* y, void* dx, void* dres,
                       long nchunks, int cc, int C, const float* coef, int relu, bool nt) {
  auto go = [&](auto rmc, auto dresc, auto ntc) {
    hipLaunchKernelGGL((bn_bwd_apply_fast_kernel<T, decltype(rmc)::value, decltype(dresc)::value, decltype(ntc)::value,
                                                 DPA_BN_U>), dim3(grid), dim3(256), 0, s, (const T*)dy, (const T*)x,
                       (const T*)y, (T*)dx, (T*)dres, nchunks, cc, C, coef, relu);
  };
  auto go2 = [&](auto rmc, auto dresc) { nt ? go(rmc, dresc, std::true_type{}) : go(rmc, dresc, std::false_type{}); };
  auto go1 = [&](auto rmc) { dres != nullptr ? go2(rmc, std::true_type{}) : go2(rmc, std::false_type{}); };
  switch (relu & 3) {
    case 0: go1(std::integral_constant<int, 0>{}); break;
    case 1: go1(std::integral_constant<int, 1>{}); break;
    case 2: go1(std::integral_constant<int, 2>{}); break;
    default: go1(std::integral_constant<int, 3>{}); break;
  }
}

// Workgroups for a grid-stride pass over `nchunks` 16-byte chunks of a [rows][cc chunks] tensor: at most g_bn_grid, and
// -- when the tensor takes more than one trip -- a multiple of cc / gcd(cc, 256), so that the grid stride is a multiple
// of the chunk-column count and a thread keeps ONE channel chunk (coefficients in registers, the lean kernels) for any
// channel count, not only powers of two (DenseNet's 64 + 32k, EfficientNet's 144 / 240 / 672 ...).
inline int grid_for(long nchunks, int cc = 0) {
  long b = (nchunks + 255) / 256;
  if (b <= g_bn_grid) return (int)(b < 1 ? 1 : b);
  b = g_bn_grid;
  if (cc > 0) {
    int a = cc, g = 256;
    while (g) { const int t = a % g; a = g; g = t; }     // a = gcd(cc, 256)
    const long m = cc / a;
    if (m <= b) b = b / m * m;
    else if (m <= 2 * b) b = m;
  }
  return (int)b;
}

void check_c(int C, int dtype) {
  const int vec = dtype == kF32 ? 4 : 8;
  if (C % vec != 0) throw std::invalid_argument("bn: C must be a multiple of 16 bytes");
}

template <typename T, int MODE, bool SMOOTH>
void col_reduce_t(dim3 grid, hipStream_t s, const void* a, const void* x, const void* y, long M, int C, const float* mean,
                  const float* invstd, int relu, const float* scale, const float* shift, float* partial, long rpb) {
  auto go = [&](auto rmc, auto ntc) {
    hipLaunchKernelGGL((col_reduce_kernel<T, MODE, SMOOTH, decltype(rmc)::value, decltype(ntc)::value>), grid,
                       dim3(kBnThreads), 0, s, (const T*)a, (const T*)x, (const T*)y, M, C, mean, invstd, relu, scale,
                       shift, partial, rpb);
  };
  using std::integral_constant;
  if constexpr (MODE == 0) {
    go(integral_constant<int, 0>{}, std::false_type{});
  } else {
    const bool nt = (relu & 256) != 0;
    switch (relu & 3) {
      case 0: nt ? go(integral_constant<int, 0>{}, std::true_type{}) : go(integral_constant<int, 0>{}, std::false_type{}); break;
      case 1: nt ? go(integral_constant<int, 1>{}, std::true_type{}) : go(integral_constant<int, 1>{}, std::false_type{}); break;
      case 2: nt ? go(integral_constant<int, 2>{}, std::true_type{}) : go(integral_constant<int, 2>{}, std::false_type{}); break;
      default: nt ? go(integral_constant<int, 3>{}, std::true_type{}) : go(integral_constant<int, 3>{}, std::false_type{}); break;
    }
  }
}

template <int MODE>
void col_reduce(const void* a, const void* x, const void* y, long M, int C, int dtype, const float* mean,
                const float* invstd, int relu, const float* scale, const float* shift, float* partial, int nblk,
                hipStream_t s) {
  check_c(C, dtype);
  const int vec = dtype == kF32 ? 4 : 8;
  const int gy = ceil_div(C / vec, kBnThreads);
  const long rpb = (M + nblk - 1) / nblk;
  dim3 grid(nblk, gy);
  if (MODE == 1 && ((relu >> 4) & 7) != 0) {
    if (scale == nullptr || shift == nullptr) throw std::invalid_argument("bn: a fused activation needs scale / shift");
    DPA_DISPATCH_DTYPE(dtype, (col_reduce_t<T, MODE, MODE == 1>(grid, s, a, x, y, M, C, mean, invstd, relu, scale, shift,
                                                                partial, rpb)));
    DPA_CHECK_HIP(hipGetLastError());
    return;
  }
  DPA_DISPATCH_DTYPE(dtype, (col_reduce_t<T, MODE, false>(grid, s, a, x, y, M, C, mean, invstd, relu, scale, shift,
                                                          partial, rpb)));
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace

int bn_partial_rows(long M, int C) {
  // enough row blocks to fill the chip (~4 workgroups per CU over all column groups), >= 64 rows each
  const int gy = ceil_div(C / 4, kBnThreads * 2) > 0 ? ceil_div(C / 8, kBnThreads) : 1;
  long nblk = 1024 / (gy > 0 ? gy : 1);
  const long max_by_rows = (M + 31) / 32;       // small tensors: more, shorter row blocks (latency, not bytes)
  if (nblk > max_by_rows) nblk = max_by_rows;
  if (nblk < 1) nblk = 1;
  return (int)nblk;
}

void bn_fold_partials_launch(const float* partial, int nparts, int C, float* out, int nout, hipStream_t s) {
  if (nparts <= 0 || nout <= 0) return;
  const int chunk = (nparts + nout - 1) / nout;
  hipLaunchKernelGGL(fold_partials_kernel, dim3(ceil_div(C, 64), nout), dim3(256), 0, s, partial, nparts, C, out,
                     chunk);
  DPA_CHECK_HIP(hipGetLastError());
}

void bn_stats_launch(const void* x, long M, int C, int dtype, float* partial, int nblk, hipStream_t s) {
  col_reduce<0>(x, nullptr, nullptr, M, C, dtype, nullptr, nullptr, 0, nullptr, nullptr, partial, nblk, s);
}

void bn_bwd_reduce_launch(const void* dy, const void* x, const void* y, long M, int C, int dtype,
                          const float* mean, const float* invstd, int relu, const float* scale,
                          const float* shift, float* partial, int nblk, hipStream_t s) {
  if ((relu & 3) == 2 && (scale == nullptr || shift == nullptr)) throw std::invalid_argument("bn: relu mode 2 needs scale/shift");
  const bool nt = (g_bn_reverse & 64) || nt_auto(M * C * (dtype == kF32 ? 4l : 2l), 4);
  col_reduce<1>(dy, x, y, M, C, dtype, mean, invstd, relu | (nt ? 256 : 0), scale, shift, partial, nblk, s);
}

void bn_sync_merge_launch(const float* gathered, int W, int C, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, long* num_batches_tracked,
                          float* mean, float* invstd, float* scale, float* shift, float* count_out, hipStream_t s) {
  if (W <= 0 || C <= 0) return;
  hipLaunchKernelGGL(bn_sync_merge_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, s, gathered, W, C, gamma, beta, eps,
                     momentum, running_mean, running_var, num_batches_tracked, mean, invstd, scale, shift, count_out);
  DPA_CHECK_HIP(hipGetLastError());
}

void bn_eval_coeffs_launch(const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float eps, int C, float* scale, float* shift,
                           hipStream_t s) {
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, s, gamma, beta,
                     running_mean, running_var, eps, C, scale, shift);
  DPA_CHECK_HIP(hipGetLastError());
}

// Workgroups (= rows of the partial slab [n][2][C]) of bn_apply_colsum_launch; 0: the launch is not covered
int bn_apply_colsum_parts(long M, int C, int dtype) {
  if (dtype == kF32 || C % 8 != 0 || C / 8 > 256 || 256 % (C / 8) != 0 || M <= 0) return 0;
  return grid_for(M * (C / 8), C / 8);
}

void bn_apply_colsum_launch(const void* x, void* y, long M, int C, int dtype, const float* scale, const float* shift,
                            float* part, int nparts, hipStream_t s) {
  if (nparts <= 0 || nparts != bn_apply_colsum_parts(M, C, dtype))
    throw std::invalid_argument("bn_apply_colsum: slab rows must come from bn_apply_colsum_parts");
  const long nchunks = M * (C / 8);
  const bool nt = nt_auto(nchunks * 16, 1);
  if (dtype == kBF16) {
    if (nt) hipLaunchKernelGGL((bn_apply_colsum_kernel<bf16_t, true, DPA_BN_U>), dim3(nparts), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, nchunks, C / 8, scale, shift, part);
    else hipLaunchKernelGGL((bn_apply_colsum_kernel<bf16_t, false, DPA_BN_U>), dim3(nparts), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, nchunks, C / 8, scale, shift, part);
  } else {
    if (nt) hipLaunchKernelGGL((bn_apply_colsum_kernel<f16_t, true, DPA_BN_U>), dim3(nparts), dim3(256), 0, s, (const f16_t*)x, (f16_t*)y, nchunks, C / 8, scale, shift, part);
    else hipLaunchKernelGGL((bn_apply_colsum_kernel<f16_t, false, DPA_BN_U>), dim3(nparts), dim3(256), 0, s, (const f16_t*)x, (f16_t*)y, nchunks, C / 8, scale, shift, part);
  }
  DPA_CHECK_HIP(hipGetLastError());
}

void bn_bwd_finalize_launch(const float* partial, int nparts, int C, float* dgamma, float* dbeta,
                            hipStream_t s) {
  // partial[.][0] = sum dy (dbeta), partial[.][1] = sum dy*xhat (dgamma)
  hipLaunchKernelGGL(sum_partials_kernel, dim3(ceil_div(C, 16)), dim3(256), 0, s, partial, nparts, C, dbeta,
                     dgamma);
  DPA_CHECK_HIP(hipGetLastError());
}

void bn_apply_launch(const void* x, const void* residual, void* y, long M, int C, int dtype,
                     const float* scale, const float* shift, int relu, hipStream_t s, uint8_t* mask,
                     const float* res_scale, const float* res_shift) {
  if ((res_scale != nullptr) != (res_shift != nullptr) || (res_scale != nullptr && residual == nullptr))
    throw std::invalid_argument("bn_apply: res_scale / res_shift come in pairs and need a residual");
  check_c(C, dtype);
  const int vec = dtype == kF32 ? 4 : 8;
  const long nchunks = M * (C / vec);
  if (nchunks <= 0) return;
  const int grid = grid_for(nchunks, C / vec);
  const bool nt = nt_auto(nchunks * 16, 1);
  if (((relu >> 4) & 7) != 0) {        // BN + smooth activation (SiLU, hard-swish, ...) in one pass
    if (mask != nullptr) throw std::invalid_argument("bn_apply: no bit mask with a smooth activation");
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_apply_kernel<T, 1, true>), dim3(grid), dim3(256), 0, s, (const T*)x,
                                                 (const T*)residual, (T*)y, nchunks, C / vec, scale, shift, relu,
                                                 (g_bn_reverse & 1) | (nt ? 6 : 0), mask, res_scale, res_shift));
    DPA_CHECK_HIP(hipGetLastError());
    return;
  }
  if ((g_bn_reverse & (13 | 128)) == 0 && ((long)grid * 256) % (C / vec) == 0) {
    DPA_DISPATCH_DTYPE(dtype, (bn_apply_fast<T>(grid, s, x, residual, y, nchunks, C / vec, scale, shift, relu, mask,
                                                res_scale, res_shift, nt)));
    DPA_CHECK_HIP(hipGetLastError());
    return;
  }
  if (g_bn_unroll >= 4) {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_apply_kernel<T, 4>), dim3(grid), dim3(256), 0, s, (const T*)x,
                                                 (const T*)residual, (T*)y, nchunks, C / vec, scale, shift, relu, (g_bn_reverse & 1) | (nt ? 6 : ((g_bn_reverse >> 2) & 3) << 1), mask, res_scale, res_shift));
  } else if (g_bn_unroll >= 2) {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_apply_kernel<T, 2>), dim3(grid), dim3(256), 0, s, (const T*)x,
                                                 (const T*)residual, (T*)y, nchunks, C / vec, scale, shift, relu, (g_bn_reverse & 1) | (nt ? 6 : ((g_bn_reverse >> 2) & 3) << 1), mask, res_scale, res_shift));
  } else {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_apply_kernel<T, 1>), dim3(grid), dim3(256), 0, s, (const T*)x,
                                                 (const T*)residual, (T*)y, nchunks, C / vec, scale, shift, relu, (g_bn_reverse & 1) | (nt ? 6 : ((g_bn_reverse >> 2) & 3) << 1), mask, res_scale, res_shift));
  }
  DPA_CHECK_HIP(hipGetLastError());
}

void bn_bwd_finalize_fused_launch(const float* partial, int nparts, int C, float* dgamma, float* dbeta,
                                  const float* mean, const float* invstd, const float* gamma, double count,
                                  float* coef, const float* scale, const float* shift, hipStream_t s) {
  hipLaunchKernelGGL(bn_bwd_finalize_fused_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, s, partial, nparts, C, dgamma,
                     dbeta, mean, invstd, gamma, (float)(1.0 / count), coef, scale, shift);
  DPA_CHECK_HIP(hipGetLastError());
}

void bn_bwd_coeffs_launch(const float* dgamma, const float* dbeta, const float* mean, const float* invstd,
                          const float* gamma, double count, const float* count_dev, int C, float* coef,
                          const float* scale, const float* shift, hipStream_t s) {
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, s, dgamma, dbeta, mean, invstd,
                     gamma, (float)(1.0 / count), count_dev, C, coef, scale, shift);
  DPA_CHECK_HIP(hipGetLastError());
}

// coef: [5][C] from bn_bwd_finalize_fused_launch / bn_bwd_coeffs_launch
void bn_bwd_apply_launch(const void* dy, const void* x, const void* y, void* dx, void* dres, long M, int C,
                         int dtype, const float* coef, int relu, hipStream_t s) {
  check_c(C, dtype);
  const int vec = dtype == kF32 ? 4 : 8;
  const long nchunks = M * (C / vec);
  if (nchunks <= 0) return;
  const int grid = grid_for(nchunks, C / vec);
  if (((relu >> 4) & 7) != 0) {
    DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true>), dim3(grid), dim3(256), 0, s, (const T*)dy,
                                                 (const T*)x, (const T*)y, (T*)dx, (T*)dres, nchunks, C / vec, C, coef,
                                                 relu, ((g_bn_reverse >> 1) & 1) | (nt_auto(nchunks * 16, 2) ? 6 : 0)));
    DPA_CHECK_HIP(hipGetLastError());
    return;
  }
  if ((g_bn_reverse & (50 | 128)) == 0 && ((long)grid * 256) % (C / vec) == 0) {
    DPA_DISPATCH_DTYPE(dtype, (bn_bwd_apply_fast<T>(grid, s, dy, x, y, dx, dres, nchunks, C / vec, C, coef, relu,
                                                    nt_auto(nchunks * 16, 2))));
    DPA_CHECK_HIP(hipGetLastError());
    return;
  }
  DPA_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)dy,
                                               (const T*)x, (const T*)y, (T*)dx, (T*)dres, nchunks, C / vec, C,
                                               coef, relu, ((g_bn_reverse >> 1) & 1) | (nt_auto(nchunks * 16, 2) ? 6 : ((g_bn_reverse >> 4) & 3) << 1)));
  DPA_CHECK_HIP(hipGetLastError());
}

// Rows of the scratch slab bn_fold_finalize needs for `nparts` partial rows (1 = no scratch pass).
int bn_fold_chunks(int nparts) {
  // The fold is latency bound: a lane keeps 8 row pairs in flight per trip, in both phases.  DPA_BN_FOLD_ROWS rows per
  // workgroup balance the trips of the first phase (rows / 32) against the last arriver's (chunks / 32): a batch-1024
  // slab has 25k rows -> 256 rows per workgroup = 8 + 3 trips (64 rows, 4 pairs in flight: 4 + 25 trips, 20 us per layer).
  if (nparts <= 64) return 1;
  int n = (nparts + DPA_BN_FOLD_ROWS - 1) / DPA_BN_FOLD_ROWS;
  if (n < 2) n = 2;
  return n > 512 ? 512 : n;
}

static void fold_finalize_launch(int mode, BnFinalizeArgs a, hipStream_t s) {
  if (a.nparts <= 0 || a.C <= 0) return;
  a.nchunk = bn_fold_chunks(a.nparts);
  a.chunk = (a.nparts + a.nchunk - 1) / a.nchunk;
  if (a.nchunk > 1 && (a.tmp == nullptr || a.counter == nullptr))
    throw std::invalid_argument("bn_fold_finalize: scratch slab and ticket counters required");
  // one ticket counter per 64-channel group; callers allocate 256 of them (hip_ops._bn_scratch)
  if (ceil_div(a.C, 64) > 256) throw std::invalid_argument("bn_fold_finalize: more than 16384 channels");
  dim3 grid(ceil_div(a.C, 64), a.nchunk);
  if (mode == 0) hipLaunchKernelGGL(bn_fold_finalize_kernel<0>, grid, dim3(256), 0, s, a);
  else if (mode == 2) hipLaunchKernelGGL(bn_fold_finalize_kernel<2>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(bn_fold_finalize_kernel<1>, grid, dim3(256), 0, s, a);
  DPA_CHECK_HIP(hipGetLastError());
}

void bn_fold_finalize_launch(const float* partial, int nparts, int C, double* tmp, int* counter, double count,
                             const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                             float* running_var, long* num_batches_tracked, float* mean, float* invstd, float* scale,
                             float* shift, hipStream_t s) {
  BnFinalizeArgs a{};
  a.partial = partial; a.nparts = nparts; a.C = C; a.tmp = tmp; a.counter = counter; a.count = count;
  a.eps = eps; a.momentum = momentum; a.gamma = gamma; a.beta = beta; a.running_mean = running_mean;
  a.running_var = running_var; a.num_batches_tracked = num_batches_tracked;
  a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift;
  fold_finalize_launch(0, a, s);
}

void bn_fold_moments_launch(const float* partial, int nparts, int C, double* tmp, int* counter, double count,
                            float* mean, float* m2, float* count_out, hipStream_t s) {
  BnFinalizeArgs a{};
  a.partial = partial; a.nparts = nparts; a.C = C; a.tmp = tmp; a.counter = counter; a.count = count;
  a.mean = mean; a.invstd = m2; a.scale = count_out;
  fold_finalize_launch(2, a, s);
}

void bn_bwd_fold_finalize_launch(const float* partial, int nparts, int C, double* tmp, int* counter, double count,
                                 const float* gamma, const float* mean, const float* invstd, const float* scale,
                                 const float* shift, float* dgamma, float* dbeta, float* coef, hipStream_t s) {
  BnFinalizeArgs a{};
  a.partial = partial; a.nparts = nparts; a.C = C; a.tmp = tmp; a.counter = counter; a.count = count;
  a.gamma = gamma; a.mean = const_cast<float*>(mean); a.invstd = const_cast<float*>(invstd);
  a.scale = const_cast<float*>(scale); a.shift = const_cast<float*>(shift);
  a.dgamma = dgamma; a.dbeta = dbeta; a.coef = coef;
  fold_finalize_launch(1, a, s);
}

void bn_set_tuning(int grid_blocks, int unroll) {
  g_bn_grid = grid_blocks > 0 ? grid_blocks : 256 * 5;
  g_bn_unroll = unroll > 0 ? unroll : 1;
}
// bit 0: forward apply walks from the end, bit 1: backward apply does; bits 2-3: forward apply streaming stores /
// loads; bits 4-5: backward apply streaming stores / loads; bit 6: backward reduce streaming loads (nontemporal);
// bit 7: generic apply kernels only (A/B against the lean streaming variants)
void bn_set_reverse(int mask) { g_bn_reverse = mask; }
void set_nt_min_bytes(long bytes) { g_nt_min_bytes = bytes; }
long nt_min_bytes() { return (g_nt_mask & 8) ? g_nt_min_bytes : -1; }
void set_nt_mask(int mask) { g_nt_mask = mask; }

}  // namespace dpa
