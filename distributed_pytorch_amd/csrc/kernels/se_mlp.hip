// Squeeze-excitation MLP on [N, C] pooled activations:  s = gate(W2 act(W1 p + b1) + b2)  -- two launches forward, three
// backward (the block used to be ~20 tiny BLAS / elementwise launches per direction; SE networks are launch bound).  fp32 VALU: the products are [N, C] x [C, S] with S = 4 .. a few hundred,
// a few MFLOP per layer -- no matrix-core shape fits a squeeze width of 6 or 26, and the time is launch latency.
//   act : 0 ReLU, 1 SiLU                 gate: 0 sigmoid, 1 hard-sigmoid
// A workgroup owns NS samples so every weight element it reads is used NS times.
#include "../common.h"
#include "kernels.h"
#include <stdexcept>

namespace dpa {
namespace {

constexpr int kSeNS = 4;          // samples per workgroup
constexpr int kSeThreads = 256;

__device__ __forceinline__ float se_act(int act, float z) { return act == 0 ? fmaxf(z, 0.f) : z * fast_sigmoid(z); }
__device__ __forceinline__ float se_act_slope(int act, float z) {
  if (act == 0) return z > 0.f ? 1.f : 0.f;
  const float sg = fast_sigmoid(z);
  return sg * (1.f + z * (1.f - sg));
}
__device__ __forceinline__ float se_gate(int gate, float z) {
  return gate == 0 ? fast_sigmoid(z) : fminf(fmaxf(z + 3.f, 0.f), 6.f) * (1.f / 6.f);
}

// ---- "rows" kernel: out[n][j] = sum_k W[j][k] * in[n][k], one wave per output j (lanes across k, coalesced rows of W),
// NS samples per workgroup, blockIdx.y strides over j.  K is the long axis (C).
//   MODE 0 (forward) : in = p;                  z1 = acc + b1, h = act(z1)
//   MODE 1 (backward): in = ds * gate'(z2) (= dz2, written once by the y == 0 workgroups); dz1 = acc * act'(z1)
template <int MODE>
__global__ __launch_bounds__(kSeThreads) void se_rows_kernel(const float* __restrict__ in, const float* __restrict__ aux_s,
                                                             const float* __restrict__ aux_z2, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ z1_in,
                                                             float* __restrict__ out0, float* __restrict__ out1,
                                                             float* __restrict__ dz2_out, int N, int K, int J, int act,
                                                             int gate) {
  extern __shared__ float sm[];      // [NS][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * kSeNS;
  for (int i = tid; i < kSeNS * K; i += kSeThreads) {
    const int ns = i / K, k = i - ns * K;
    float v = 0.f;
    if (n0 + ns < N) {
      const long o = (long)(n0 + ns) * K + k;
      v = in[o];
      if (MODE == 1) {
        const float sv = aux_s[o], zv = aux_z2[o];
        v *= gate == 0 ? sv * (1.f - sv) : ((zv > -3.f && zv < 3.f) ? (1.f / 6.f) : 0.f);
        if (blockIdx.y == 0) dz2_out[o] = v;
      }
    }
    sm[i] = v;
  }
  __syncthreads();
  for (int j = blockIdx.y * (kSeThreads / 64) + wave; j < J; j += gridDim.y * (kSeThreads / 64)) {
    float acc[kSeNS];
#pragma unroll
    for (int ns = 0; ns < kSeNS; ++ns) acc[ns] = 0.f;
    const float* wr = w + (long)j * K;
    for (int k = lane; k < K; k += 64) {
      const float wv = wr[k];
#pragma unroll
      for (int ns = 0; ns < kSeNS; ++ns) acc[ns] += wv * sm[ns * K + k];
    }
#pragma unroll
    for (int ns = 0; ns < kSeNS; ++ns) acc[ns] = wave_sum(acc[ns]);
    if (lane == 0) {
#pragma unroll
      for (int ns = 0; ns < kSeNS; ++ns)
        if (n0 + ns < N) {
          const long o = (long)(n0 + ns) * J + j;
          if (MODE == 0) {
            const float z = acc[ns] + bias[j];
            out0[o] = z;
            out1[o] = se_act(act, z);
          } else {
            out0[o] = acc[ns] * se_act_slope(act, z1_in[o]);
          }
        }
    }
  }
}

// ---- "cols" kernel: out[n][c] = sum_j W(c, j) * in[n][j], one thread per output c, the short axis J (= S) in LDS.
//   MODE 0 (forward) : W(c, j) = w[c * J + j] (W2 rows);  z2 = acc + b2, s = gate(z2)
//   MODE 1 (backward): W(c, j) = w[j * C + c] (W1, coalesced across the threads);  dp = acc * inv_hw
template <int MODE>
__global__ __launch_bounds__(kSeThreads) void se_cols_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out0,
                                                             float* __restrict__ out1, int N, int C, int J, int gate,
                                                             float inv_hw) {
  extern __shared__ float sm[];      // [NS][J]
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * kSeNS;
  for (int i = tid; i < kSeNS * J; i += kSeThreads) {
    const int ns = i / J, j = i - ns * J;
    sm[i] = n0 + ns < N ? in[(long)(n0 + ns) * J + j] : 0.f;
  }
  __syncthreads();
  const int c = blockIdx.y * kSeThreads + tid;
  if (c >= C) return;
  float acc[kSeNS];
#pragma unroll
  for (int ns = 0; ns < kSeNS; ++ns) acc[ns] = 0.f;
  for (int j = 0; j < J; ++j) {
    const float wv = MODE == 0 ? w[(long)c * J + j] : w[(long)j * C + c];
#pragma unroll
    for (int ns = 0; ns < kSeNS; ++ns) acc[ns] += wv * sm[ns * J + j];
  }
#pragma unroll
  for (int ns = 0; ns < kSeNS; ++ns)
    if (n0 + ns < N) {
      const long o = (long)(n0 + ns) * C + c;
      if (MODE == 0) {
        const float z = acc[ns] + bias[c];
        out0[o] = z;
        out1[o] = se_gate(gate, z);
      } else {
        out0[o] = acc[ns] * inv_hw;
      }
    }
}

// Parameter gradients: an output element is owned by 8 lanes that each walk one eighth of the batch with eight loads in
// flight and meet in LDS in a fixed order (deterministic).  (One thread per element walking the whole batch four loads
// at a time was 64 dependent trips on 150 workgroups: 83 us per SE block on EfficientNet-B0.)
//   [0, S*C)            dW1[j][c] = sum_n dz1[n][j] p[n][c]
//   [S*C, 2*S*C)        dW2[c][j] = sum_n dz2[n][c] h[n][j]
//   then db1[S], db2[C]
__device__ __forceinline__ float se_dot_n(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb, int N) {
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  int n = 0;
  for (; n + 7 < N; n += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += a[(long)(n + u) * lda] * b[(long)(n + u) * ldb];
  }
  for (; n < N; ++n) acc[0] += a[(long)n * lda] * b[(long)n * ldb];
  return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}
__device__ __forceinline__ float se_sum_n(const float* __restrict__ a, long lda, int N) {
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  int n = 0;
  for (; n + 7 < N; n += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += a[(long)(n + u) * lda];
  }
  for (; n < N; ++n) acc[0] += a[(long)n * lda];
  return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

constexpr int kSeWgOut = 32;       // outputs per workgroup (x 8 batch slices = 256 threads)

__global__ __launch_bounds__(256) void se_mlp_wgrad_kernel(const float* __restrict__ dz1, const float* __restrict__ dz2,
                                                           const float* __restrict__ p, const float* __restrict__ h,
                                                           float* __restrict__ dw1, float* __restrict__ db1,
                                                           float* __restrict__ dw2, float* __restrict__ db2, int N, int C,
                                                           int S) {
  __shared__ float red[8][kSeWgOut];
  const int ol = threadIdx.x & (kSeWgOut - 1), sl = threadIdx.x / kSeWgOut;
  const long i = (long)blockIdx.x * kSeWgOut + ol;
  const long sc = (long)S * C;
  const int n0 = (int)((long)N * sl / 8), nn = (int)((long)N * (sl + 1) / 8) - n0;
  float v = 0.f;
  float* out = nullptr;
  if (i < sc) {
    const int j = (int)(i / C), c = (int)(i - (long)j * C);
    v = se_dot_n(dz1 + (long)n0 * S + j, S, p + (long)n0 * C + c, C, nn);
    out = dw1 + i;
  } else if (i < 2 * sc) {
    const long k = i - sc;
    const int c = (int)(k / S), j = (int)(k - (long)c * S);
    v = se_dot_n(dz2 + (long)n0 * C + c, C, h + (long)n0 * S + j, S, nn);
    out = dw2 + k;
  } else if (i < 2 * sc + S) {
    const int j = (int)(i - 2 * sc);
    v = se_sum_n(dz1 + (long)n0 * S + j, S, nn);
    out = db1 + j;
  } else if (i < 2 * sc + S + C) {
    const int c = (int)(i - 2 * sc - S);
    v = se_sum_n(dz2 + (long)n0 * C + c, C, nn);
    out = db2 + c;
  }
  red[sl][ol] = v;
  __syncthreads();
  if (sl == 0 && out != nullptr)
    *out = ((red[0][ol] + red[1][ol]) + (red[2][ol] + red[3][ol])) + ((red[4][ol] + red[5][ol]) + (red[6][ol] + red[7][ol]));
}

void se_check(int N, int C, int S) {
  if (N <= 0 || C <= 0 || S <= 0) throw std::invalid_argument("se_mlp: empty problem");
  if ((size_t)kSeNS * C * sizeof(float) > 150 * 1024) throw std::invalid_argument("se_mlp: C too large for the LDS tile");
}

template <typename K> void se_lds_attr(K kern, size_t lds) {
  if (lds > 64 * 1024)
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      150 * 1024));
}

int se_row_blocks(int J) {
  const int b = ceil_div(J, kSeThreads / 64);
  return b < 16 ? b : 16;
}

}  // namespace

void se_mlp_fwd_launch(const float* p, const float* w1, const float* b1, const float* w2, const float* b2, float* z1,
                       float* h, float* z2, float* s, int N, int C, int S, int act, int gate, hipStream_t st) {
  se_check(N, C, S);
  const size_t lds_c = (size_t)kSeNS * C * sizeof(float), lds_s = (size_t)kSeNS * S * sizeof(float);
  se_lds_attr(se_rows_kernel<0>, lds_c);
  hipLaunchKernelGGL(se_rows_kernel<0>, dim3(ceil_div(N, kSeNS), se_row_blocks(S)), dim3(kSeThreads), lds_c, st, p, nullptr,
                     nullptr, w1, b1, nullptr, z1, h, nullptr, N, C, S, act, gate);
  hipLaunchKernelGGL(se_cols_kernel<0>, dim3(ceil_div(N, kSeNS), ceil_div(C, kSeThreads)), dim3(kSeThreads), lds_s, st, h,
                     w2, b2, z2, s, N, C, S, gate, 1.f);
  DPA_CHECK_HIP(hipGetLastError());
}

void se_mlp_bwd_launch(const float* ds, const float* s, const float* z2, const float* z1, const float* p, const float* h,
                       const float* w1, const float* w2t, float* dz2, float* dz1, float* dp, float* dw1, float* db1,
                       float* dw2, float* db2, int N, int C, int S, int act, int gate, float inv_hw, hipStream_t st) {
  se_check(N, C, S);
  const size_t lds_c = (size_t)kSeNS * C * sizeof(float), lds_s = (size_t)kSeNS * S * sizeof(float);
  se_lds_attr(se_rows_kernel<1>, lds_c);
  hipLaunchKernelGGL(se_rows_kernel<1>, dim3(ceil_div(N, kSeNS), se_row_blocks(S)), dim3(kSeThreads), lds_c, st, ds, s, z2,
                     w2t, nullptr, z1, dz1, nullptr, dz2, N, C, S, act, gate);
  hipLaunchKernelGGL(se_cols_kernel<1>, dim3(ceil_div(N, kSeNS), ceil_div(C, kSeThreads)), dim3(kSeThreads), lds_s, st, dz1,
                     w1, nullptr, dp, nullptr, N, C, S, gate, inv_hw);
  if (dw1 != nullptr) {
    const long total = 2l * S * C + S + C;
    hipLaunchKernelGGL(se_mlp_wgrad_kernel, dim3((unsigned)((total + kSeWgOut - 1) / kSeWgOut)), dim3(256), 0, st, dz1, dz2, p, h, dw1,
                       db1, dw2, db2, N, C, S);
  }
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
