// Filter-row weight gradient for R x 3 stride-1 "same" convolutions on gfx950 MFMA:
//
//     dW[n][(r*3 + s)*Cred + c] = sum_{img, y, x} dY[img][y][x][n] * X[img][y + r - ph][x + s - 1][c]
//
// wgrad.hip stages the im2col operand once per TAP: a 3x3 layer reads every input element nine times per output-channel
// tile.  Here a workgroup owns one filter ROW r and 64 input channels, i.e. the three taps s = 0, 1, 2 of that row: the
// three shifted views of the activation differ by one pixel, so ONE staged tile of the input row -- with one extra
// entry on either side -- serves all three (fragment reads start at entry k + s), and the dY tile is shared by the
// three taps too.  Per MFMA the staged bytes drop by ~45 % against the 256 x 128 tile of wgrad.hip.
//
// The reduction axis is walked in whole image rows so that the shift never leaks into a neighbouring row: a K tile is RPT
// segments of SEG pixel slots; a segment holds one image row (W <= SEG, slots x >= W are zero in dY) or, for W = 7, two
// rows with a zero slot behind each (slot k -> row k / (W + 1), x = k % (W + 1)).  The staged activation segment has
// SEG + 2 entries, entry e holding slot e - 1 (zeros outside the image: that IS the convolution's padding).  Slots of
// dY that are zero multiply whatever sits in the corresponding activation entries, so no mask is applied anywhere in
// the MFMA loop.  SEG is a multiple of the MFMA K step (16): 12.5 % of the slots are padding for W = 7, 14, 28, 56.
//
// Wide images (W = NCH * W', NCH = 2 / 4, 33 <= W' <= 64: 112 = 2 x 56, 224 = 4 x 56 -- VGG) are walked as NCH "virtual
// rows" per image row: a virtual row is a contiguous run of W' pixels, so the addressing is that of an image of width
// W' with H * NCH rows, except that (a) the vertical tap shift is NCH virtual rows and (b) the two extra entries of a
// staged activation segment are real pixels of the neighbouring chunk unless the chunk touches the image border.
//
// Pixel-major LDS tiles + gfx950 transposed reads, fp32 partial slabs [splits][N][R*3*Cred] and the deterministic slab
// fold are those of wgrad.hip.  Replaces cuDNN's backward-filter for the 3x3 layers (reference: loss.backward(),
// nd_imagenet.py:302) -- SURVEY.md N-K3.
#include "../common.h"
#include "kernels.h"
#include <stdexcept>

namespace dpa {
namespace {

typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr3;

__device__ __forceinline__ uint2 tr_read3(const unsigned char* p) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr3)(p));
  uint2 r;
  __builtin_memcpy(&r, &v, 8);
  return r;
}

struct Row3Params {
  const void* dY;      // [NB*H*W][ldy]
  const void* X;       // [NB*H*W][ldx]
  float* slab;         // [splits][N][R*3*Cred]
  int NB, H, W, N, Cred, R, ph, ldy, ldx;
  int splits, rows_per_split;     // image rows (NB*H of them) per split: a multiple of the rows of a K tile
  int RS, WP;          // image rows per segment, slot stride of a row inside the segment (SEG when RS == 1)
  int NCH;             // virtual rows per image row (1, 2, 4); H and W above are then H * NCH and W / NCH
};

template <typename T, int BI, int WI, int WJ, int SEG, int RPT>
__global__ __launch_bounds__(64 * WI * WJ) void wgrad_row3_kernel(const Row3Params p) {
  static_assert(sizeof(T) == 2, "16-bit operands");
  constexpr int NT = 64 * WI * WJ;
  constexpr int CJ = 64;                               // input channels per workgroup
  constexpr int BKP = SEG * RPT;                       // pixel slots per K tile
  constexpr int QR = RPT * (SEG + 2);                  // staged activation entries per K tile
  constexpr int LDP = BI * 2 + 64, LDQ = CJ * 2 + 64;  // row strides = 64 (mod 128): conflict-free transposed reads
  constexpr int CPP = BI * 2 / 16, CPQ = CJ * 2 / 16;
  constexpr int P_IT = BKP * CPP / NT, Q_IT = (QR * CPQ + NT - 1) / NT;
  constexpr int WTI = BI / WI, WTJ = 3 * CJ / WJ;
  constexpr int MI = WTI / 32, NJ = WTJ / 32;
  constexpr int STAGE = BKP * LDP + QR * LDQ;
  static_assert(P_IT >= 1 && BKP * CPP % NT == 0 && MI >= 1 && NJ >= 1 && WTJ % 32 == 0 && CJ % 32 == 0, "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave / WJ, wj = wave % WJ;

  const int ncb = p.Cred / CJ;
  const uint32_t tiles_i = (p.N + BI - 1) / BI, tiles_j = (uint32_t)(p.R * ncb);
  const uint32_t tiles = tiles_i * tiles_j;
  const uint32_t lid = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t split = lid / tiles, tile = lid - split * tiles;
  const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
  const int r = tj / ncb, c0 = (tj - r * ncb) * CJ;
  const int i0 = ti * BI;
  const int dy = (r - p.ph) * p.NCH;                   // vertical tap shift in (virtual) rows
  const int NR = p.NB * p.H;
  const int row_begin = split * p.rows_per_split;
  const int row_end = min(NR, row_begin + p.rows_per_split);
  const int rows_per_tile = RPT * p.RS;
  const int nk = (row_end - row_begin + rows_per_tile - 1) / rows_per_tile;

  // 32-bit byte offsets relative to the first pixel of this split (64-bit bases): a split's rows span < 4 GiB (host checked)
  const char* __restrict__ Yl = reinterpret_cast<const char*>(p.dY) + (size_t)row_begin * p.W * p.ldy * sizeof(T);
  // the activation base sits one image row + one pixel EARLIER when dy = -1 ...: offsets below stay non-negative by
  // basing at the split's first row shifted by dy (rows before the tensor are never dereferenced: masked)
  // (and one PIXEL earlier: entry 0 of a segment is the pixel left of its first slot)
  const char* __restrict__ Xl = reinterpret_cast<const char*>(p.X) + (((long)row_begin + dy) * p.W - 1) * (long)p.ldx * (long)sizeof(T);

  // ---- dY staging: fixed (slot, chunk) per thread and iteration
  uint32_t poff[P_IT];
  int prow[P_IT];                      // image row of the slot relative to the K tile's first row; < 0: a zero slot
#pragma unroll
  for (int it = 0; it < P_IT; ++it) {
    const int idx = tid + it * NT;
    const int pk = idx / CPP, cp = idx % CPP;
    const int q = pk / SEG, k = pk % SEG;
    const int j = p.RS > 1 ? k / p.WP : 0, x = p.RS > 1 ? k - j * p.WP : k;
    const bool ok = x < p.W && j < p.RS && i0 + cp * 8 < p.N;
    prow[it] = ok ? q * p.RS + j : -1;
    poff[it] = ok ? (uint32_t)((((q * p.RS + j) * p.W + x) * p.ldy + i0 + cp * 8) * (int)sizeof(T)) : 0u;
  }
  // ---- activation staging: entry e of segment q holds slot e - 1
  uint32_t qoff[Q_IT];
  int qrow[Q_IT], qy[Q_IT];            // relative image row (< 0: a zero entry), its y coordinate (advanced per K tile)
  int qhalo[Q_IT];                     // NCH > 1: -1 / +1 = the entry left / right of the chunk (valid unless at the border)
#pragma unroll
  for (int it = 0; it < Q_IT; ++it) {
    const int idx = tid + it * NT;
    const int qr = idx / CPQ, cq = idx % CPQ;
    const int q = qr / (SEG + 2), e = qr % (SEG + 2);
    const int k = e - 1;
    const int j = (p.RS > 1 && k >= 0) ? k / p.WP : 0, x = (p.RS > 1 && k >= 0) ? k - j * p.WP : k;
    const bool inner = qr < QR && k >= 0 && x < p.W && j < p.RS;
    qhalo[it] = (qr < QR && p.NCH > 1) ? (k < 0 ? -1 : (k == p.W ? 1 : 0)) : 0;
    const bool ok = inner || qhalo[it] != 0;
    const int rel = q * p.RS + j;
    qrow[it] = ok ? rel : -1;
    qy[it] = ok ? (row_begin + rel) % p.H : 0;
    qoff[it] = ok ? (uint32_t)(((rel * p.W + x + 1) * p.ldx + c0 + cq * 8) * (int)sizeof(T)) : 0u;
  }
  const uint32_t pstep = (uint32_t)(rows_per_tile * p.W * p.ldy * (int)sizeof(T));
  const uint32_t qstep = (uint32_t)(rows_per_tile * p.W * p.ldx * (int)sizeof(T));
  const int ystep = rows_per_tile % p.H;

  uint4 rp[P_IT], rq[Q_IT];
  uint32_t pmask[P_IT], qmask[Q_IT];
  auto load_tile = [&](int row0) {          // row0: first image row of the K tile
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
      const bool ok = prow[it] >= 0 && row0 + prow[it] < row_end;
      pmask[it] = ok ? 0xffffffffu : 0u;
      rp[it] = *reinterpret_cast<const uint4*>(Yl + (ok ? poff[it] : 0u));
      poff[it] += pstep;
    }
#pragma unroll
    for (int it = 0; it < Q_IT; ++it) {
      const int chunk = qy[it] & (p.NCH - 1);        // NCH is a power of two and divides the rows per image
      const bool ok = qrow[it] >= 0 && row0 + qrow[it] < row_end && (unsigned)(qy[it] + dy) < (unsigned)p.H &&
                      (qhalo[it] == 0 || (qhalo[it] < 0 ? chunk != 0 : chunk != p.NCH - 1));
      qmask[it] = ok ? 0xffffffffu : 0u;
      rq[it] = *reinterpret_cast<const uint4*>(ok ? Xl + qoff[it] : reinterpret_cast<const char*>(p.X));
      qoff[it] += qstep;
      int y = qy[it] + ystep;
      qy[it] = y >= p.H ? y - p.H : y;
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* sp = smem + buf * STAGE;
    unsigned char* sq = sp + BKP * LDP;
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
      const int idx = tid + it * NT;
      uint4 v = rp[it];
      v.x &= pmask[it]; v.y &= pmask[it]; v.z &= pmask[it]; v.w &= pmask[it];
      *reinterpret_cast<uint4*>(sp + (idx / CPP) * LDP + (idx % CPP) * 16) = v;
    }
#pragma unroll
    for (int it = 0; it < Q_IT; ++it) {
      const int idx = tid + it * NT;
      if (idx < QR * CPQ) {
        uint4 v = rq[it];
        v.x &= qmask[it]; v.y &= qmask[it]; v.z &= qmask[it]; v.w &= qmask[it];
        *reinterpret_cast<uint4*>(sq + (idx / CPQ) * LDQ + (idx % CPQ) * 16) = v;
      }
    }
  };

  f32x16_t acc[MI][NJ];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < NJ; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  const int g = lane >> 4, gi = lane & 15;
  const int trow = 8 * (g >> 1) + (gi >> 2);           // lane address inside a 4-row x 16-col transposed block (wgrad.hip)
  const int tcol = 16 * (g & 1) + 4 * (gi & 3);
  // this wave's three 32-column blocks of the (tap, channel) axis: tap s = column / 64, channel = column % 64
  int bs[NJ], bc[NJ];
#pragma unroll
  for (int b = 0; b < NJ; ++b) {
    const int jc = wj * WTJ + b * 32;
    bs[b] = jc / CJ;
    bc[b] = jc % CJ;
  }

  auto compute = [&](int cur) {
    const unsigned char* sp = smem + cur * STAGE;
    const unsigned char* sq = sp + BKP * LDP;
#pragma unroll
    for (int q = 0; q < RPT; ++q)
#pragma unroll
      for (int ks = 0; ks < SEG / 16; ++ks) {
        uint4 af[MI], bf[NJ];
#pragma unroll
        for (int a = 0; a < MI; ++a) {
          const unsigned char* base = sp + (q * SEG + ks * 16 + trow) * LDP + (wi * WTI + a * 32 + tcol) * 2;
          const uint2 lo = tr_read3(base), hi = tr_read3(base + 4 * LDP);
          af[a] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int b = 0; b < NJ; ++b) {
          const unsigned char* base = sq + (q * (SEG + 2) + ks * 16 + bs[b] + trow) * LDQ + (bc[b] + tcol) * 2;
          const uint2 lo = tr_read3(base), hi = tr_read3(base + 4 * LDQ);
          bf[b] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
          for (int b = 0; b < NJ; ++b) {
            if constexpr (TypeInfo<T>::kCode == kBF16) {
              bf16x8_t av, bv;
              __builtin_memcpy(&av, &af[a], 16);
              __builtin_memcpy(&bv, &bf[b], 16);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[a][b], 0, 0, 0);
            } else {
              f16x8_t av, bv;
              __builtin_memcpy(&av, &af[a], 16);
              __builtin_memcpy(&bv, &bf[b], 16);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[a][b], 0, 0, 0);
            }
          }
      }
  };

  // one K tile in flight in registers while one is consumed from LDS; tiles past the range are all zeros
  load_tile(row_begin);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    load_tile(row_begin + (kt + 1) * rows_per_tile);
    compute(0);
    store_tile(1);
    __syncthreads();
    load_tile(row_begin + (kt + 2) * rows_per_tile);
    compute(1);
    store_tile(0);
    __syncthreads();
  }

  // acc[a][b][e]: i = wi*WTI + a*32 + (e&3) + 8*(e>>2) + 4*lh ; column block b -> tap bs[b], channel bc[b] + l31
  const int Ktot = p.R * 3 * p.Cred;
  float* __restrict__ out = p.slab + (size_t)split * p.N * Ktot;
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int b = 0; b < NJ; ++b) {
      const int col = (r * 3 + bs[b]) * p.Cred + c0 + bc[b] + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = i0 + wi * WTI + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (i < p.N) out[(size_t)i * Ktot + col] = acc[a][b][e];
      }
    }
}

template <typename T, int BI, int WI, int WJ, int SEG, int RPT>
void launch_row3w(const Row3Params& p, hipStream_t stream) {
  constexpr int BKP = SEG * RPT, QR = RPT * (SEG + 2);
  constexpr int lds = 2 * (BKP * (BI * 2 + 64) + QR * (64 * 2 + 64));
  auto kern = wgrad_row3_kernel<T, BI, WI, WJ, SEG, RPT>;
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    DPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  const long blocks = (long)ceil_div(p.N, BI) * p.R * (p.Cred / 64) * p.splits;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WI * WJ), lds, stream, p);
}

int g_row3w_variant = 0;   // wave layout under test (bench/experiments/wgrad_row3_layers.py)

template <typename T, int SEG, int RPT>
void dispatch_row3w(const Row3Params& p, hipStream_t stream) {
  const int v = g_row3w_variant;
  if (p.N >= 256) {
    if (v == 1) launch_row3w<T, 256, 2, 2, SEG, RPT>(p, stream);      // 4 waves of 128 x 96
    else launch_row3w<T, 256, 4, 2, SEG, RPT>(p, stream);             // 8 waves of 64 x 96
  } else if (p.N > 64) {
    if (v == 1) launch_row3w<T, 128, 1, 2, SEG, RPT>(p, stream);      // 2 waves of 128 x 96
    else launch_row3w<T, 128, 2, 2, SEG, RPT>(p, stream);             // 4 waves of 64 x 96
  } else {
    if (v == 1) launch_row3w<T, 64, 1, 2, SEG, RPT>(p, stream);       // 2 waves of 64 x 96
    else if (v == 2) launch_row3w<T, 64, 1, 1, SEG, RPT>(p, stream);  // 1 wave of 64 x 192
    else launch_row3w<T, 64, 2, 2, SEG, RPT>(p, stream);              // 4 waves of 32 x 96
  }
}

int g_row3w_mode = -1;      // -1 automatic (output-channel counts where it measured faster than wgrad.hip), 0 never, 1 whenever covered

// segment geometry for an image width: false = not covered
bool row3w_geometry(int W, int* seg, int* rpt, int* rs, int* wp, int* nch) {
  *nch = 1;
  if (W > 64) {         // wide images as 2 / 4 chunks of 33..64 pixels per row
    int n = 0;
    if (W % 2 == 0 && W / 2 >= 33 && W / 2 <= 64) n = 2;
    else if (W % 4 == 0 && W / 4 >= 33 && W / 4 <= 64) n = 4;
    if (n == 0) return false;
    *nch = n; *seg = 64; *rpt = 1; *rs = 1; *wp = 64;
    return true;
  }
  if (W == 7) { *seg = 16; *rpt = 4; *rs = 2; *wp = 8; return true; }           // two rows + a zero slot each per segment
  if (W >= 8 && W <= 16) { *seg = 16; *rpt = 4; *rs = 1; *wp = 16; return true; }
  if (W >= 17 && W <= 32) { *seg = 32; *rpt = 2; *rs = 1; *wp = 32; return true; }
  if (W >= 33 && W <= 64) { *seg = 64; *rpt = 1; *rs = 1; *wp = 64; return true; }
  return false;
}

int row3w_tile_bi(int N) { return N >= 256 ? 256 : (N > 64 ? 128 : 64); }

}  // namespace

void wgrad_row3_set_mode(int mode) { g_row3w_mode = mode; }
void wgrad_row3_set_variant(int v) { g_row3w_variant = v; }

// Splits of the image rows for the filter-row weight gradient; 0 = the problem is not covered (use wgrad_launch).
int wgrad_row3_splits(int NB, int H, int W, int N, int Cred, int R, int dtype, int* rows_per_split) {
  int seg, rpt, rs, wp, nch;
  if (g_row3w_mode == 0 || dtype == kF32 || Cred % 64 != 0 || N % 8 != 0 || R < 1 || R > 7 || R % 2 == 0) return 0;
  if (!row3w_geometry(W, &seg, &rpt, &rs, &wp, &nch)) return 0;
  // measured (profiles/r3/wgrad_row3_layers.txt): 1.46x on N = 64, 0.75x on N >= 128 where the 256 x 128 tile of wgrad.hip
  // already runs at 620..800 TFLOP/s and the split count here is capped by the slab size
  if (g_row3w_mode < 0 && N > 64) return 0;
  const long NR = (long)NB * H * nch;                  // (virtual) rows
  const int rows_per_tile = rpt * rs;
  if (NR < 2l * rows_per_tile || NR >= (1l << 31)) return 0;
  const int bi = row3w_tile_bi(N);
  const long tiles = (long)ceil_div(N, bi) * R * (Cred / 64);
  const long target = 768l * 256 / (bi == 256 ? 512 : 256);
  long splits = (target + tiles - 1) / tiles;
  const long max_by_work = (NR + 8l * rows_per_tile - 1) / (8l * rows_per_tile);      // >= 8 K tiles per workgroup
  if (splits > max_by_work) splits = max_by_work;
  const long out_elems = (long)N * R * 3 * Cred;
  const long max_by_slab = (64l << 20) / (out_elems * 4) + 1;
  if (splits > max_by_slab) splits = max_by_slab;
  if (splits < 1) splits = 1;
  long rps = (NR + splits - 1) / splits;
  rps = (rps + rows_per_tile - 1) / rows_per_tile * rows_per_tile;
  splits = (NR + rps - 1) / rps;
  *rows_per_split = (int)rps;
  return (int)splits;
}

void wgrad_row3_launch(const void* dY, const void* X, float* slab, int NB, int H, int W, int N, int Cred, int R, int ph,
                       int ldy, int ldx, int splits, int rows_per_split, int dtype, hipStream_t stream) {
  Row3Params p{};
  int seg, rpt;
  if (!row3w_geometry(W, &seg, &rpt, &p.RS, &p.WP, &p.NCH) || dtype == kF32 || Cred % 64 != 0 || N % 8 != 0 || ldy % 8 != 0 ||
      ldx % 8 != 0 || R < 1 || R > 7 || ph < 0 || ph >= R)
    throw std::invalid_argument("wgrad_row3: problem not covered (ask wgrad_row3_splits first)");
  if (splits < 1 || rows_per_split % (rpt * p.RS) != 0 || (long)splits * rows_per_split < (long)NB * H * p.NCH)
    throw std::invalid_argument("wgrad_row3: bad split of the image rows");
  H *= p.NCH;          // virtual rows per image ...
  W /= p.NCH;          // ... of W' pixels
  // 32-bit offsets inside one split (+ two K tiles of lookahead, + the filter's vertical reach)
  const double span = ((double)rows_per_split + 2.0 * rpt * p.RS + (double)R * p.NCH) * W * 2.0;
  if (span * ldy >= 4294967296.0 || span * ldx >= 4294967296.0)
    throw std::invalid_argument("wgrad_row3: a split spans more than 4 GiB");
  p.dY = dY; p.X = X; p.slab = slab;
  p.NB = NB; p.H = H; p.W = W; p.N = N; p.Cred = Cred; p.R = R; p.ph = ph; p.ldy = ldy; p.ldx = ldx;
  p.splits = splits; p.rows_per_split = rows_per_split;
  if (dtype == kBF16) {
    if (seg == 16) dispatch_row3w<bf16_t, 16, 4>(p, stream);
    else if (seg == 32) dispatch_row3w<bf16_t, 32, 2>(p, stream);
    else dispatch_row3w<bf16_t, 64, 1>(p, stream);
  } else {
    if (seg == 16) dispatch_row3w<f16_t, 16, 4>(p, stream);
    else if (seg == 32) dispatch_row3w<f16_t, 32, 2>(p, stream);
    else dispatch_row3w<f16_t, 64, 1>(p, stream);
  }
  DPA_CHECK_HIP(hipGetLastError());
}

}  // namespace dpa
