// pybind11 surface of the native library.  Tensors cross as raw device addresses (integers from
// torch.Tensor.data_ptr()) and the stream as the hipStream_t of torch's current stream, so this
// file needs no torch headers and the extension builds in seconds with plain hipcc.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "kernels/kernels.h"
#include "comm/comm.h"

namespace py = pybind11;
using namespace dpa;

namespace {
template <typename T> T* ptr(uintptr_t p) { return reinterpret_cast<T*>(p); }
hipStream_t st(uintptr_t s) { return reinterpret_cast<hipStream_t>(s); }
}  // namespace

PYBIND11_MODULE(_C, m) {
  m.doc() = "distributed_pytorch_amd native kernels + comm runtime (gfx950 / MI355X)";
  m.attr("F32") = (int)kF32;
  m.attr("BF16") = (int)kBF16;
  m.attr("F16") = (int)kF16;
  m.attr("IGEMM_RELU") = (int)kIgemmRelu;
  m.attr("IGEMM_SCATTER") = (int)kIgemmScatter;

  m.def("igemm",
        [](uintptr_t A, uintptr_t B, uintptr_t C, uintptr_t R, uintptr_t bias, uintptr_t stats, uintptr_t taps, int M,
           int N,
           int Cred, int ntaps, int lda, int ldb, int ldc, int IH, int IW, int SY, int SX, int OH, int OW, int OHF,
           int OWF, int OSY, int OSX, int OY0, int OX0, int flags, int dtype, int out_dtype, uintptr_t stream,
           uintptr_t bn_x, uintptr_t bn_z, uintptr_t bn_coef, int bn_relu, int groups, long a_gstride, long b_gstride,
           long c_gstride, int stats_ld, uintptr_t r_mask, uintptr_t a_scale, uintptr_t a_shift, uintptr_t a2, int lda2,
           int cred2, uintptr_t b2, int ldb2) {
          IgemmParams p;
          p.A2 = ptr<const void>(a2); p.lda2 = lda2; p.Cred2 = cred2;
          p.B2 = ptr<const void>(b2); p.ldb2 = ldb2;
          p.r_mask = ptr<const uint8_t>(r_mask);
          p.a_scale = ptr<const float>(a_scale);
          p.a_shift = ptr<const float>(a_shift);
          p.groups = groups; p.a_gstride = a_gstride; p.b_gstride = b_gstride; p.c_gstride = c_gstride;
          p.stats_ld = stats_ld;
          p.bn_x = ptr<const void>(bn_x);
          p.bn_z = ptr<const void>(bn_z);
          if (flags & kIgemmReluMask) p.relu_mask = ptr<uint8_t>(bn_z);      // (the slot doubles as the mask OUTPUT)
          p.bn_coef = ptr<const float>(bn_coef);
          p.bn_relu = bn_relu;
          p.A = ptr<const void>(A);
          p.B = ptr<const void>(B);
          p.C = ptr<void>(C);
          p.R = ptr<const void>(R);
          p.bias = ptr<const float>(bias);
          p.stats = ptr<float>(stats);
          p.taps = ptr<const int>(taps);
          p.M = M; p.N = N; p.Cred = Cred; p.ntaps = ntaps;
          p.lda = lda; p.ldb = ldb; p.ldc = ldc;
          p.IH = IH; p.IW = IW; p.SY = SY; p.SX = SX;
          p.OHF = OHF; p.OWF = OWF; p.OSY = OSY; p.OSX = OSX; p.OY0 = OY0; p.OX0 = OX0;
          p.flags = flags;
          p.div_ohow = make_fastdiv((uint32_t)(OH * OW));
          p.div_ow = make_fastdiv((uint32_t)OW);
          igemm_launch(p, dtype, out_dtype, st(stream));
        },
        py::arg("A"), py::arg("B"), py::arg("C"), py::arg("R"), py::arg("bias"), py::arg("stats"), py::arg("taps"),
        py::arg("M"), py::arg("N"), py::arg("Cred"), py::arg("ntaps"), py::arg("lda"), py::arg("ldb"), py::arg("ldc"),
        py::arg("IH"), py::arg("IW"), py::arg("SY"), py::arg("SX"), py::arg("OH"), py::arg("OW"), py::arg("OHF"),
        py::arg("OWF"), py::arg("OSY"), py::arg("OSX"), py::arg("OY0"), py::arg("OX0"), py::arg("flags"),
        py::arg("dtype"), py::arg("out_dtype"), py::arg("stream"), py::arg("bn_x") = 0, py::arg("bn_z") = 0,
        py::arg("bn_coef") = 0, py::arg("bn_relu") = 0, py::arg("groups") = 1, py::arg("a_gstride") = 0,
        py::arg("b_gstride") = 0, py::arg("c_gstride") = 0, py::arg("stats_ld") = 0, py::arg("r_mask") = 0,
        py::arg("a_scale") = 0, py::arg("a_shift") = 0, py::arg("a2") = 0, py::arg("lda2") = 0, py::arg("cred2") = 0, py::arg("b2") = 0,
        py::arg("ldb2") = 0);
  m.attr("IGEMM_BN_BWD") = (int)kIgemmBnBwd;
  m.attr("IGEMM_AXF") = (int)kIgemmAxf;
  m.attr("IGEMM_DENSE1") = (int)kIgemmDense1;
  m.attr("IGEMM_RSTRIDED") = (int)kIgemmRStrided;
  m.attr("IGEMM_XMASK") = (int)kIgemmXMask;
  m.def("igemm_axf_ok", &igemm_axf_ok);
  m.attr("IGEMM_ROW3") = (int)kIgemmRow3;
  m.def("igemm_set_row3", &igemm_set_row3);
  m.def("igemm_set_tail_min", &igemm_set_tail_min);
  m.def("igemm_set_lean", &igemm_set_lean);
  m.attr("IGEMM_BN_FOLD") = (int)kIgemmBnFold;
  m.attr("IGEMM_RELU_MASK") = (int)kIgemmReluMask;
  m.def("bnfold_coeffs", [](uintptr_t raw, uintptr_t w, int ldw, uintptr_t dbeta_sum, uintptr_t mean, uintptr_t invstd,
                            uintptr_t gamma, double count, int N, int C, uintptr_t dgamma_out, float beta_g,
                            uintptr_t dbeta_out, float beta_b, uintptr_t coef, uintptr_t wt_scaled, int ldwt, uintptr_t ax, int dtype,
                            uintptr_t s) {
    bnfold_coeffs_launch(ptr<const float>(raw), ptr<const void>(w), ldw, ptr<const float>(dbeta_sum), ptr<const float>(mean),
                         ptr<const float>(invstd), ptr<const float>(gamma), count, N, C, ptr<float>(dgamma_out), beta_g,
                         ptr<float>(dbeta_out), beta_b, ptr<float>(coef), ptr<void>(wt_scaled), ldwt, ptr<void>(ax), dtype, st(s));
  });
  m.def("bnfold_fwd", [](uintptr_t t, uintptr_t w, uintptr_t cs, double count, uintptr_t gamma, uintptr_t beta, float eps,
                         float momentum, uintptr_t rm, uintptr_t rv, uintptr_t nbt, uintptr_t coef, uintptr_t wsc, int N, int C,
                         int dtype, uintptr_t s) {
    bnfold_fwd_launch(ptr<const float>(t), ptr<const float>(w), ptr<const float>(cs), count, ptr<const float>(gamma),
                      ptr<const float>(beta), eps, momentum, ptr<float>(rm), ptr<float>(rv), ptr<long>(nbt), ptr<float>(coef),
                      ptr<void>(wsc), N, C, dtype, st(s));
  });
  m.def("bnfold_wgrad", [](uintptr_t raw, uintptr_t wg, uintptr_t coef, uintptr_t cs, uintptr_t out, int N, int C, float beta,
                           uintptr_t s) {
    bnfold_wgrad_launch(ptr<const float>(raw), ptr<const float>(wg), ptr<const float>(coef), ptr<const float>(cs),
                        ptr<float>(out), N, C, beta, st(s));
  });
  m.def("conv8_set_mode", &conv8_set_mode);
  m.def("conv8_set_min_k", &conv8_set_min_k);
  m.def("conv8_set_tail_split", &conv8_set_tail_split);
  m.def("conv8_launches", &conv8_launches);
  m.def("conv8_last", &conv8_last);
  m.def("conv8_error", [](uintptr_t s) { return conv8_error(st(s)); });
  m.def("igemm_set_lean_ablate", &igemm_set_lean_ablate);
  m.def("igemm_set_nt_store", &igemm_set_nt_store);
  m.def("set_nt_min_bytes", &set_nt_min_bytes);
  m.def("set_nt_mask", &set_nt_mask);
  m.def("set_stream_grid", &set_stream_grid);
  m.def("igemm_tile_m", &igemm_tile_m);
  m.def("igemm_stats_rows", &igemm_stats_rows);
  m.def("igemm_force_config", &igemm_force_config);
  m.def("igemm_num_configs", &igemm_num_configs);

  m.def("wgrad_row3_splits", [](int NB, int H, int W, int N, int Cred, int R, int dtype) {
    int rps = 0;
    const int s = wgrad_row3_splits(NB, H, W, N, Cred, R, dtype, &rps);
    return std::make_pair(s, rps);
  });
  m.def("wgrad_row3", [](uintptr_t dY, uintptr_t X, uintptr_t slab, int NB, int H, int W, int N, int Cred, int R, int ph,
                         int ldy, int ldx, int splits, int rps, int dtype, uintptr_t s) {
    wgrad_row3_launch(ptr<const void>(dY), ptr<const void>(X), ptr<float>(slab), NB, H, W, N, Cred, R, ph, ldy, ldx, splits,
                      rps, dtype, st(s));
  });
  m.def("wgrad_row3_set_mode", &wgrad_row3_set_mode);
  m.def("wgrad_row3_set_variant", &wgrad_row3_set_variant);
  m.def("wgrad8_set_mode", &wgrad8_set_mode);
  m.def("wgrad8_splits", [](long M, int N, int N2, int Cred, int dtype, int ntaps, bool gather) -> py::tuple {
    int pps = 0;
    if (!wgrad8_covers(M, N, N2, Cred, ntaps, gather, dtype)) return py::make_tuple(0, 0);
    const int s = wgrad8_choose_splits(M, N, N2, ntaps * Cred, &pps);
    return py::make_tuple(s, pps);
  }, py::arg("M"), py::arg("N"), py::arg("N2"), py::arg("Cred"), py::arg("dtype"), py::arg("ntaps") = 1,
     py::arg("gather") = false);
  m.def("wgrad8", [](uintptr_t dY, uintptr_t X, uintptr_t slab, int M, int N, int Cc, int ldy, int ldx, int splits, int pps,
                     int dtype, uintptr_t stream, uintptr_t dY2, int N2, int ldy2) {
    WgradParams p;
    p.dY = ptr<const void>(dY); p.X = ptr<const void>(X); p.slab = ptr<float>(slab); p.taps = nullptr;
    p.dY2 = ptr<const void>(dY2); p.N2 = N2; p.ldy2 = ldy2;
    p.M = M; p.N = N; p.Cred = Cc; p.ntaps = 1; p.Ktot = Cc; p.ldy = ldy; p.ldx = ldx;
    p.IH = p.IW = 1; p.SY = p.SX = 1; p.splits = splits; p.pix_per_split = pps; p.taps_zero = 1;
    p.div_ohow = make_fastdiv(1u); p.div_ow = make_fastdiv(1u);
    wgrad8_launch(p, dtype, st(stream));
  }, py::arg("dY"), py::arg("X"), py::arg("slab"), py::arg("M"), py::arg("N"), py::arg("C"), py::arg("ldy"), py::arg("ldx"),
     py::arg("splits"), py::arg("pps"), py::arg("dtype"), py::arg("stream"), py::arg("dY2") = 0, py::arg("N2") = 0,
     py::arg("ldy2") = 0);
  m.def("wgrad_force_config", &wgrad_force_config);
  m.def("wgrad_num_configs", &wgrad_num_configs);
  m.def("wgrad_set_target_blocks", &wgrad_set_target_blocks);
  m.def("wgrad_choose_splits", [](long M, int N, int Ktot, int dtype, int groups) {
    int pps = 0;
    const int s = wgrad_choose_splits(M, N, Ktot, dtype, &pps, groups);
    return py::make_tuple(s, pps);
  }, py::arg("M"), py::arg("N"), py::arg("Ktot"), py::arg("dtype"), py::arg("groups") = 1);
  m.def("wgrad",
        [](uintptr_t dY, uintptr_t X, uintptr_t slab, uintptr_t taps, int M, int N, int Cred, int ntaps, int ldy,
           int ldx, int IH, int IW, int SY, int SX, int OH, int OW, int splits, int pps, int taps_zero, int dtype,
           uintptr_t stream, int groups, long y_gstride, long x_gstride, uintptr_t x_scale, uintptr_t x_shift,
           uintptr_t dY2, int N2, int ldy2, int use8) {
          WgradParams p;
          p.dY2 = ptr<const void>(dY2); p.N2 = N2; p.ldy2 = ldy2;
          p.x_scale = ptr<const float>(x_scale);
          p.x_shift = ptr<const float>(x_shift);
          p.groups = groups; p.y_gstride = y_gstride; p.x_gstride = x_gstride; p.slab_rows = groups * N;
          p.taps_zero = taps_zero;
          p.dY = ptr<const void>(dY);
          p.X = ptr<const void>(X);
          p.slab = ptr<float>(slab);
          p.taps = ptr<const int>(taps);
          p.M = M; p.N = N; p.Cred = Cred; p.ntaps = ntaps; p.Ktot = ntaps * Cred;
          p.ldy = ldy; p.ldx = ldx; p.IH = IH; p.IW = IW; p.SY = SY; p.SX = SX;
          p.splits = splits; p.pix_per_split = pps;
          p.div_ohow = make_fastdiv((uint32_t)(OH * OW));
          p.div_ow = make_fastdiv((uint32_t)OW);
          if (use8) wgrad8_launch(p, dtype, st(stream));      // splits from wgrad8_splits
          else wgrad_launch(p, dtype, st(stream));
        },
        py::arg("dY"), py::arg("X"), py::arg("slab"), py::arg("taps"), py::arg("M"), py::arg("N"), py::arg("Cred"),
        py::arg("ntaps"), py::arg("ldy"), py::arg("ldx"), py::arg("IH"), py::arg("IW"), py::arg("SY"), py::arg("SX"),
        py::arg("OH"), py::arg("OW"), py::arg("splits"), py::arg("pps"), py::arg("taps_zero"), py::arg("dtype"),
        py::arg("stream"), py::arg("groups") = 1, py::arg("y_gstride") = 0, py::arg("x_gstride") = 0,
        py::arg("x_scale") = 0, py::arg("x_shift") = 0, py::arg("dY2") = 0, py::arg("N2") = 0, py::arg("ldy2") = 0,
        py::arg("use8") = 0);
  m.def("slab_reduce", [](uintptr_t slab, int splits, long n, uintptr_t out, int out_dtype, float beta, uintptr_t s) {
    slab_reduce_launch(ptr<const float>(slab), splits, n, ptr<void>(out), out_dtype, beta, st(s));
  });

  // ---- batch norm
  m.def("bn_partial_rows", &bn_partial_rows);
  m.def("bn_stats", [](uintptr_t x, long M, int C, int dtype, uintptr_t partial, int nblk, uintptr_t s) {
    bn_stats_launch(ptr<const void>(x), M, C, dtype, ptr<float>(partial), nblk, st(s));
  });
  m.def("bn_fold_partials", [](uintptr_t partial, int nparts, int C, uintptr_t out, int nout, uintptr_t s) {
    bn_fold_partials_launch(ptr<const float>(partial), nparts, C, ptr<float>(out), nout, st(s));
  });
  m.def("bn_stats_floats", [](long nparts, long C) { return bn_stats_floats(nparts, C); });
  m.def("bn_fold_moments", [](uintptr_t partial, int nparts, int C, uintptr_t tmp, uintptr_t counter, double count,
                              uintptr_t mean, uintptr_t m2, uintptr_t count_out, uintptr_t s) {
    bn_fold_moments_launch(ptr<const float>(partial), nparts, C, ptr<double>(tmp), ptr<int>(counter), count,
                           ptr<float>(mean), ptr<float>(m2), ptr<float>(count_out), st(s));
  });
  m.def("bn_sync_merge", [](uintptr_t gathered, int W, int C, uintptr_t gamma, uintptr_t beta, float eps, float momentum,
                            uintptr_t rm, uintptr_t rv, uintptr_t nbt, uintptr_t mean, uintptr_t invstd, uintptr_t scale,
                            uintptr_t shift, uintptr_t count_out, uintptr_t s) {
    bn_sync_merge_launch(ptr<const float>(gathered), W, C, ptr<const float>(gamma), ptr<const float>(beta), eps, momentum,
                         ptr<float>(rm), ptr<float>(rv), ptr<long>(nbt), ptr<float>(mean), ptr<float>(invstd),
                         ptr<float>(scale), ptr<float>(shift), ptr<float>(count_out), st(s));
  });
  m.def("bn_eval_coeffs", [](uintptr_t gamma, uintptr_t beta, uintptr_t rm, uintptr_t rv, float eps, int C,
                             uintptr_t scale, uintptr_t shift, uintptr_t s) {
    bn_eval_coeffs_launch(ptr<const float>(gamma), ptr<const float>(beta), ptr<const float>(rm), ptr<const float>(rv),
                          eps, C, ptr<float>(scale), ptr<float>(shift), st(s));
  });
  m.def("bn_set_tuning", &bn_set_tuning);
  m.def("bn_set_reverse", &bn_set_reverse);
  m.def("bn_fold_chunks", &bn_fold_chunks);
  m.def("bn_fold_finalize", [](uintptr_t partial, int nparts, int C, uintptr_t tmp, uintptr_t counter, double count,
                               uintptr_t gamma, uintptr_t beta, float eps, float momentum, uintptr_t rm, uintptr_t rv,
                               uintptr_t nbt, uintptr_t mean, uintptr_t invstd, uintptr_t scale, uintptr_t shift,
                               uintptr_t s) {
    bn_fold_finalize_launch(ptr<const float>(partial), nparts, C, ptr<double>(tmp), ptr<int>(counter), count,
                            ptr<const float>(gamma), ptr<const float>(beta), eps, momentum, ptr<float>(rm),
                            ptr<float>(rv), ptr<long>(nbt), ptr<float>(mean), ptr<float>(invstd), ptr<float>(scale),
                            ptr<float>(shift), st(s));
  });
  m.def("bn_bwd_fold_finalize", [](uintptr_t partial, int nparts, int C, uintptr_t tmp, uintptr_t counter,
                                   double count, uintptr_t gamma, uintptr_t mean, uintptr_t invstd, uintptr_t scale,
                                   uintptr_t shift, uintptr_t dgamma, uintptr_t dbeta, uintptr_t coef, uintptr_t s) {
    bn_bwd_fold_finalize_launch(ptr<const float>(partial), nparts, C, ptr<double>(tmp), ptr<int>(counter), count,
                                ptr<const float>(gamma), ptr<const float>(mean), ptr<const float>(invstd),
                                ptr<const float>(scale), ptr<const float>(shift), ptr<float>(dgamma),
                                ptr<float>(dbeta), ptr<float>(coef), st(s));
  });
  m.def("bn_apply", [](uintptr_t x, uintptr_t res, uintptr_t y, long M, int C, int dtype, uintptr_t scale,
                       uintptr_t shift, int relu, uintptr_t s, uintptr_t mask, uintptr_t res_scale,
                       uintptr_t res_shift) {
    bn_apply_launch(ptr<const void>(x), ptr<const void>(res), ptr<void>(y), M, C, dtype, ptr<const float>(scale),
                    ptr<const float>(shift), relu, st(s), ptr<uint8_t>(mask), ptr<const float>(res_scale),
                    ptr<const float>(res_shift));
  }, py::arg("x"), py::arg("res"), py::arg("y"), py::arg("M"), py::arg("C"), py::arg("dtype"), py::arg("scale"),
     py::arg("shift"), py::arg("relu"), py::arg("stream"), py::arg("mask") = 0, py::arg("res_scale") = 0,
     py::arg("res_shift") = 0);
  m.def("bn_bwd_reduce", [](uintptr_t dy, uintptr_t x, uintptr_t y, long M, int C, int dtype, uintptr_t mean,
                            uintptr_t invstd, int relu, uintptr_t scale, uintptr_t shift, uintptr_t partial, int nblk,
                            uintptr_t s) {
    bn_bwd_reduce_launch(ptr<const void>(dy), ptr<const void>(x), ptr<const void>(y), M, C, dtype,
                         ptr<const float>(mean), ptr<const float>(invstd), relu, ptr<const float>(scale),
                         ptr<const float>(shift), ptr<float>(partial), nblk, st(s));
  });
  m.def("bn_apply_colsum_parts", &bn_apply_colsum_parts);
  m.def("bn_apply_colsum", [](uintptr_t x, uintptr_t y, long M, int C, int dtype, uintptr_t scale, uintptr_t shift,
                              uintptr_t part, int nparts, uintptr_t s) {
    bn_apply_colsum_launch(ptr<const void>(x), ptr<void>(y), M, C, dtype, ptr<const float>(scale), ptr<const float>(shift),
                           ptr<float>(part), nparts, st(s));
  });
  m.def("bn_bwd_finalize", [](uintptr_t partial, int nparts, int C, uintptr_t dgamma, uintptr_t dbeta, uintptr_t s) {
    bn_bwd_finalize_launch(ptr<const float>(partial), nparts, C, ptr<float>(dgamma), ptr<float>(dbeta), st(s));
  });
  m.def("bn_bwd_finalize_fused", [](uintptr_t partial, int nparts, int C, uintptr_t dgamma, uintptr_t dbeta,
                                     uintptr_t mean, uintptr_t invstd, uintptr_t gamma, double count, uintptr_t coef,
                                     uintptr_t scale, uintptr_t shift, uintptr_t s) {
    bn_bwd_finalize_fused_launch(ptr<const float>(partial), nparts, C, ptr<float>(dgamma), ptr<float>(dbeta),
                                 ptr<const float>(mean), ptr<const float>(invstd), ptr<const float>(gamma), count,
                                 ptr<float>(coef), ptr<const float>(scale), ptr<const float>(shift), st(s));
  });
  m.def("bn_bwd_coeffs", [](uintptr_t dgamma, uintptr_t dbeta, uintptr_t mean, uintptr_t invstd, uintptr_t gamma,
                            double count, uintptr_t count_dev, int C, uintptr_t coef, uintptr_t scale, uintptr_t shift,
                            uintptr_t s) {
    bn_bwd_coeffs_launch(ptr<const float>(dgamma), ptr<const float>(dbeta), ptr<const float>(mean),
                         ptr<const float>(invstd), ptr<const float>(gamma), count, ptr<const float>(count_dev), C,
                         ptr<float>(coef), ptr<const float>(scale), ptr<const float>(shift), st(s));
  });
  m.def("bn_bwd_apply", [](uintptr_t dy, uintptr_t x, uintptr_t y, uintptr_t dx, uintptr_t dres, long M, int C,
                           int dtype, uintptr_t coef, int relu, uintptr_t s) {
    bn_bwd_apply_launch(ptr<const void>(dy), ptr<const void>(x), ptr<const void>(y), ptr<void>(dx), ptr<void>(dres), M,
                        C, dtype, ptr<const float>(coef), relu, st(s));
  });

  // ---- pooling
  m.def("maxpool_fwd", [](uintptr_t x, uintptr_t y, uintptr_t idx, int NB, int IH, int IW, int C, int OH, int OW, int kh, int kw,
                          int sh, int sw, int ph, int pw, int dtype, uintptr_t s) {
    maxpool_fwd_launch(ptr<const void>(x), ptr<void>(y), ptr<uint8_t>(idx), NB, IH, IW, C, OH, OW, kh, kw, sh, sw, ph, pw, dtype, st(s));
  });
  m.def("bn_relu_maxpool_partial_rows", &bn_relu_maxpool_partial_rows);
  m.def("bn_relu_maxpool_fwd", [](uintptr_t x, uintptr_t y, uintptr_t idx, uintptr_t scale, uintptr_t shift, int NB, int IH,
                                  int IW, int C, int OH, int OW, int kh, int kw, int sh, int sw, int ph, int pw, int dtype,
                                  uintptr_t s) {
    bn_relu_maxpool_fwd_launch(ptr<const void>(x), ptr<void>(y), ptr<uint8_t>(idx), ptr<const float>(scale),
                               ptr<const float>(shift), NB, IH, IW, C, OH, OW, kh, kw, sh, sw, ph, pw, dtype, st(s));
  });
  m.def("bn_relu_maxpool_bwd_reduce", [](uintptr_t idx, uintptr_t dyp, uintptr_t x, uintptr_t bn, uintptr_t partial,
                                         int nblk, int NB, int IH, int IW, int C, int OH, int OW, int kh, int kw, int sh,
                                         int sw, int ph, int pw, int dtype, uintptr_t s) {
    bn_relu_maxpool_bwd_reduce_launch(ptr<const uint8_t>(idx), ptr<const void>(dyp), ptr<const void>(x),
                                      ptr<const float>(bn), ptr<float>(partial), nblk, NB, IH, IW, C, OH, OW, kh, kw, sh,
                                      sw, ph, pw, dtype, st(s));
  });
  m.def("bn_relu_maxpool_bwd_apply", [](uintptr_t idx, uintptr_t dyp, uintptr_t x, uintptr_t bn, uintptr_t coef,
                                        uintptr_t dx, int NB, int IH, int IW, int C, int OH, int OW, int kh, int kw, int sh,
                                        int sw, int ph, int pw, int dtype, uintptr_t s) {
    bn_relu_maxpool_bwd_apply_launch(ptr<const uint8_t>(idx), ptr<const void>(dyp), ptr<const void>(x),
                                     ptr<const float>(bn), ptr<const float>(coef), ptr<void>(dx), NB, IH, IW, C, OH, OW,
                                     kh, kw, sh, sw, ph, pw, dtype, st(s));
  });
  m.def("maxpool_bwd", [](uintptr_t x, uintptr_t dy, uintptr_t dx, int NB, int IH, int IW, int C, int OH, int OW,
                          int kh, int kw, int sh, int sw, int ph, int pw, int dtype, uintptr_t s) {
    maxpool_bwd_launch(ptr<const uint8_t>(x), ptr<const void>(dy), ptr<void>(dx), NB, IH, IW, C, OH, OW, kh, kw, sh, sw,
                       ph, pw, dtype, st(s));
  });
  m.def("avgpool_fwd", [](uintptr_t x, uintptr_t y, int NB, int IH, int IW, int C, int OH, int OW, int dtype,
                          uintptr_t s) {
    avgpool_fwd_launch(ptr<const void>(x), ptr<void>(y), NB, IH, IW, C, OH, OW, dtype, st(s));
  });
  m.def("avgpool_win", [](uintptr_t src, uintptr_t dst, int backward, int NB, int IH, int IW, int C, int OH, int OW,
                          int kh, int kw, int sh, int sw, int ph, int pw, int dtype, uintptr_t s) {
    avgpool_win_launch(ptr<const void>(src), ptr<void>(dst), backward, NB, IH, IW, C, OH, OW, kh, kw, sh, sw, ph, pw,
                       dtype, st(s));
  });
  m.def("avgpool_bwd", [](uintptr_t dy, uintptr_t dx, int NB, int IH, int IW, int C, int OH, int OW, int dtype,
                          uintptr_t s) {
    avgpool_bwd_launch(ptr<const void>(dy), ptr<void>(dx), NB, IH, IW, C, OH, OW, dtype, st(s));
  });

  // ---- loss
  m.def("xent_fwd_bwd", [](uintptr_t logits, uintptr_t target, int N, int C, int ldl, int dtype, uintptr_t loss,
                           uintptr_t dlogits, uintptr_t rowbuf, uintptr_t metrics, uintptr_t s) {
    xent_fwd_bwd_launch(ptr<const void>(logits), ptr<const long>(target), N, C, ldl, dtype, ptr<float>(loss),
                        ptr<void>(dlogits), ptr<float>(rowbuf), ptr<float>(metrics), st(s));
  });
  m.def("xent_bwd_scale", [](uintptr_t dlogits, long n, int dtype, uintptr_t gout, uintptr_t s) {
    xent_bwd_scale_launch(ptr<void>(dlogits), n, dtype, ptr<const float>(gout), st(s));
  });

  // ---- elementwise
  m.def("normalize_u8", [](uintptr_t x, uintptr_t y, long npix, int cin, int cout, int hw, int nchw, uintptr_t mean,
                           uintptr_t istd, int dtype, uintptr_t s) {
    normalize_u8_launch(ptr<const uint8_t>(x), ptr<void>(y), npix, cin, cout, hw, nchw, ptr<const float>(mean),
                        ptr<const float>(istd), dtype, st(s));
  });
  m.def("to_s2d", [](uintptr_t x, uintptr_t y, int NB, int H2, int W2, int nchw, int in_dtype, uintptr_t mean,
                     uintptr_t inv_std, int out_dtype, uintptr_t s, int padl, int padr) {
    to_s2d_launch(ptr<const void>(x), ptr<void>(y), NB, H2, W2, nchw, in_dtype, ptr<const float>(mean),
                  ptr<const float>(inv_std), out_dtype, st(s), padl, padr);
  }, py::arg("x"), py::arg("y"), py::arg("NB"), py::arg("H2"), py::arg("W2"), py::arg("nchw"), py::arg("in_dtype"),
     py::arg("mean"), py::arg("inv_std"), py::arg("out_dtype"), py::arg("s"), py::arg("padl") = 0, py::arg("padr") = 0);
  m.def("nchw_to_nhwc", [](uintptr_t x, uintptr_t y, int NB, int C, int HW, int cpad, int in_dtype, int out_dtype,
                           uintptr_t s) {
    nchw_to_nhwc_launch(ptr<const void>(x), ptr<void>(y), NB, C, HW, cpad, in_dtype, out_dtype, st(s));
  });
  m.def("relu_fwd", [](uintptr_t x, uintptr_t y, long n, int dtype, uintptr_t s) {
    relu_fwd_launch(ptr<const void>(x), ptr<void>(y), n, dtype, st(s));
  });
  m.def("relu_bwd", [](uintptr_t y, uintptr_t dy, uintptr_t dx, long n, int dtype, uintptr_t s) {
    relu_bwd_launch(ptr<const void>(y), ptr<const void>(dy), ptr<void>(dx), n, dtype, st(s));
  });
  m.def("gemm8", [](uintptr_t A, uintptr_t B, uintptr_t Cc, uintptr_t bias, int M, int N, int K, long lda, long ldb, long ldc,
                    int relu, int dtype, uintptr_t s) {
    gemm8_launch(ptr<const void>(A), ptr<const void>(B), ptr<void>(Cc), ptr<const float>(bias), M, N, K, lda, ldb, ldc, relu,
                 dtype, st(s));
  });
  m.def("dw_conv_rows", &dw_conv_rows);
  m.def("dw_set_sliding", &dw_set_sliding);
  m.def("dw_wgrad_rows", &dw_wgrad_rows);
  m.def("dw_conv", [](uintptr_t in, uintptr_t w, int w_dtype, uintptr_t bias, uintptr_t out, uintptr_t stats, int backward,
                      int NB, int IH, int IW, int C, int OH, int OW, int KH, int KW, int sh, int sw, int ph, int pw,
                      int relu, int dtype, uintptr_t s) {
    dw_conv_launch(ptr<const void>(in), ptr<const void>(w), w_dtype, ptr<const float>(bias), ptr<void>(out),
                   ptr<float>(stats), backward, NB, IH, IW, C, OH, OW, KH, KW, sh, sw, ph, pw, relu, dtype, st(s));
  });
  m.def("dw_wgrad", [](uintptr_t dy, uintptr_t x, uintptr_t partial, int NB, int IH, int IW, int C, int OH, int OW, int KH,
                       int KW, int sh, int sw, int ph, int pw, int dtype, uintptr_t s) {
    dw_wgrad_launch(ptr<const void>(dy), ptr<const void>(x), ptr<float>(partial), NB, IH, IW, C, OH, OW, KH, KW, sh, sw,
                    ph, pw, dtype, st(s));
  });
  m.def("ln_partial_rows", &ln_partial_rows);
  m.def("ln_fwd", [](uintptr_t x, uintptr_t w, uintptr_t b, uintptr_t y, uintptr_t mean, uintptr_t rstd, long rows, int C,
                     float eps, int dtype, uintptr_t s) {
    ln_fwd_launch(ptr<const void>(x), ptr<const float>(w), ptr<const float>(b), ptr<void>(y), ptr<float>(mean),
                  ptr<float>(rstd), rows, C, eps, dtype, st(s));
  });
  m.def("ln_bwd", [](uintptr_t dy, uintptr_t x, uintptr_t w, uintptr_t mean, uintptr_t rstd, uintptr_t dx,
                     uintptr_t partial, uintptr_t dgamma, uintptr_t dbeta, float beta_w, float beta_b, long rows, int C,
                     int dtype, uintptr_t s) {
    ln_bwd_launch(ptr<const void>(dy), ptr<const void>(x), ptr<const float>(w), ptr<const float>(mean),
                  ptr<const float>(rstd), ptr<void>(dx), ptr<float>(partial), ptr<float>(dgamma), ptr<float>(dbeta),
                  beta_w, beta_b, rows, C, dtype, st(s));
  });
  m.def("add_fwd", [](uintptr_t a, uintptr_t b, uintptr_t y, long n, int dtype, uintptr_t s) {
    add_fwd_launch(ptr<const void>(a), ptr<const void>(b), ptr<void>(y), n, dtype, st(s));
  });
  m.def("act_fwd", [](uintptr_t x, uintptr_t y, long n, int act, int dtype, uintptr_t s) {
    act_fwd_launch(ptr<const void>(x), ptr<void>(y), n, act, dtype, st(s));
  });
  m.def("act_bwd", [](uintptr_t x, uintptr_t dy, uintptr_t dx, long n, int act, int dtype, uintptr_t s) {
    act_bwd_launch(ptr<const void>(x), ptr<const void>(dy), ptr<void>(dx), n, act, dtype, st(s));
  });
  m.def("se_scale_fwd", [](uintptr_t x, uintptr_t sc, uintptr_t y, long NB, int HW, int C, int dtype, uintptr_t s,
                           uintptr_t add) {
    se_scale_fwd_launch(ptr<const void>(x), ptr<const float>(sc), ptr<void>(y), NB, HW, C, dtype, st(s),
                        ptr<const float>(add));
  }, py::arg("x"), py::arg("sc"), py::arg("y"), py::arg("NB"), py::arg("HW"), py::arg("C"), py::arg("dtype"),
     py::arg("stream"), py::arg("add") = 0);
  m.def("se_scale_bwd", [](uintptr_t dy, uintptr_t x, uintptr_t sc, uintptr_t dx, uintptr_t ds, long NB, int HW, int C,
                           int dtype, uintptr_t s) {
    se_scale_bwd_launch(ptr<const void>(dy), ptr<const void>(x), ptr<const float>(sc), ptr<void>(dx), ptr<float>(ds), NB,
                        HW, C, dtype, st(s));
  });
  m.def("add_relu_fwd", [](uintptr_t a, uintptr_t b, uintptr_t y, long n, int dtype, uintptr_t s) {
    add_relu_fwd_launch(ptr<const void>(a), ptr<const void>(b), ptr<void>(y), n, dtype, st(s));
  });
  m.def("dropout_fwd", [](uintptr_t x, uintptr_t y, uintptr_t mask, long n, float p, uint64_t seed, uint64_t offset,
                          uintptr_t counter, int dtype, uintptr_t s) {
    dropout_fwd_launch(ptr<const void>(x), ptr<void>(y), ptr<uint8_t>(mask), n, p, seed, offset,
                       ptr<uint64_t>(counter), dtype, st(s));
  });
  m.def("dropout_bwd", [](uintptr_t dy, uintptr_t mask, uintptr_t dx, long n, float p, int dtype, uintptr_t s) {
    dropout_bwd_launch(ptr<const void>(dy), ptr<const uint8_t>(mask), ptr<void>(dx), n, p, dtype, st(s));
  });
  m.def("cast", [](uintptr_t x, int in_dtype, uintptr_t y, int out_dtype, long n, uintptr_t s) {
    cast_launch(ptr<const void>(x), in_dtype, ptr<void>(y), out_dtype, n, st(s));
  });
  m.def("transpose_ktc", [](uintptr_t w, uintptr_t wt, int K, int T, int C, int in_dtype, int out_dtype, uintptr_t s) {
    transpose_ktc_launch(ptr<const void>(w), ptr<void>(wt), K, T, C, in_dtype, out_dtype, st(s));
  });
  m.def("se_mlp_fwd", [](uintptr_t p, uintptr_t w1, uintptr_t b1, uintptr_t w2, uintptr_t b2, uintptr_t z1, uintptr_t h,
                         uintptr_t z2, uintptr_t s_, int N, int C, int S, int act, int gate, uintptr_t s) {
    se_mlp_fwd_launch(ptr<const float>(p), ptr<const float>(w1), ptr<const float>(b1), ptr<const float>(w2),
                      ptr<const float>(b2), ptr<float>(z1), ptr<float>(h), ptr<float>(z2), ptr<float>(s_), N, C, S, act,
                      gate, st(s));
  });
  m.def("se_mlp_bwd", [](uintptr_t ds, uintptr_t s_, uintptr_t z2, uintptr_t z1, uintptr_t p, uintptr_t h, uintptr_t w1,
                         uintptr_t w2t, uintptr_t dz2, uintptr_t dz1, uintptr_t dp, uintptr_t dw1, uintptr_t db1,
                         uintptr_t dw2, uintptr_t db2, int N, int C, int S, int act, int gate, float inv_hw, uintptr_t s) {
    se_mlp_bwd_launch(ptr<const float>(ds), ptr<const float>(s_), ptr<const float>(z2), ptr<const float>(z1),
                      ptr<const float>(p), ptr<const float>(h), ptr<const float>(w1), ptr<const float>(w2t),
                      ptr<float>(dz2), ptr<float>(dz1), ptr<float>(dp), ptr<float>(dw1), ptr<float>(db1), ptr<float>(dw2),
                      ptr<float>(db2), N, C, S, act, gate, inv_hw, st(s));
  });
  m.def("attn_max_tokens", &attn_max_tokens);
  m.def("window_partition", [](uintptr_t src, uintptr_t dst, int reverse, int B, int H, int W, int C, int wh, int ww, int nh,
                               int nw, int sh, int sw, int grid_mode, int dtype, uintptr_t s) {
    window_partition_launch(ptr<const void>(src), ptr<void>(dst), reverse, B, H, W, C, wh, ww, nh, nw, sh, sw, grid_mode,
                            dtype, st(s));
  });
  m.def("qk_normalize", [](uintptr_t x, uintptr_t dy, uintptr_t out, uintptr_t mult, uintptr_t inv, uintptr_t dmult_part,
                           long tokens, int H, int D, int backward, int dtype, uintptr_t s) {
    qk_normalize_launch(ptr<const void>(x), ptr<const void>(dy), ptr<void>(out), ptr<const float>(mult), ptr<float>(inv),
                        ptr<float>(dmult_part), tokens, H, D, backward, dtype, st(s));
  });
  m.def("attn_dbias_splits", &attn_dbias_splits);
  m.def("attn_fwd", [](uintptr_t qkv, uintptr_t out, uintptr_t lse, uintptr_t bias, int nbias, uintptr_t scale, int B, int H,
                       int S, int D, int dtype, uintptr_t s) {
    attn_fwd_launch(ptr<const void>(qkv), ptr<void>(out), ptr<float>(lse), ptr<const float>(bias), nbias,
                    ptr<const float>(scale), B, H, S, D, dtype, st(s));
  });
  m.def("attn_bwd", [](uintptr_t qkv, uintptr_t out, uintptr_t dout, uintptr_t lse, uintptr_t delta, uintptr_t bias,
                       int nbias, uintptr_t scale, uintptr_t dqkv, uintptr_t dbias, int nsplit, int B, int H, int S, int D,
                       int dtype, uintptr_t s) {
    attn_bwd_launch(ptr<const void>(qkv), ptr<const void>(out), ptr<const void>(dout), ptr<const float>(lse),
                    ptr<float>(delta), ptr<const float>(bias), nbias, ptr<const float>(scale), ptr<void>(dqkv),
                    ptr<float>(dbias), nsplit, B, H, S, D, dtype, st(s));
  });
  m.def("shuffle_pair", [](uintptr_t a, uintptr_t b, uintptr_t o0, uintptr_t o1, long rows, int bf, int P, int backward,
                           int dtype, uintptr_t s) {
    shuffle_pair_launch(ptr<const void>(a), ptr<const void>(b), ptr<void>(o0), ptr<void>(o1), rows, bf, P, backward, dtype,
                        st(s));
  });
  m.def("subsample", [](uintptr_t src, uintptr_t dst, long NB, int IH, int IW, long row_bytes, int stride, uintptr_t s) {
    subsample_launch(ptr<const void>(src), ptr<void>(dst), NB, IH, IW, row_bytes, stride, st(s));
  });
  m.def("copy_rows", [](uintptr_t src, long ld_src, uintptr_t dst, long ld_dst, long rows, long row_bytes, uintptr_t s) {
    copy_rows_launch(ptr<const void>(src), ld_src, ptr<void>(dst), ld_dst, rows, row_bytes, st(s));
  });
  m.def("axpby", [](uintptr_t x, int x_dtype, uintptr_t y, long n, float alpha, float beta, uintptr_t s) {
    axpby_launch(ptr<const void>(x), x_dtype, ptr<float>(y), n, alpha, beta, st(s));
  });

  // ---- optimizer / loss scaler
  m.def("sgd_chunk_elems", &sgd_chunk_elems);
  m.def("sgd_table_entry_bytes", []() { return (int)sizeof(SgdTensor); });
  m.def("sgd_multi_tensor", [](uintptr_t table, int ntensors, long total_chunks, int g_dtype, int lp_dtype,
                               uintptr_t hyper, uintptr_t inv_scale, uintptr_t found_inf, uintptr_t s) {
    sgd_multi_tensor_launch(ptr<const SgdTensor>(table), ntensors, total_chunks, g_dtype, lp_dtype,
                            ptr<const float>(hyper), ptr<const float>(inv_scale), ptr<const float>(found_inf), st(s));
  });
  m.def("check_finite", [](uintptr_t g, long n, int dtype, uintptr_t found_inf, uintptr_t s) {
    check_finite_flat_launch(ptr<const void>(g), n, dtype, ptr<float>(found_inf), st(s));
  });
  m.def("scaler_update", [](uintptr_t scale, uintptr_t inv_scale, uintptr_t good, uintptr_t found_inf, int interval,
                            float growth, float backoff, float max_scale, float min_scale, uintptr_t s) {
    scaler_update_launch(ptr<float>(scale), ptr<float>(inv_scale), ptr<int>(good), ptr<float>(found_inf), interval,
                         growth, backoff, max_scale, min_scale, st(s));
  });

  bind_comm(m);
  bind_store(m);
  bind_p2p(m);
  bind_peer(m);
}
