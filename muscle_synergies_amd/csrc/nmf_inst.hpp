// nmf_inst.hpp -- kernel tables: one translation unit per (real, G, CH), eight ranks (K = 1..8) each,
// so that `make -j` compiles them in parallel.
#pragma once
#include "nmf_kernels.hpp"

namespace hipnmf {

template <typename real>
struct KernelSet {
  using Fn = void (*)(SolveArgs<real>);
  Fn fit_persistent, slice_pass, reduce_slices, hupdate, slice_resid, resid_finalize;
  Fn fit_persistent_kl;  // Kullback-Leibler loss (persistent path only); nullptr where not built
  Fn fit_coop_xcd;      // cooperative kernel, same-XCD exchange (one matrix)
  Fn fit_coop;           // cooperative multi-workgroup fit of few long matrices (Frobenius); nullptr where not built
  int G, CH, K, MP, NACC, max_threads;
  bool row_major;  // the instance streams X row-major (rows of MP values) instead of channel-major
  size_t (*smem_bytes)(int nw);
};

template <typename real, int G, int CH, int K>
KernelSet<real> make_kernel_set() {
  KernelSet<real> ks;
  ks.fit_persistent = fit_persistent_kernel<real, G, CH, K>;
  ks.fit_persistent_kl = fit_persistent_kernel<real, G, CH, K, 1>;  // H in registers whatever the Frobenius kernels do
  ks.fit_coop = fit_coop_kernel<real, G, CH, K>;
  ks.fit_coop_xcd = fit_coop_kernel<real, G, CH, K, true>;
  ks.slice_pass = slice_pass_kernel<real, G, CH, K>;
  ks.reduce_slices = reduce_slices_kernel<real, G, CH, K>;
  ks.hupdate = hupdate_kernel<real, G, CH, K>;
  ks.slice_resid = slice_resid_kernel<real, G, CH, K>;
  ks.resid_finalize = resid_finalize_kernel<real, G, CH, K>;
  ks.row_major = x_row_major<G, CH>();
  ks.G = G;
  ks.CH = CH;
  ks.K = K;
  ks.MP = G * CH;
  ks.NACC = Cfg<real, G, CH, K>::NACC;
  ks.max_threads = max_threads<real, G, CH, K>();
  ks.smem_bytes = &Smem<real, G, CH, K>::bytes;
  return ks;
}

#define HIPNMF_DECLARE_TABLE(REAL, TAG) const KernelSet<REAL>* kernels_##TAG(int K);

HIPNMF_DECLARE_TABLE(float, f32_g1c4)
HIPNMF_DECLARE_TABLE(float, f32_g2c4)
HIPNMF_DECLARE_TABLE(float, f32_g4c4)
HIPNMF_DECLARE_TABLE(float, f32_g2c8)
HIPNMF_DECLARE_TABLE(float, f32_g1c16)
HIPNMF_DECLARE_TABLE(float, f32_g1c8)
HIPNMF_DECLARE_TABLE(float, f32_g4c8)
HIPNMF_DECLARE_TABLE(double, f64_g1c8)
HIPNMF_DECLARE_TABLE(double, f64_g1c4)
HIPNMF_DECLARE_TABLE(double, f64_g2c4)
HIPNMF_DECLARE_TABLE(double, f64_g4c4)
HIPNMF_DECLARE_TABLE(double, f64_g4c8)

#define HIPNMF_DEFINE_TABLE(REAL, TAG, G, CH)                                                          \
  const KernelSet<REAL>* kernels_##TAG(int K) {                                                        \
    static const KernelSet<REAL> tbl[8] = {                                                            \
        make_kernel_set<REAL, G, CH, 1>(), make_kernel_set<REAL, G, CH, 2>(),                          \
        make_kernel_set<REAL, G, CH, 3>(), make_kernel_set<REAL, G, CH, 4>(),                          \
        make_kernel_set<REAL, G, CH, 5>(), make_kernel_set<REAL, G, CH, 6>(),                          \
        make_kernel_set<REAL, G, CH, 7>(), make_kernel_set<REAL, G, CH, 8>()};                         \
    return (K >= 1 && K <= 8) ? &tbl[K - 1] : nullptr;                                                 \
  }

}  // namespace hipnmf
