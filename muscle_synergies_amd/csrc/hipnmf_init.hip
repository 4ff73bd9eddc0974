// hipnmf_init.hip -- C ABI of the on-device NNDSVD building blocks (include/hip_nmf.h, row f-2).
#include <algorithm>
#include <cstdint>

#include "hipnmf_internal.hpp"
#include "init_kernels.hpp"
#include "nmf_kernels.hpp"  // x_to_channel_major_kernel

using namespace hipnmf;

namespace {

// validates the shape fields of *p and returns X in channel-major form (in place or converted into workspace)
template <typename real>
int canonical_x(hipnmf_handle* h, const hipnmf_problem* p, const real* X, InitArgs* a) {
  if (!h) return fail(HIPNMF_ERR_BAD_ARG, "handle is NULL");
  if (!p || p->struct_size != (int32_t)sizeof(hipnmf_problem)) return fail(HIPNMF_ERR_BAD_ARG, "bad hipnmf_problem");
  if (!X) return fail(HIPNMF_ERR_BAD_ARG, "X is NULL");
  if (p->batch < 1 || p->n_samples < 1 || p->n_samples > 2000000000LL || p->n_features < 1 || p->n_components < 1)
    return fail(HIPNMF_ERR_BAD_ARG, "bad shape");
  if (p->n_features > GRAM_MAXM || p->n_components > 8)
    return fail(HIPNMF_ERR_UNSUPPORTED, "init kernels support n_features <= %d and n_components <= 8", GRAM_MAXM);
  if (p->x_layout != HIPNMF_X_ROW_MAJOR && p->x_layout != HIPNMF_X_CHANNEL_MAJOR)
    return fail(HIPNMF_ERR_BAD_ARG, "bad x_layout %d", p->x_layout);
  const long long min_ld = (p->x_layout == HIPNMF_X_ROW_MAJOR) ? p->n_features : p->n_samples;
  if (p->ldx < min_ld) return fail(HIPNMF_ERR_BAD_ARG, "ldx too small");
  HIP_TRY(hipSetDevice(h->device));
  const int B = p->batch, m = p->n_features;
  const long long T = p->n_samples;
  a->T = (int)T;
  a->m = m;
  a->k = p->n_components;
  if (p->x_layout == HIPNMF_X_CHANNEL_MAJOR) {
    a->X = X;
    a->bstride = p->x_batch_stride;
    a->ld = p->ldx;
    return HIPNMF_OK;
  }
  int rc = hipnmf_ensure_ws(h, sizeof(real) * (size_t)B * m * T);
  if (rc) return rc;
  real* xc = static_cast<real*>(h->ws);
  dim3 blk(32, 8);
  dim3 grd((unsigned)((T + 31) / 32), (unsigned)((m + 31) / 32), (unsigned)B);
  hipLaunchKernelGGL(x_to_channel_major_kernel<real>, grd, blk, 0, h->stream, X, (long long)p->x_batch_stride,
                     (long long)p->ldx, (int)p->x_layout, xc, (long long)m * T, T, (int)T, m);
  a->X = xc;
  a->bstride = (long long)m * T;
  a->ld = T;
  return HIPNMF_OK;
}

int finish(hipnmf_handle* h) {
  HIP_TRY(hipGetLastError());
  if (!h->async_mode) HIP_TRY(hipStreamSynchronize(h->stream));
  return HIPNMF_OK;
}

template <typename real>
int gram_impl(hipnmf_handle* h, const hipnmf_problem* p, const real* X, double* gram, double* colsum) {
  if (!gram || !colsum) return fail(HIPNMF_ERR_BAD_ARG, "gram and colsum must be non-NULL");
  InitArgs a{};
  int rc = canonical_x<real>(h, p, X, &a);
  if (rc) return rc;
  a.gram = gram;
  a.colsum = colsum;
  hipLaunchKernelGGL(gram_kernel<real>, dim3(p->batch), dim3(256), 0, h->stream, a);
  return finish(h);
}

template <typename real>
int stats_impl(hipnmf_handle* h, const hipnmf_problem* p, const real* X, const double* V, const double* inv_s,
               double* stats) {
  if (!V || !inv_s || !stats) return fail(HIPNMF_ERR_BAD_ARG, "V, inv_s and stats must be non-NULL");
  InitArgs a{};
  int rc = canonical_x<real>(h, p, X, &a);
  if (rc) return rc;
  a.V = V;
  a.inv_s = inv_s;
  a.stats = stats;
  hipLaunchKernelGGL(nndsvd_stats_kernel<real>, dim3(p->batch), dim3(256), 0, h->stream, a);
  return finish(h);
}

template <typename real>
int write_impl(hipnmf_handle* h, const hipnmf_problem* p, const real* X, const double* V, const double* inv_s,
               const double* coef, const double* fill, double eps, real* W0) {
  if (!V || !inv_s || !coef || !fill || !W0) return fail(HIPNMF_ERR_BAD_ARG, "NULL argument");
  InitArgs a{};
  int rc = canonical_x<real>(h, p, X, &a);
  if (rc) return rc;
  a.V = V;
  a.inv_s = inv_s;
  a.coef = coef;
  a.fill = fill;
  a.eps = eps;
  a.W0 = W0;
  hipLaunchKernelGGL(nndsvd_write_kernel<real>, dim3(p->batch), dim3(256), 0, h->stream, a);
  return finish(h);
}

}  // namespace

extern "C" {
int hipnmf_gram_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, double* gram, double* colsum) {
  return gram_impl<float>(h, p, X, gram, colsum);
}
int hipnmf_gram_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, double* gram, double* colsum) {
  return gram_impl<double>(h, p, X, gram, colsum);
}
int hipnmf_nndsvd_stats_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, const double* V,
                            const double* inv_s, double* stats) {
  return stats_impl<float>(h, p, X, V, inv_s, stats);
}
int hipnmf_nndsvd_stats_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, const double* V,
                            const double* inv_s, double* stats) {
  return stats_impl<double>(h, p, X, V, inv_s, stats);
}
int hipnmf_nndsvd_write_f32(hipnmf_handle* h, const hipnmf_problem* p, const float* X, const double* V,
                            const double* inv_s, const double* coef, const double* fill, double eps, float* W0) {
  return write_impl<float>(h, p, X, V, inv_s, coef, fill, eps, W0);
}
int hipnmf_nndsvd_write_f64(hipnmf_handle* h, const hipnmf_problem* p, const double* X, const double* V,
                            const double* inv_s, const double* coef, const double* fill, double eps, double* W0) {
  return write_impl<double>(h, p, X, V, inv_s, coef, fill, eps, W0);
}
}
