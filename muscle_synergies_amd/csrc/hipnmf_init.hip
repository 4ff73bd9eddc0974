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
  if (p->n_features > HIPNMF_MAX_FEATURES || p->n_components > HIPNMF_MAX_COMPONENTS)
    return fail(HIPNMF_ERR_UNSUPPORTED, "shape outside the compiled kernel set: n_features=%d (max %d), n_components=%d (max %d)",
                p->n_features, HIPNMF_MAX_FEATURES, p->n_components, HIPNMF_MAX_COMPONENTS);
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
  for (int b0 = 0; b0 < B; b0 += 65535) {  // the batch rides on grid.z (HIP limit 65535)
    dim3 grd((unsigned)((T + 31) / 32), (unsigned)((m + 31) / 32), (unsigned)std::min(65535, B - b0));
    HIPNMF_LAUNCH(x_to_channel_major_kernel<real>, grd, blk, 0, h->stream, X + (long long)b0 * p->x_batch_stride,
                       (long long)p->x_batch_stride, (long long)p->ldx, (int)p->x_layout, xc + (size_t)b0 * m * T, (long long)m * T, T,
                       (int)T, m);
  }
  a->X = xc;
  a->bstride = (long long)m * T;
  a->ld = T;
  return HIPNMF_OK;
}

// the Gram matrix in GRAM_BLK x GRAM_BLK blocks, the batch on grid.y
template <typename real>
void launch_gram(hipnmf_handle* h, int B, const InitArgs& a) {
  const int nblk = (a.m + GRAM_BLK - 1) / GRAM_BLK;
  for (int b0 = 0; b0 < B; b0 += 65535) {
    InitArgs ab = a;
    ab.X = static_cast<const real*>(a.X) + (long long)b0 * a.bstride;
    ab.gram = a.gram + (size_t)b0 * a.m * a.m;
    ab.colsum = a.colsum + (size_t)b0 * a.m;
    HIPNMF_LAUNCH(gram_kernel<real>, dim3(nblk * nblk, std::min(65535, B - b0)), dim3(256), 0, h->stream, ab);
  }
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
  launch_gram<real>(h, p->batch, a);
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
  HIPNMF_LAUNCH(nndsvd_stats_kernel<real>, dim3(p->batch, (a.k + NNDSVD_KB - 1) / NNDSVD_KB), dim3(256),
                     sizeof(double) * NNDSVD_KB * (size_t)a.m, h->stream, a);
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
  HIPNMF_LAUNCH(nndsvd_write_kernel<real>, dim3(p->batch, (a.k + NNDSVD_KB - 1) / NNDSVD_KB), dim3(256),
                     sizeof(double) * NNDSVD_KB * (size_t)a.m, h->stream, a);
  return finish(h);
}

template <typename real>
int random_init_impl(hipnmf_handle* h, const hipnmf_problem* p, uint64_t seed, int first_matrix, const real* X, real* W, real* H,
                     const int* index = nullptr) {
  if (!W || !H) return fail(HIPNMF_ERR_BAD_ARG, "W and H must be non-NULL");
  if (p && p->w_layout != HIPNMF_W_ROW_MAJOR && p->w_layout != HIPNMF_W_COMPONENT_MAJOR)
    return fail(HIPNMF_ERR_BAD_ARG, "bad w_layout %d", p->w_layout);
  InitArgs a{};
  int rc;
  size_t x_bytes = 0;
  // column sums through the Gram kernel (fp64, fixed order); they live behind a possibly converted X in the workspace,
  // which therefore gets its final size BEFORE canonical_x converts into it (growing it afterwards would lose X)
  if (h && p && p->struct_size == (int32_t)sizeof(hipnmf_problem) && p->batch >= 1 && p->n_samples >= 1 &&
      p->n_samples <= 2000000000LL && p->n_features >= 1 && p->n_features <= HIPNMF_MAX_FEATURES) {
    const size_t Bm = (size_t)p->batch * p->n_features;
    x_bytes = (p->x_layout == HIPNMF_X_CHANNEL_MAJOR) ? 0 : (sizeof(real) * Bm * (size_t)p->n_samples + 255) / 256 * 256;
    HIP_TRY(hipSetDevice(h->device));
    rc = hipnmf_ensure_ws(h, x_bytes + sizeof(double) * (Bm * p->n_features + Bm));
    if (rc) return rc;
  }
  if (p && (p->n_features > HIPNMF_MAX_FEATURES || p->n_components > HIPNMF_MAX_COMPONENTS))
    return fail(HIPNMF_ERR_UNSUPPORTED, "shape outside the compiled kernel set: n_features=%d (max %d), n_components=%d (max %d)",
                p->n_features, HIPNMF_MAX_FEATURES, p->n_components, HIPNMF_MAX_COMPONENTS);
  rc = canonical_x<real>(h, p, X, &a);  // validates everything (and reports what the shortcut above skipped)
  if (rc) return rc;
  const int B = p->batch, m = p->n_features;
  a.gram = reinterpret_cast<double*>(static_cast<char*>(h->ws) + x_bytes);
  a.colsum = a.gram + (size_t)B * m * m;
  if (m <= GRAM_BLK)
    launch_gram<real>(h, B, a);  // (one block: the column sums of round 2's kernel, bit for bit)
  else  // only the column sums are needed here
    HIPNMF_LAUNCH(colsum_kernel<real>, dim3(B), dim3(256), 0, h->stream, a);
  RandomInitArgs r{};
  r.colsum = a.colsum;
  r.W = W;
  r.H = H;
  r.seed = seed;
  r.T = p->n_samples;
  r.m = m;
  r.k = p->n_components;
  r.w_component_major = p->w_layout == HIPNMF_W_COMPONENT_MAJOR;
  r.first_matrix = first_matrix;
  const long long n = p->n_samples * r.k + (long long)r.k * m;
  const unsigned gx = (unsigned)std::min<long long>((n + 255) / 256, 1024);
  for (int b0 = 0; b0 < B; b0 += 65535) {  // grid.y carries the batch
    const int nb = std::min(65535, B - b0);
    RandomInitArgs rb = r;
    rb.colsum = r.colsum + (size_t)b0 * m;
    rb.W = static_cast<real*>(W) + (size_t)b0 * p->n_samples * r.k;
    rb.H = static_cast<real*>(H) + (size_t)b0 * r.k * m;
    rb.first_matrix = first_matrix + (index ? 0 : b0);
    rb.index = index ? index + b0 : nullptr;
    HIPNMF_LAUNCH(random_init_kernel<real>, dim3(gx, nb), dim3(256), 0, h->stream, rb);
  }
  return finish(h);
}

}  // namespace

template <typename real>
int hipnmf_random_init_indexed(hipnmf_handle* h, const hipnmf_problem* p, uint64_t seed, int first_matrix, const int* index,
                               const real* X, real* W, real* H) {
  return random_init_impl<real>(h, p, seed, first_matrix, X, W, H, index);
}
template int hipnmf_random_init_indexed<float>(hipnmf_handle*, const hipnmf_problem*, uint64_t, int, const int*, const float*, float*, float*);
template int hipnmf_random_init_indexed<double>(hipnmf_handle*, const hipnmf_problem*, uint64_t, int, const int*, const double*, double*,
                                                double*);

extern "C" {
int hipnmf_random_init_f32(hipnmf_handle* h, const hipnmf_problem* p, uint64_t seed, int32_t first_matrix, const float* X, float* W,
                           float* H) {
  return random_init_impl<float>(h, p, seed, first_matrix, X, W, H);
}
int hipnmf_random_init_f64(hipnmf_handle* h, const hipnmf_problem* p, uint64_t seed, int32_t first_matrix, const double* X,
                           double* W, double* H) {
  return random_init_impl<double>(h, p, seed, first_matrix, X, W, H);
}
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
