/*
 * nmf_mu_oracle.c -- plain C restatement of the NMF multiplicative-update solver (double precision).
 *
 * TEST INFRASTRUCTURE ONLY: the checker, never the product.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library (oracle/_build/libnmf_mu_oracle.so); nothing under
 * muscle_synergies_amd/ does.
 *
 * What it restates: the arithmetic the reference reaches at src/muscle_synergies/analysis.py:862-863, i.e.
 * scikit-learn's NMF(solver='mu', beta_loss='frobenius') -- third-party, not vendored under /root/reference
 * (requirements.txt:3 pins scikit-learn>=0.21,<=0.24; the image has 1.7.2).  Lines cited are
 * sklearn/decomposition/_nmf.py of 1.7.2.  Unlike the NumPy oracle (which goes through BLAS like sklearn),
 * every sum here is a plain left-to-right loop, so the result is defined by this file alone; it is pinned
 * against the NumPy oracle and the golden fixtures in tests/test_oracle_c.py.
 *
 * Layouts: X is T x m row-major, W is T x k row-major, H is k x m row-major (sklearn orientation).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

#define NMF_EPSILON 1.1920928955078125e-07 /* np.finfo(np.float32).eps for fp32 and fp64 (_nmf.py:39) */

/* ||X - W H||_F  (_beta_divergence, beta = 2, square_root = True; _nmf.py:120-134) */
double nmf_oracle_frobenius_error(const double* X, const double* W, const double* H, long T, int m, int k,
                                  double* sse_col /* [m] or NULL */) {
  double tot = 0.0;
  if (sse_col)
    for (int j = 0; j < m; ++j) sse_col[j] = 0.0;
  for (long t = 0; t < T; ++t)
    for (int j = 0; j < m; ++j) {
      double rec = 0.0;
      for (int c = 0; c < k; ++c) rec += W[t * k + c] * H[c * m + j];
      const double r = X[t * m + j] - rec;
      tot += r * r;
      if (sse_col) sse_col[j] += r * r;
    }
  return sqrt(tot);
}

/* W *= (X H^T) / (W (H H^T))  (_multiplicative_update_w, _nmf.py:540-554, 615-631) */
static void update_w(const double* X, double* W, const double* H, long T, int m, int k, double l1, double l2,
                     double* HHt) {
  for (int a = 0; a < k; ++a)
    for (int b = 0; b < k; ++b) {
      double s = 0.0;
      for (int j = 0; j < m; ++j) s += H[a * m + j] * H[b * m + j];
      HHt[a * k + b] = s;
    }
  double num[64], den[64];
  for (long t = 0; t < T; ++t) {
    for (int c = 0; c < k; ++c) {
      double n = 0.0, d = 0.0;
      for (int j = 0; j < m; ++j) n += X[t * m + j] * H[c * m + j];
      for (int c2 = 0; c2 < k; ++c2) d += W[t * k + c2] * HHt[c2 * k + c];
      if (l1 > 0) d += l1;
      if (l2 > 0) d = d + l2 * W[t * k + c];
      if (d == 0.0) d = NMF_EPSILON;
      num[c] = n;
      den[c] = d;
    }
    for (int c = 0; c < k; ++c) W[t * k + c] *= num[c] / den[c];
  }
}

/* H *= (W^T X) / ((W^T W) H)  (_multiplicative_update_h, _nmf.py:638-640, 701-728) */
static void update_h(const double* X, const double* W, double* H, long T, int m, int k, double l1, double l2,
                     double* WtX, double* WtW, double* Hnew) {
  for (int i = 0; i < k * m; ++i) WtX[i] = 0.0;
  for (int i = 0; i < k * k; ++i) WtW[i] = 0.0;
  for (long t = 0; t < T; ++t)
    for (int c = 0; c < k; ++c) {
      const double w = W[t * k + c];
      for (int j = 0; j < m; ++j) WtX[c * m + j] += w * X[t * m + j];
      for (int c2 = 0; c2 < k; ++c2) WtW[c * k + c2] += w * W[t * k + c2];
    }
  for (int c = 0; c < k; ++c)
    for (int j = 0; j < m; ++j) {
      double d = 0.0;
      for (int c2 = 0; c2 < k; ++c2) d += WtW[c * k + c2] * H[c2 * m + j];
      if (l1 > 0) d += l1;
      if (l2 > 0) d = d + l2 * H[c * m + j];
      if (d == 0.0) d = NMF_EPSILON;
      Hnew[c * m + j] = H[c * m + j] * (WtX[c * m + j] / d);
    }
  for (int i = 0; i < k * m; ++i) H[i] = Hnew[i];
}

/*
 * _fit_multiplicative_update (_nmf.py:731-893): W first, then H with the new W; when tol > 0 the error is
 * evaluated every `check_every` iterations and the loop stops when (prev - err) / err_init < tol.
 * Returns n_iter (the last executed iteration), or -1 on a bad argument.  W and H are updated in place.
 */
int nmf_oracle_fit(const double* X, double* W, double* H, long T, int m, int k, int max_iter, double tol,
                   int check_every, int update_h_flag, double l1w, double l1h, double l2w, double l2h,
                   double* err_out /* final ||X - WH||_F or NULL */) {
  if (!X || !W || !H || T < 1 || m < 1 || k < 1 || k > 64 || max_iter < 1 || check_every < 1) return -1;
  double* buf = (double*)malloc(sizeof(double) * (size_t)(k * k * 2 + k * m * 2));
  if (!buf) return -1;
  double *HHt = buf, *WtW = buf + k * k, *WtX = WtW + k * k, *Hnew = WtX + k * m;
  double err_init = 0.0, prev = 0.0;
  if (tol > 0) {
    err_init = nmf_oracle_frobenius_error(X, W, H, T, m, k, NULL);
    prev = err_init;
  }
  int n_iter = 0;
  for (int it = 1; it <= max_iter; ++it) {
    n_iter = it;
    update_w(X, W, H, T, m, k, l1w, l2w, HHt);
    if (update_h_flag) update_h(X, W, H, T, m, k, l1h, l2h, WtX, WtW, Hnew);
    if (tol > 0 && it % check_every == 0) {
      const double err = nmf_oracle_frobenius_error(X, W, H, T, m, k, NULL);
      if ((prev - err) / err_init < tol) break;
      prev = err;
    }
  }
  if (err_out) *err_out = nmf_oracle_frobenius_error(X, W, H, T, m, k, NULL);
  free(buf);
  return n_iter;
}
