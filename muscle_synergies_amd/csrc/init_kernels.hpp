// init_kernels.hpp -- T-long passes of the on-device NNDSVD initialisation.
//
// Reference semantics: sklearn _initialize_nmf (sklearn/decomposition/_nmf.py:221-373).  The leading singular
// triplets of X (T x m, T >> m) are taken from the Gram matrix X^T X = V S^2 V^T; u_j = X v_j / s_j.
// One workgroup per matrix; every accumulation is fp64 with a fixed summation order.
#pragma once
#include <hip/hip_runtime.h>

namespace hipnmf {

struct InitArgs {
  const void* X;  // canonical channel-major [B][m][ld]
  long long bstride, ld;
  int T, m, k;
  double* gram;    // [B][m][m]
  double* colsum;  // [B][m]
  const double* V;      // [B][k][m]
  const double* inv_s;  // [B][k]
  double* stats;        // [B][k][4]: sum u+^2, sum u-^2, pivot value (signed, largest |u|), unused
  const double* coef;   // [B][k][2]: scale, sign (+1: positive part of u, -1: negative part, 0: |u|)
  const double* fill;   // [B]
  double eps;
  void* W0;  // [B][T][k]
};

constexpr int GRAM_TILE = 64;  // rows per LDS tile
constexpr int GRAM_BLK = 32;   // channels per block of the Gram matrix: one workgroup computes one GRAM_BLK x GRAM_BLK block
constexpr int NNDSVD_KB = 8;   // components per workgroup of the two projection kernels

// gram[b][j][j2] = sum_t X[t][j] X[t][j2];  colsum[b][j] = sum_t X[t][j].  Any number of channels: grid = (nblk * nblk, B) with
// nblk = ceil(m / GRAM_BLK); block (jb, j2b) of the matrix per workgroup, every entry accumulated by ONE thread in time order
// (fixed summation order whatever the grid), the column sums by the blocks of the first block column.
template <typename real>
__global__ void __launch_bounds__(256) gram_kernel(InitArgs a) {
  __shared__ double xa[GRAM_BLK][GRAM_TILE + 1], xb[GRAM_BLK][GRAM_TILE + 1];
  const int b = blockIdx.y, m = a.m;
  const int nblk = (m + GRAM_BLK - 1) / GRAM_BLK;
  const int jb = (int)blockIdx.x / nblk, j2b = (int)blockIdx.x % nblk;
  const bool diag = jb == j2b;
  const real* __restrict__ Xb = static_cast<const real*>(a.X) + (long long)b * a.bstride;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};  // 4 of the block's 1 024 (j, j2) pairs per thread
  double csum = 0.0;
  for (int t0 = 0; t0 < a.T; t0 += GRAM_TILE) {
    __syncthreads();
    for (int i = threadIdx.x; i < GRAM_BLK * GRAM_TILE; i += blockDim.x) {
      const int jl = i / GRAM_TILE, tt = i % GRAM_TILE, t = t0 + tt;
      const int j = jb * GRAM_BLK + jl, j2 = j2b * GRAM_BLK + jl;
      xa[jl][tt] = (t < a.T && j < m) ? (double)Xb[(long long)j * a.ld + t] : 0.0;
      if (!diag) xb[jl][tt] = (t < a.T && j2 < m) ? (double)Xb[(long long)j2 * a.ld + t] : 0.0;
    }
    __syncthreads();
    const double(*xr)[GRAM_TILE + 1] = diag ? xa : xb;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pidx = threadIdx.x + q * 256;
      const int jl = pidx / GRAM_BLK, j2l = pidx % GRAM_BLK;
      double s = acc[q];
      for (int tt = 0; tt < GRAM_TILE; ++tt) s = fma(xa[jl][tt], xr[j2l][tt], s);
      acc[q] = s;
    }
    if (j2b == 0 && threadIdx.x < GRAM_BLK) {
      double s = csum;
      for (int tt = 0; tt < GRAM_TILE; ++tt) s += xa[threadIdx.x][tt];
      csum = s;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int pidx = threadIdx.x + q * 256;
    const int j = jb * GRAM_BLK + pidx / GRAM_BLK, j2 = j2b * GRAM_BLK + pidx % GRAM_BLK;
    if (j < m && j2 < m) a.gram[(long long)b * m * m + (long long)j * m + j2] = acc[q];
  }
  if (j2b == 0 && threadIdx.x < GRAM_BLK && jb * GRAM_BLK + (int)threadIdx.x < m)
    a.colsum[(long long)b * m + jb * GRAM_BLK + threadIdx.x] = csum;
}

// u_j[t] = (sum_c X[t][c] V[j][c]) * inv_s[j] for the KMAX components of one block; Vs: [KMAX][m] in LDS
template <typename real, int KMAX>
__device__ __forceinline__ void project_row(const real* __restrict__ Xb, long long ld, int t, int m, int k,
                                            const double* Vs, const double* is, double (&u)[KMAX]) {
#pragma unroll
  for (int j = 0; j < KMAX; ++j) u[j] = 0.0;
  for (int c = 0; c < m; ++c) {
    const double x = (double)Xb[(long long)c * ld + t];
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
      if (j < k) u[j] = fma(x, Vs[j * m + c], u[j]);
  }
#pragma unroll
  for (int j = 0; j < KMAX; ++j) u[j] *= is[j < k ? j : 0];
}

// grid = (B, ceil(k / NNDSVD_KB)): workgroup (b, kb) handles components [kb * NNDSVD_KB, ...) of matrix b; dynamic LDS: the block's
// rows of V, NNDSVD_KB * m doubles (any number of channels and components)
template <typename real>
__global__ void __launch_bounds__(256) nndsvd_stats_kernel(InitArgs a) {
  constexpr int KMAX = NNDSVD_KB;
  extern __shared__ __attribute__((aligned(16))) unsigned char init_smem[];
  double* Vs = reinterpret_cast<double*>(init_smem);
  __shared__ double is[KMAX];
  __shared__ double red[4][KMAX][3];
  const int b = blockIdx.x, m = a.m, j0 = (int)blockIdx.y * KMAX;
  const int k = a.k - j0 < KMAX ? a.k - j0 : KMAX;  // components of this block
  const real* __restrict__ Xb = static_cast<const real*>(a.X) + (long long)b * a.bstride;
  for (int i = threadIdx.x; i < k * m; i += blockDim.x) Vs[i] = a.V[((long long)b * a.k + j0) * m + i];
  if ((int)threadIdx.x < k) is[threadIdx.x] = a.inv_s[(long long)b * a.k + j0 + threadIdx.x];
  __syncthreads();
  double sp[KMAX], sn[KMAX], piv[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) sp[j] = sn[j] = piv[j] = 0.0;
  for (int t = threadIdx.x; t < a.T; t += blockDim.x) {
    double u[KMAX];
    project_row<real, KMAX>(Xb, a.ld, t, m, k, Vs, is, u);
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      const double up = u[j] > 0 ? u[j] : 0.0, un = u[j] < 0 ? -u[j] : 0.0;
      sp[j] = fma(up, up, sp[j]);
      sn[j] = fma(un, un, sn[j]);
      if (fabs(u[j]) > fabs(piv[j])) piv[j] = u[j];  // first occurrence wins inside a thread (np.argmax)
    }
  }
  // wave reduction (fixed order), then across the 4 waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      sp[j] += __shfl_xor(sp[j], off, 64);
      sn[j] += __shfl_xor(sn[j], off, 64);
      const double o = __shfl_xor(piv[j], off, 64);
      if (fabs(o) > fabs(piv[j])) piv[j] = o;
    }
    if (lane == 0) {
      red[wave][j][0] = sp[j];
      red[wave][j][1] = sn[j];
      red[wave][j][2] = piv[j];
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < k) {
    const int j = threadIdx.x;
    double p = 0.0, n = 0.0, pv = 0.0;
    for (int w = 0; w < 4; ++w) {
      p += red[w][j][0];
      n += red[w][j][1];
      if (fabs(red[w][j][2]) > fabs(pv)) pv = red[w][j][2];
    }
    double* o = a.stats + ((long long)b * a.k + j0 + j) * 4;
    o[0] = p;
    o[1] = n;
    o[2] = pv;
    o[3] = 0.0;
  }
}

template <typename real>
__global__ void __launch_bounds__(256) nndsvd_write_kernel(InitArgs a) {
  constexpr int KMAX = NNDSVD_KB;
  extern __shared__ __attribute__((aligned(16))) unsigned char init_smem[];
  double* Vs = reinterpret_cast<double*>(init_smem);
  __shared__ double is[KMAX], cf[KMAX][2];
  const int b = blockIdx.x, m = a.m, j0 = (int)blockIdx.y * KMAX;
  const int k = a.k - j0 < KMAX ? a.k - j0 : KMAX;
  const real* __restrict__ Xb = static_cast<const real*>(a.X) + (long long)b * a.bstride;
  real* __restrict__ Wb = static_cast<real*>(a.W0) + (long long)b * a.T * a.k;
  for (int i = threadIdx.x; i < k * m; i += blockDim.x) Vs[i] = a.V[((long long)b * a.k + j0) * m + i];
  if ((int)threadIdx.x < k) {
    is[threadIdx.x] = a.inv_s[(long long)b * a.k + j0 + threadIdx.x];
    cf[threadIdx.x][0] = a.coef[((long long)b * a.k + j0 + threadIdx.x) * 2];
    cf[threadIdx.x][1] = a.coef[((long long)b * a.k + j0 + threadIdx.x) * 2 + 1];
  }
  __syncthreads();
  const double fill = a.fill[b];
  for (int t = threadIdx.x; t < a.T; t += blockDim.x) {
    double u[KMAX];
    project_row<real, KMAX>(Xb, a.ld, t, m, k, Vs, is, u);
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
      if (j < k) {
        const double part = cf[j][1] > 0 ? (u[j] > 0 ? u[j] : 0.0) : (cf[j][1] < 0 ? (u[j] < 0 ? -u[j] : 0.0) : fabs(u[j]));
        real w = (real)(cf[j][0] * part);
        if ((double)w < a.eps) w = (real)0;      // W[W < eps] = 0            (_nmf.py:355)
        if (w == (real)0) w = (real)fill;        // nndsvda: zeros -> X.mean() (_nmf.py:360-362); fill = 0 for nndsvd
        Wb[(long long)t * a.k + j0 + j] = w;
      }
  }
}


// -------------------------------------------------------------------------------------------------
// init='random' of sklearn (_nmf.py:303-314) on the device: avg = sqrt(X.mean() / k), H = avg |N(0,1)|, W = avg |N(0,1)|.
// Counter-based (Philox4x32-10 keyed by the seed; the counter is the LOGICAL element index, so the values do not
// depend on the layout, the launch geometry or the batch split over GPUs) with Box-Muller; stream 0 = H, 1 = W.
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
// |N(0,1)| number `idx` of stream (matrix, which) under `seed`: one Philox block yields 4 normals, idx picks its own
__device__ __forceinline__ double abs_normal(unsigned long long seed, unsigned matrix, unsigned which, unsigned long long idx) {
  unsigned c[4] = {(unsigned)(idx >> 2), (unsigned)((idx >> 2) >> 32), matrix, which};
  philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
  const int pair = (int)(idx & 2);  // elements 0,1 of a block share (c0, c1), elements 2,3 share (c2, c3)
  const double u1 = ((double)c[pair] + 0.5) * (1.0 / 4294967296.0), u2 = ((double)c[pair + 1] + 0.5) * (1.0 / 4294967296.0);
  const double r = sqrt(-2.0 * log(u1)), a = 6.283185307179586476925 * u2;
  return fabs((idx & 1) ? r * sin(a) : r * cos(a));
}

struct RandomInitArgs {
  const double* colsum;  // [B][m] (gram_kernel)
  void* W;               // per w_layout: [B][T][k] or [B][k][T]
  void* H;               // [B][k][m]
  unsigned long long seed;
  long long T;
  int m, k, w_component_major, first_matrix;  // first_matrix: global index of matrix 0 (batches scattered over GPUs)
  const int* index;      // [B] or nullptr: matrix b of this (compacted) batch is matrix index[b] of the original one
};

// colsum[b][j] = sum_t X[t][j] for any number of channels (without the m x m products of the Gram kernel above): one workgroup per
// matrix, channel-major X, fp64 sums in a fixed order
template <typename real>
__global__ void __launch_bounds__(256) colsum_kernel(InitArgs a) {
  __shared__ double part[256];
  const int b = blockIdx.x;
  const real* __restrict__ Xb = static_cast<const real*>(a.X) + (long long)b * a.bstride;
  for (int j = 0; j < a.m; ++j) {
    double s = 0.0;
    for (int t = threadIdx.x; t < a.T; t += 256) s += (double)Xb[(long long)j * a.ld + t];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) a.colsum[(long long)b * a.m + j] = part[0];
    __syncthreads();
  }
}

template <typename real>
__global__ void __launch_bounds__(256) random_init_kernel(RandomInitArgs a) {
  const int b = blockIdx.y;
  double tot = 0.0;
  for (int j = 0; j < a.m; ++j) tot += a.colsum[(long long)b * a.m + j];
  const double avg = sqrt(tot / ((double)a.T * (double)a.m) / (double)a.k);
  const unsigned matrix = (unsigned)(a.first_matrix + (a.index ? a.index[b] : b));
  real* W = static_cast<real*>(a.W) + (long long)b * a.T * a.k;
  real* H = static_cast<real*>(a.H) + (long long)b * a.k * a.m;
  const long long nw = a.T * a.k, nh = (long long)a.k * a.m;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nw + nh; i += (long long)gridDim.x * blockDim.x) {
    if (i < nh) {
      H[i] = (real)(avg * abs_normal(a.seed, matrix, 0u, (unsigned long long)i));  // logical index c * m + j
    } else {
      const long long e = i - nh;  // logical index t * k + c
      const long long t = e / a.k;
      const int c = (int)(e % a.k);
      const real v = (real)(avg * abs_normal(a.seed, matrix, 1u, (unsigned long long)e));
      W[a.w_component_major ? (long long)c * a.T + t : e] = v;
    }
  }
}

}  // namespace hipnmf
