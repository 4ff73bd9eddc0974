// nmf_big.hpp -- the general-shape path of the solver (round 4): every (n_features, n_components) beyond the instances of
// nmf_wide.hpp -- up to 512 channels (HD-EMG grids of 256 electrodes are ordinary) and 64 components, fp32 and fp64, and float64
// with more than 16 components on more than 64 channels -- so that no solver='mu' call of the reference leaves the GPU
// (src/muscle_synergies/analysis.py:829-846 accepts any 1 <= n <= max <= n_muscles; :862-863 is the seam).
//
// Arithmetic replaced: sklearn/decomposition/_nmf.py (1.7.2) _multiplicative_update_w (:540-554, 615-631),
// _multiplicative_update_h (:638-640, 701-728), _beta_divergence (:85-134), loop + stop rule (:731-893); beta_loss = 'frobenius'.
//
// Shapes are run-time values here except the padded component count KP (16 / 32 / 48 / 64: the accumulators live in registers).
// One iteration is four launches over row slices of every matrix (grid.x = slice, grid.y = matrix), replayed as a hipGraph:
//   big_hht_kernel      H H^T (KP x KP, zero padded)
//   big_pass_w_kernel   per 16-row subtile of a wave: numerator^T = H X^T accumulated over 16-channel blocks (the channel-blocked
//                       loop: X is streamed once, H staged in LDS in blocks of CBH channels), denominator^T = (H H^T) W^T,
//                       W <- W * num / den in the accumulator layout, one 16-byte store per lane and component block
//   big_records_kernel  W^T X (KP x CB channels per workgroup, grid.z over channel blocks) and W^T W (the last grid.z) of the slice:
//                       the updated rows of W and the rows of X staged per wave in LDS and read back transposed (the contraction
//                       runs over rows here); the four waves' accumulators summed in wave order -> the slice's record
//   big_hupdate_kernel  records summed in slice order, H <- H * (W^T X) / ((W^T W) H), 64 channels per workgroup
// and the residual (stop rule every check_every iterations, reconstruction_err_, per-column SSE for VAF) is
//   big_resid_kernel    R = X - W H per 16-channel block on the pipe, per-column sums reduced over the rows of the subtile by
//                       cross-lane adds, accumulated per wave in LDS, summed in wave order -> the slice's column record
//   big_resid_finalize_kernel   (= wide_resid_finalize_kernel with the column buffer sized at run time)
// All four contractions use v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 with the operand conventions of nmf_wide.hpp
// (A: lane (i, g) <-> A[i][g], B: lane (j, g) <-> B[g][j], D: lane (j, g), register r <-> D[4 g + r][j] once the rows of A are
// permuted by WideMma::arow).  X is read twice per iteration (update pass, record pass) and W written once and read twice:
// (8 m + 12 k) sizeof bytes per row against the 4 m + 8 k of the one-pass kernels -- the price of accumulators that no longer
// fit one wave; with k >= 16 the arithmetic intensity (~k flop / byte) keeps the pipe, not the stream, the bound.
// Every sum has a fixed order (slices in order, waves in order, lanes by a fixed butterfly): results are bitwise reproducible.
#pragma once
#include "nmf_wide.hpp"

namespace hipnmf {

template <typename real>
struct BigArgs {
  const real* X;  // row-major [T][ldx], rows 16-byte aligned
  long long x_bstride, ldx;
  real* W;  // row-major [T][KP] (components >= k are zero and stay zero)
  long long w_bstride;
  real* H;            // [B][k][m]
  real* HHt;          // [B][KP][KP]
  real* part;         // [B][S][REC]  REC = KP MP + KP KP: [W^T X | W^T W] per slice
  real* colpart;      // [B][S][2 MP] sse | xsq per slice
  const real* state;  // [B][8]: entry 3 != 0 = matrix converged (its workgroups return at once), or nullptr
  int T, m, k, KP, MP, xchunks;  // xchunks: 16-byte pieces of a row of X that hold data
  int S, rows_per_slice;
  int CBH;  // channels of H staged in LDS at a time (a multiple of 16; MP when all of H fits)
  real l1w, l2w;
  int kl;   // 1: Kullback-Leibler updates (_nmf.py:556-591, 642-684) -- see the KL notes in every kernel; column records are 3 MP then
  int update_h;  // 0: H stays fixed (NMF.transform): the one-pass kernel (nmf_big1.hpp) skips the record
  const real* hht_part;  // one-pass kernel: H H^T as partial products per 64-channel block, [B][n_hblk][KP][KP] (big_hupdate_kernel /
  int n_hblk;            // big_hht_part_kernel write them, the kernel adds them in block order): no H H^T launch per iteration
};
// the one-pass kernel of nmf_big1.hpp (instances: inst_big1_f32.hip): KP components padded, 8 waves x 16 NQ channels, RS subtiles per round
template <typename real>
struct Big1Kernel {
  void (*fn)(BigArgs<real>);
  size_t smem;
  int KP, NQ, RS;
  const char* name;
  void (*resid)(BigArgs<real>);  // big1_resid_kernel<real, KP, NQ>: the residual on the same decomposition (512 threads, no LDS)
  void (*fn_kl)(BigArgs<real>);  // the Kullback-Leibler flavour (LOSS = 1) or nullptr; its LDS and name:
  size_t smem_kl;
  const char* name_kl;
};
const Big1Kernel<float>* big1_kernel_f32(int KP, int MP);    // nullptr: no instance covers MP channels
const Big1Kernel<double>* big1_kernel_f64(int KP, int MP);  // (inst_big1_f64.hip: 16 / 32 padded components, up to 256 channels)
// values per slice of the column record: sse | xsq (| the Kullback-Leibler divergence per column)
template <typename real>
__device__ __forceinline__ int big_ncol(const BigArgs<real>& a) { return (a.kl ? 3 : 2) * a.MP; }
// x / wh of the Kullback-Leibler updates, wh >= EPSILON (float: the bare reciprocal, as in the other KL kernels)
__device__ __forceinline__ float big_quot(float x, float wh) { return kl_quot(x, wh); }
__device__ __forceinline__ double big_quot(double x, double wh) { return x / wh; }

constexpr int BIG_CB = 64;  // channels per workgroup of the record kernel

// 16 bytes = VEC consecutive elements of a row, or zeros
template <typename real>
__device__ __forceinline__ void big_load4(const real* __restrict__ row, int col, bool ok, real (&out)[4]) {
  if (ok) {
    __builtin_memcpy(out, __builtin_assume_aligned(row + col, 16), 4 * sizeof(real));
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = (real)0;
  }
}

// H H^T, zero padded to KP x KP.  grid (B, KP): one workgroup per ROW of the result; the 256 threads split the channels into
// 256 / KP contiguous segments per entry, each summed in order, the segments added in a fixed tree (the first version -- one
// workgroup per matrix, a thread per entry over all channels -- took 0.24 ms of the 1.13 ms iteration at 512 x 32).
// Kullback-Leibler: column 0 holds rowsum(H), the W update's denominator (_nmf.py:577-581); the rest stays 0.
template <typename real>
__global__ void __launch_bounds__(256) big_hht_kernel(BigArgs<real> a) {
  __shared__ real partial[256];
  const int b = blockIdx.x, c = blockIdx.y, KP = a.KP;
  if (a.state && a.state[(long long)b * 8 + 3] != (real)0) return;
  const real* __restrict__ Hb = a.H + (long long)b * a.k * a.m;
  real* __restrict__ out = a.HHt + (long long)b * KP * KP + (long long)c * KP;
  const int nseg = 256 / KP;  // 16 / 8 / 5 / 4 segments for KP = 16 / 32 / 48 / 64
  const int c2 = threadIdx.x % KP, seg = threadIdx.x / KP;
  real s = (real)0;
  if (seg < nseg && c < a.k) {
    const int len = (a.m + nseg - 1) / nseg, j0 = seg * len, j1 = (j0 + len < a.m) ? j0 + len : a.m;
    const real* __restrict__ h1 = Hb + (long long)c * a.m;
    if (a.kl) {
      if (c2 == 0)
        for (int jj = j0; jj < j1; ++jj) s += h1[jj];
    } else if (c2 < a.k) {
      const real* __restrict__ h2 = Hb + (long long)c2 * a.m;
      for (int jj = j0; jj < j1; ++jj) s = fma_(h1[jj], h2[jj], s);
    }
  }
  partial[threadIdx.x] = s;
  __syncthreads();
  if ((int)threadIdx.x < KP) {
    real tot = partial[threadIdx.x];
    for (int g = 1; g < nseg; ++g) tot += partial[g * KP + threadIdx.x];  // (segments in order)
    out[threadIdx.x] = tot;
  }
}

// rows [row_begin, row_end) of matrix b handled by slice blockIdx.x
template <typename real>
__device__ __forceinline__ void big_slice(const BigArgs<real>& a, int& row_begin, int& row_end) {
  row_begin = (int)blockIdx.x * a.rows_per_slice;
  row_end = row_begin + a.rows_per_slice;
  if (row_end > a.T) row_end = a.T;
}

// stages H[:, cb0 .. cb0 + CBH) (zero beyond k x m) as sH[KP][CBH + 4]
template <typename real>
__device__ __forceinline__ void big_stage_h(const BigArgs<real>& a, const real* __restrict__ Hb, int cb0, real* __restrict__ sH) {
  const int SH = a.CBH + 4;
  for (int idx = threadIdx.x; idx < a.KP * a.CBH; idx += blockDim.x) {
    const int c = idx / a.CBH, jj = idx % a.CBH;
    sH[c * SH + jj] = (c < a.k && cb0 + jj < a.m) ? Hb[(long long)c * a.m + cb0 + jj] : (real)0;
  }
}

// W <- W * (X H^T) / (W (H H^T))  (_nmf.py:540-554, 615-631) for the rows of one slice.  grid (S, B), 256 threads; dynamic LDS:
// sH [KP][CBH + 4] + sHHt [KP][KP + 4].
template <typename real, int KP>
__global__ void __launch_bounds__(256) big_pass_w_kernel(BigArgs<real> a) {
  using M = WideMma<real>;
  using acc = typename M::acc;
  constexpr int NKB = KP / 16, SK = KP + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
  const int b = blockIdx.y;
  if (a.state && a.state[(long long)b * 8 + 3] != (real)0) return;
  const int SH = a.CBH + 4;
  real* const sH = reinterpret_cast<real*>(big_smem);
  real* const sHHt = sH + KP * SH;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
  const int ar = M::arow(j);
  const real* __restrict__ Xb = a.X + (long long)b * a.x_bstride;
  real* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  const real* __restrict__ Hb = a.H + (long long)b * a.k * a.m;
  int row_begin, row_end;
  big_slice(a, row_begin, row_end);
  for (int idx = tid; idx < KP * KP; idx += 256) sHHt[(idx / KP) * SK + idx % KP] = a.HHt[(long long)b * KP * KP + idx];
  const bool h_resident = a.CBH >= a.MP;
  if (h_resident) big_stage_h(a, Hb, 0, sH);
  __syncthreads();
  constexpr int V = 16 / (int)sizeof(real);  // elements per 16-byte piece
  for (int t0 = row_begin + 16 * wave; t0 - 16 * wave < row_end; t0 += 64) {  // (all four waves make the same number of trips)
    const int row = t0 + j;
    const bool rok = row < row_end;
    const real* __restrict__ xrow = Xb + (long long)row * a.ldx;
    // W fragment: lane (row j, g) <-> components 16 kb + 4 g .. + 3; register s is the B operand of k-step s
    // (requested first: in flight while the numerator runs)
    real w[NKB][4];
    real* __restrict__ wrow = Wb + (long long)row * KP;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (rok) {
        __builtin_memcpy(w[kb], __builtin_assume_aligned(wrow + 16 * kb + 4 * g, 16), 4 * sizeof(real));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) w[kb][r] = (real)0;
      }
    }
    acc num[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) num[kb] = acc{0, 0, 0, 0};
    for (int cb0 = 0; cb0 < a.MP; cb0 += a.CBH) {
      if (!h_resident) {
        __syncthreads();
        big_stage_h(a, Hb, cb0, sH);
        __syncthreads();
      }
      const int cend = (cb0 + a.CBH < a.MP) ? cb0 + a.CBH : a.MP;
      for (int ch0 = cb0; ch0 < cend; ch0 += 64) {  // four 16-channel blocks per trip: their loads go out together
        real x[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = ch0 + 16 * q + 4 * g;
          const bool cok = rok && ch0 + 16 * q < cend;
          if constexpr (V == 4) {
            big_load4<real>(xrow, col, cok && (col >> 2) < a.xchunks, x[q]);
          } else {  // fp64: two 16-byte pieces
            real lo[4] = {0, 0, 0, 0};
            if (cok && (col >> 1) < a.xchunks) __builtin_memcpy(lo, __builtin_assume_aligned(xrow + col, 16), 16);
            if (cok && ((col + 2) >> 1) < a.xchunks) __builtin_memcpy(lo + 2, __builtin_assume_aligned(xrow + col + 2, 16), 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) x[q][r] = lo[r];
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (ch0 + 16 * q < cend) {
            if (a.kl) {
              // Kullback-Leibler: the block of (W H)^T on the pipe as in big_resid_kernel -- A = H^T (channels x components), B = the
              // W fragment, D: lane (row j, g), register r <-> channel ch + 4 g + r: exactly where the lane's X values sit, so
              // Q = X / max(W H, eps) replaces them and the numerator below is (X / WH) H^T
              acc rec = acc{0, 0, 0, 0};
#pragma unroll
              for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int s = 0; s < 4; ++s) rec = M::mma(sH[(16 * kb + 4 * g + s) * SH + (ch0 + 16 * q - cb0) + ar], w[kb][s], rec);
#pragma unroll
              for (int r = 0; r < 4; ++r) x[q][r] = big_quot(x[q][r], kl_floor(rec[r]));
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
              real ha[4];
              wide_lds_read<real, 4>(sH + (16 * kb + ar) * SH + (ch0 + 16 * q - cb0) + 4 * g, ha);
#pragma unroll
              for (int s = 0; s < 4; ++s) num[kb] = M::mma(ha[s], x[q][s], num[kb]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      acc den = acc{0, 0, 0, 0};
      if (a.kl) {  // rowsum(H) of the lane's components (big_hht_kernel left it in column 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) den[r] = sHHt[(16 * kb + 4 * g + r) * SK];
      } else {
#pragma unroll
        for (int kb2 = 0; kb2 < NKB; ++kb2) {
          real ha[4];
          wide_lds_read<real, 4>(sHHt + (16 * kb + ar) * SK + 16 * kb2 + 4 * g, ha);
#pragma unroll
          for (int s = 0; s < 4; ++s) den = M::mma(ha[s], w[kb2][s], den);
        }
      }
      real wn[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        real d = den[r];
        if (a.l1w > (real)0) d = d + a.l1w;
        if (a.l2w > (real)0) d = d + a.l2w * w[kb][r];
        d = (d == (real)0) ? eps_val<real>() : d;
        wn[r] = w[kb][r] * (num[kb][r] / d);
      }
      if (rok) __builtin_memcpy(__builtin_assume_aligned(wrow + 16 * kb + 4 * g, 16), wn, 4 * sizeof(real));
    }
  }
}

// Record of a slice: W^T X for the CB channels [z CB, ...) (z < nz - 1) or W^T W (z = nz - 1).  grid (S, B, nz), 256 threads;
// dynamic LDS: per wave a W stage [16][KP + 4] and a B-source stage [16][CB + 4], then (reused) the reduction buffer.
template <typename real, int KP>
__global__ void __launch_bounds__(256) big_records_kernel(BigArgs<real> a) {
  using M = WideMma<real>;
  using acc = typename M::acc;
  constexpr int NKB = KP / 16, CB = BIG_CB, NCB = CB / 16, SWs = KP + 4, SXs = CB + 4;
  constexpr int PERWAVE = 16 * SWs + 16 * SXs;
  extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
  const int b = blockIdx.y;
  if (a.state && a.state[(long long)b * 8 + 3] != (real)0) return;
  real* const smem = reinterpret_cast<real*>(big_smem);
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
  const int ar = M::arow(j);
  real* const wst = smem + wave * PERWAVE;
  real* const xst = wst + 16 * SWs;
  const bool is_wtw = (int)blockIdx.z == (int)gridDim.z - 1;
  // Kullback-Leibler: the channel blocks accumulate W'^T Q' with Q' = X / max(W' H, eps) (_nmf.py:660-663) -- the block of H sits
  // in LDS behind the stages, W' H comes from the pipe in the layout of the staged X values (the
  // contraction's k-step s then covers rows 4 g + s instead of 4 s + g: the same 16 rows) -- and the last grid.z leaves
  // colsum(W') in column 0 of the W^T W block (its B source is a column of ones).
  const bool klq = a.kl != 0 && !is_wtw;
  constexpr int SHB = CB + 4;
  real* const sHb = smem + 4 * PERWAVE;  // [KP][CB + 4] behind the stages (the reduction buffer at the end may overlay it: dead by then)
  if (klq) {
    const real* __restrict__ Hb = a.H + (long long)b * a.k * a.m;
    for (int idx = tid; idx < KP * CB; idx += 256) {
      const int c = idx / CB, jj = idx % CB, ch = (int)blockIdx.z * CB + jj;
      sHb[c * SHB + jj] = (c < a.k && ch < a.m) ? Hb[(long long)c * a.m + ch] : (real)0;
    }
    __syncthreads();
  }
  const real* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  // the B operand's source: a block of CB channels of X, or W itself (W^T W)
  const real* __restrict__ src = is_wtw ? Wb : a.X + (long long)b * a.x_bstride;
  const long long src_ld = is_wtw ? (long long)KP : a.ldx;
  const int src_col0 = is_wtw ? 0 : (int)blockIdx.z * CB;
  const int src_pieces = is_wtw ? KP / (16 / (int)sizeof(real)) : a.xchunks;  // valid 16-byte pieces per source row
  constexpr int V = 16 / (int)sizeof(real);
  int row_begin, row_end;
  big_slice(a, row_begin, row_end);
  acc accu[NKB][NCB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) accu[kb][cb] = acc{0, 0, 0, 0};
  for (int t0 = row_begin + 16 * wave; t0 < row_end; t0 += 64) {
    const int row = t0 + j;
    const bool rok = row < row_end;
    // stage: lane (row j, g) brings 16-byte pieces of its row of W and of the source block
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      real w4[4];
      big_load4<real>(Wb + (long long)row * KP, 16 * kb + 4 * g, rok, w4);  // (fp64: 32 bytes, two aligned halves)
      wide_lds_write<real, 4>(wst + j * SWs + 16 * kb + 4 * g, w4);
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int col = src_col0 + 16 * cb + 4 * g;
      real x4[4] = {0, 0, 0, 0};
      if (a.kl && is_wtw) {  // a column of ones: (W^T B)[c][0] = colsum(W)[c]
        x4[0] = (rok && col == 0) ? (real)1 : (real)0;
      } else if constexpr (V == 4) {
        big_load4<real>(src + (long long)row * src_ld, col, rok && (col >> 2) < src_pieces, x4);
      } else {
        const real* p = src + (long long)row * src_ld + col;
        if (rok && (col >> 1) < src_pieces) __builtin_memcpy(x4, __builtin_assume_aligned(p, 16), 16);
        if (rok && ((col + 2) >> 1) < src_pieces) __builtin_memcpy(x4 + 2, __builtin_assume_aligned(p + 2, 16), 16);
      }
      wide_lds_write<real, 4>(xst + j * SXs + 16 * cb + 4 * g, x4);
    }
    wide_wave_lds_fence();
    // contraction over the 16 rows: k-step s covers rows 4 s + g' (g' = the lane's g)
    real av[NKB][4], bv[NCB][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int rs = klq ? 4 * g + s : 4 * s + g;  // the row k-step s covers
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) av[kb][s] = wst[rs * SWs + 16 * kb + ar];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) bv[cb][s] = xst[rs * SXs + 16 * cb + j];
    }
    if (klq) {
      // (W' H) block: A = W' (rows x components, row arow(j) of the stage), B = H (components x channels);
      // D: lane (channel j, g), register r <-> row 4 g + r -- the rows of bv
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        acc rec = acc{0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          real wa[4];
          wide_lds_read<real, 4>(wst + ar * SWs + 16 * kb + 4 * g, wa);
#pragma unroll
          for (int s = 0; s < 4; ++s) rec = M::mma(wa[s], sHb[(16 * kb + 4 * g + s) * SHB + 16 * cb + j], rec);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) bv[cb][s] = big_quot(bv[cb][s], kl_floor(rec[s]));
      }
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int s = 0; s < 4; ++s) accu[kb][cb] = M::mma(av[kb][s], bv[cb][s], accu[kb][cb]);
    wide_wave_lds_fence();
  }
  // the four waves' tiles summed in wave order; D: lane (j, g), register r <-> [component 16 kb + 4 g + r][channel 16 cb + j]
  __syncthreads();
  real* const red = smem;  // [4][KP][CB]
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * KP + 16 * kb + 4 * g + r) * CB + 16 * cb + j] = accu[kb][cb][r];
  __syncthreads();
  const int rec = KP * a.MP + KP * KP;
  real* __restrict__ out = a.part + ((long long)b * a.S + blockIdx.x) * rec;
  const int ncols = is_wtw ? KP : (a.MP - src_col0 < CB ? a.MP - src_col0 : CB);
  for (int idx = tid; idx < KP * CB; idx += 256) {
    const int c = idx / CB, jj = idx % CB;
    if (jj < ncols) {
      const real s = ((red[idx] + red[KP * CB + idx]) + red[2 * KP * CB + idx]) + red[3 * KP * CB + idx];
      if (is_wtw)
        out[KP * a.MP + c * KP + jj] = s;
      else
        out[c * a.MP + src_col0 + jj] = s;
    }
  }
}

// H <- H * (W^T X) / ((W^T W) H)  (_nmf.py:638-640, 701-728): records summed in slice order.  grid (B, ceil(m / 64)), 256 threads;
// dynamic LDS: sB [k][k] + sNum [k][64] + sH [k][64].
template <typename real>
struct BigHArgs {
  real* H;
  const real* part;   // [B][S][rec]: per record W^T X as [k][ldA] at offset 0 and W^T W as [k][ldB] at offset offB
  const real* state;
  int m, k, S;
  int rec, ldA, offB, ldB;  // slice records of big_records_kernel: KP MP + KP KP, MP, KP MP, KP; packed sums: k m + k k, m, k m, k
  real l1h, l2h;
  int kl;  // 1: H <- H * (W^T Q) / colsum(W)  (_nmf.py:663-684; colsum in column 0 of the second block), H[H < eps64] = 0 (:866-868)
  real* hht_part;  // or nullptr: [B][gridDim.y][KP][KP], this workgroup's 64 channels of the NEW H times themselves (zero padded)
  int KP;
};
// partial H H^T of 64 channels held in LDS as sH [k][64] (columns >= ncol are ignored): out [KP][KP], zero beyond k
template <typename real>
__device__ __forceinline__ void big_hht_partial(const real* __restrict__ sHn, int k, int ncol, int KP, real* __restrict__ out, int kl) {
  for (int idx = threadIdx.x; idx < KP * KP; idx += blockDim.x) {
    const int c = idx / KP, c2 = idx % KP;
    real s = (real)0;
    if (kl) {  // Kullback-Leibler: column 0 holds the block's share of rowsum(H) (_nmf.py:577-581), the rest stays 0
      if (c < k && c2 == 0)
        for (int jj = 0; jj < ncol; ++jj) s += sHn[c * 64 + jj];
    } else if (c < k && c2 < k) {
      // the lanes of a wave differ in c2: each starts c2 channels further on (rows are 64 values apart: without the skew all
      // lanes would sit on one bank); a fixed order per entry all the same
      int jj = c2 % ncol;
      for (int n = 0; n < ncol; ++n) {
        s = fma_(sHn[c * 64 + jj], sHn[c2 * 64 + jj], s);
        jj = (jj + 1 == ncol) ? 0 : jj + 1;
      }
    }
    out[idx] = s;
  }
}
// the same partial products straight from H (before the first iteration; transform): grid (B, ceil(m / 64)), 256 threads,
// dynamic LDS k * 64 values
template <typename real>
__global__ void __launch_bounds__(256) big_hht_part_kernel(BigHArgs<real> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
  real* const sHn = reinterpret_cast<real*>(big_smem);
  const int b = blockIdx.x, c0 = (int)blockIdx.y * 64, k = a.k;
  const int ncol = (a.m - c0 < 64) ? a.m - c0 : 64;
  const real* __restrict__ Hb = a.H + (long long)b * k * a.m;
  for (int idx = threadIdx.x; idx < k * ncol; idx += 256) sHn[(idx / ncol) * 64 + idx % ncol] = Hb[(long long)(idx / ncol) * a.m + c0 + idx % ncol];
  __syncthreads();
  big_hht_partial<real>(sHn, k, ncol, a.KP, a.hht_part + ((long long)b * gridDim.y + blockIdx.y) * a.KP * a.KP, a.kl);
}
template <typename real>
__global__ void __launch_bounds__(256) big_hupdate_kernel(BigHArgs<real> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
  const int b = blockIdx.x, tid = threadIdx.x, c0 = (int)blockIdx.y * 64;
  if (a.state && a.state[(long long)b * 8 + 3] != (real)0) return;
  const int k = a.k, ncol = (a.m - c0 < 64) ? a.m - c0 : 64;
  real* const sB = reinterpret_cast<real*>(big_smem);  // [k][k]
  real* const sNum = sB + k * k;                       // [k][64]
  real* const sHo = sNum + k * 64;                     // [k][64] the old H
  const int rec = a.rec;
  const real* __restrict__ pb = a.part + (long long)b * a.S * rec;
  real* __restrict__ Hb = a.H + (long long)b * k * a.m;
  // records summed in slice order; four loads in flight per value (the additions keep the order)
  auto slice_sum = [&](int off) __attribute__((always_inline)) -> real {
    real s = (real)0;
    int q = 0;
    for (; q + 4 <= a.S; q += 4) {
      const real v0 = pb[(long long)q * rec + off], v1 = pb[(long long)(q + 1) * rec + off], v2 = pb[(long long)(q + 2) * rec + off],
                 v3 = pb[(long long)(q + 3) * rec + off];
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; q < a.S; ++q) s += pb[(long long)q * rec + off];
    return s;
  };
  for (int idx = tid; idx < k * k; idx += 256) sB[idx] = slice_sum(a.offB + (idx / k) * a.ldB + idx % k);
  for (int idx = tid; idx < k * ncol; idx += 256) {
    const int c = idx / ncol, jj = idx % ncol;
    sNum[c * 64 + jj] = slice_sum(c * a.ldA + c0 + jj);
    sHo[c * 64 + jj] = Hb[(long long)c * a.m + c0 + jj];
  }
  __syncthreads();
  for (int idx = tid; idx < k * ncol; idx += 256) {
    const int c = idx / ncol, jj = idx % ncol;
    real d;
    if (a.kl) {
      d = sB[c * k];
      if (d == (real)0) d = (real)1;
    } else {
      d = sB[c * k] * sHo[jj];
      for (int c2 = 1; c2 < k; ++c2) d = fma_(sB[c * k + c2], sHo[c2 * 64 + jj], d);
    }
    const real hold = sHo[c * 64 + jj];
    if (a.l1h > (real)0) d = d + a.l1h;
    if (a.l2h > (real)0) d = d + a.l2h * hold;
    d = (d == (real)0) ? eps_val<real>() : d;
    real hn = hold * (sNum[c * 64 + jj] / d);
    if (a.kl && hn < (real)2.220446049250313e-16) hn = (real)0;
    Hb[(long long)c * a.m + c0 + jj] = hn;
    if (a.hht_part) sNum[c * 64 + jj] = hn;  // (each thread overwrites only the numerators it has consumed)
  }
  if (a.hht_part) {  // the next iteration's H H^T, this block's share (the one-pass kernel adds the blocks in order)
    __syncthreads();
    big_hht_partial<real>(sNum, k, ncol, a.KP, a.hht_part + ((long long)b * gridDim.y + blockIdx.y) * a.KP * a.KP, a.kl);
  }
}

// Time-shard building blocks (hipnmf_shard_* for wide shapes): the slice records summed in slice order and packed as the
// all-reduce wants them -- sums[b] = [W^T X (k x m) | W^T W (k x k)] -- and the slices' column records summed into sse | xsq.
template <typename real>
__global__ void __launch_bounds__(256) big_pack_sums_kernel(BigArgs<real> a, real* __restrict__ sums) {
  const int b = blockIdx.x, k = a.k, m = a.m;
  const int rec = a.KP * a.MP + a.KP * a.KP;
  const real* __restrict__ pb = a.part + (long long)b * a.S * rec;
  real* __restrict__ out = sums + (long long)b * (k * m + k * k);
  for (int idx = threadIdx.x; idx < k * m + k * k; idx += blockDim.x) {
    const int off = idx < k * m ? (idx / m) * a.MP + idx % m : a.KP * a.MP + ((idx - k * m) / k) * a.KP + (idx - k * m) % k;
    real s = (real)0;
    for (int q = 0; q < a.S; ++q) s += pb[(long long)q * rec + off];
    out[idx] = s;
  }
}
template <typename real>
__global__ void __launch_bounds__(256) big_colsum_kernel(BigArgs<real> a, real* __restrict__ sse_col, real* __restrict__ xsq_col) {
  const int b = blockIdx.x;
  const int nc = big_ncol(a);
  const real* __restrict__ cb = a.colpart + (long long)b * a.S * nc;
  for (int jj = threadIdx.x; jj < a.m; jj += blockDim.x) {
    real s0 = (real)0, s1 = (real)0;
    for (int q = 0; q < a.S; ++q) {
      s0 += cb[(long long)q * nc + jj];
      s1 += cb[(long long)q * nc + a.MP + jj];
    }
    if (a.kl) {  // Kullback-Leibler: the divergence per column takes the place of the squared error
      s0 = (real)0;
      for (int q = 0; q < a.S; ++q) s0 += cb[(long long)q * nc + 2 * a.MP + jj];
    }
    sse_col[(long long)b * a.m + jj] = s0;
    if (xsq_col) xsq_col[(long long)b * a.m + jj] = s1;
  }
}

// per-column sum((X - W H)^2) | sum(X^2) of one slice.  grid (S, B), 256 threads; dynamic LDS: sH [KP][CBH + 4] + per-wave
// column accumulators [4][2 MP].
template <typename real, int KP>
__global__ void __launch_bounds__(256) big_resid_kernel(BigArgs<real> a) {
  using M = WideMma<real>;
  using acc = typename M::acc;
  constexpr int NKB = KP / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
  const int b = blockIdx.y;
  if (a.state && a.state[(long long)b * 8 + 3] != (real)0) return;
  const int SH = a.CBH + 4;
  real* const sH = reinterpret_cast<real*>(big_smem);
  real* const cols = sH + KP * SH;  // [4][NC], NC = 2 MP (sse | xsq) or 3 MP (| the Kullback-Leibler divergence per column)
  const int NC = big_ncol(a);
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
  const int ar = M::arow(j);
  const real* __restrict__ Xb = a.X + (long long)b * a.x_bstride;
  const real* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  const real* __restrict__ Hb = a.H + (long long)b * a.k * a.m;
  int row_begin, row_end;
  big_slice(a, row_begin, row_end);
  for (int idx = tid; idx < 4 * NC; idx += 256) cols[idx] = (real)0;
  const bool h_resident = a.CBH >= a.MP;
  if (h_resident) big_stage_h(a, Hb, 0, sH);
  __syncthreads();
  constexpr int V = 16 / (int)sizeof(real);
  real* const mycols = cols + wave * NC;
  for (int cb0 = 0; cb0 < a.MP; cb0 += a.CBH) {
    if (!h_resident) {
      __syncthreads();
      big_stage_h(a, Hb, cb0, sH);
      __syncthreads();
    }
    const int cend = (cb0 + a.CBH < a.MP) ? cb0 + a.CBH : a.MP;
    for (int t0 = row_begin + 16 * wave; t0 < row_end; t0 += 64) {
      const int row = t0 + j;
      const bool rok = row < row_end;
      real w[NKB][4];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) big_load4<real>(Wb + (long long)row * KP, 16 * kb + 4 * g, rok, w[kb]);
      const real* __restrict__ xrow = Xb + (long long)row * a.ldx;
      for (int ch0 = cb0; ch0 < cend; ch0 += 64) {  // four 16-channel blocks per trip: their loads go out together
        real x[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = ch0 + 16 * q + 4 * g;
          const bool cok = rok && ch0 + 16 * q < cend;
#pragma unroll
          for (int r = 0; r < 4; ++r) x[q][r] = (real)0;
          if constexpr (V == 4) {
            big_load4<real>(xrow, col, cok && (col >> 2) < a.xchunks, x[q]);
          } else {
            if (cok && (col >> 1) < a.xchunks) __builtin_memcpy(x[q], __builtin_assume_aligned(xrow + col, 16), 16);
            if (cok && ((col + 2) >> 1) < a.xchunks) __builtin_memcpy(x[q] + 2, __builtin_assume_aligned(xrow + col + 2, 16), 16);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = ch0 + 16 * q;
          if (ch < cend) {
            // R^T block: A = H^T (channel 16-block x components), B = the W fragment; D: lane (row j, g), register r <-> channel ch + 4 g + r
            acc rec = acc{0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
              for (int s = 0; s < 4; ++s) rec = M::mma(sH[(16 * kb + 4 * g + s) * SH + (ch - cb0) + ar], w[kb][s], rec);
            real sse[4], xsq[4], kld[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const real d = x[q][r] - rec[r];
              sse[r] = d * d;
              xsq[r] = x[q][r] * x[q][r];
              kld[r] = (real)0;
            }
            if (a.kl) {  // element by element x log(x / wh) - x + wh, zeros of X skipped, wh clamped (_nmf.py:140-161), as nmf_wide.hpp
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const real xv = x[q][r], whv = rec[r];
                const real whc = whv < eps_val<real>() ? eps_val<real>() : whv;
                const real xs_ = xv > eps_val<real>() ? xv : eps_val<real>();
                const real lg = fma_(xv, log_(xs_ / whc), whv - xv);
                kld[r] = (xv > eps_val<real>()) ? lg : whv;
              }
#pragma unroll
              for (int off = 1; off < 16; off <<= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) kld[r] += __shfl_xor(kld[r], off, 64);
            }
            // sum over the 16 rows of the subtile (lanes j = 0 .. 15 of the same g): a fixed butterfly
#pragma unroll
            for (int off = 1; off < 16; off <<= 1)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                sse[r] += __shfl_xor(sse[r], off, 64);
                xsq[r] += __shfl_xor(xsq[r], off, 64);
              }
            if (j == 0) {  // one lane per 4 channels adds the subtile's sums to the wave's column accumulators (program order)
              const int col = ch + 4 * g;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                mycols[col + r] += sse[r];
                mycols[a.MP + col + r] += xsq[r];
                if (a.kl) mycols[2 * a.MP + col + r] += kld[r];
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  real* __restrict__ out = a.colpart + ((long long)b * a.S + blockIdx.x) * NC;
  for (int idx = tid; idx < NC; idx += 256) out[idx] = ((cols[idx] + cols[NC + idx]) + cols[2 * NC + idx]) + cols[3 * NC + idx];
}

// wide_resid_finalize_kernel with the column buffer in dynamic LDS (2 MP values): error, stop rule (_nmf.py:872-884), outputs
template <typename real>
__global__ void __launch_bounds__(1024) big_resid_finalize_kernel(WideSliceArgs<real> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
  real* const cols = reinterpret_cast<real*>(big_smem);
  const int b = blockIdx.x, tid = threadIdx.x;
  real* st = a.state + (long long)b * 8;
  const bool done = st[3] != (real)0;
  if (done && a.it != -1) return;
  const int NC = (a.kl ? 3 : 2) * a.MP;
  wide_sum_slices<real>(a.colpart + (long long)b * a.S * NC, NC, a.S, cols);
  __syncthreads();
  if (tid == 0) {
    real tot = (real)0;
    for (int jj = 0; jj < a.MP; ++jj) tot += cols[(a.kl ? 2 * a.MP : 0) + jj];
    const real err = a.kl ? sqrt_((real)2 * (tot > (real)0 ? tot : (real)0)) : sqrt_(tot);  // sqrt(2 KL) (_nmf.py:185-189)
    if (a.it == 0) {
      st[0] = err;
      st[1] = err;
    } else if (a.it == 1) {
      st[4] += (real)1;
      if ((st[1] - err) / st[0] < a.tol) {
        st[3] = (real)1;
        st[5] = st[4] * (real)a.check_every;  // n_iter_ of this matrix
      }
      st[1] = err;
    } else {
      if (a.err_out) a.err_out[b] = err;
      if (a.n_iter_out) a.n_iter_out[b] = done ? (int)st[5] : a.max_iter;
    }
    st[2] = err;
  }
  if (a.it == -1) {
    for (int jj = tid; jj < a.m; jj += blockDim.x) {
      if (a.sse_col_out) a.sse_col_out[(long long)b * a.m + jj] = cols[jj];
      if (a.xsq_col_out) a.xsq_col_out[(long long)b * a.m + jj] = cols[a.MP + jj];
    }
  }
}

}  // namespace hipnmf
