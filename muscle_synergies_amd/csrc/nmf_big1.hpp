// nmf_big1.hpp -- ONE pass over X per iteration for the general shapes (round 5): the row-local W update AND the slice's record
// [W'^T X | W'^T W'] in a single kernel, replacing big_pass_w_kernel + big_records_kernel of nmf_big.hpp (which read X twice and
// W three times per iteration: ~2.4x the algorithmic bytes, 0.29 of the fp32 matrix peak at 512 x 32 in round 4).
//
// Arithmetic replaced: sklearn/decomposition/_nmf.py (1.7.2) _multiplicative_update_w (:540-554, 615-631) and the two T-long
// contractions of _multiplicative_update_h (:638-640); reached from src/muscle_synergies/analysis.py:862-863.
//
// Why the accumulators no longer fit one wave -- and what replaces "a wave owns rows".  W^T X is KP x MP values (32 x 512 = 256
// registers per lane in one wave).  Here the EIGHT waves of a workgroup own CHANNEL blocks instead: wave w holds the CW = 16 NQ
// channels [w CW, (w + 1) CW) of everything channel-indexed, for the whole slice, in registers:
//     hreg   H[:, block w]           KP x CW      the A operand of the numerator          (KP CW / 64 registers)
//     accA   (W'^T X)[:, block w]    KP x CW      its block of the slice's record         (KP CW / 64 registers)
// and the rows of a slice go by in ROUNDS of 16 RS rows.  A round:
//   a  every wave: partial numerator^T = H_w X_w^T (KP x rows) over ITS channels, X_w (rows x CW) in registers straight from
//      HBM (16-byte loads; lane (row j, g) <-> channels 16 q + 4 g .. + 3); partial -> LDS (P[w], 16-byte pieces)
//   -- barrier --
//   c  the RS NKB (subtile, component block) UNITS of the round are dealt over the waves: sum of the eight partials (wave
//      order), denominator^T = (H H^T) W^T on the pipe, W' = W num / den, 16-byte store to HBM and a row-major copy to LDS (Wst)
//   -- barrier --
//   e  every wave: accA += W'^T X_w.  The contraction now runs over rows: X_w goes through a private LDS stage once (written
//      as loaded, read back with the channel on the low lane bits), W'^T is read transposed from Wst; the unit owners also
//      accumulate their block row of W'^T W'.  The registers of a staged subtile are free: the NEXT round's loads of that
//      subtile are issued right there (prefetch distance: half a round of matrix-pipe work).
// X is read once, W read once and written once: the algorithmic 4 m + 8 k bytes per row.  Two workgroup barriers per round of
// 64 rows (~17 000 matrix-pipe cycles at 512 x 32).  Every sum has a fixed order (channel blocks in wave order, rounds in
// order, units in subtile order): bitwise reproducible.  Operand conventions: nmf_wide.hpp (WideMma).
#pragma once
#include <type_traits>
#include "nmf_big.hpp"

namespace hipnmf {

// HL: the wave's block of H (the numerator's A operand) lives in LDS instead of registers -- the instances whose register
// budget (256 at two waves per SIMD) it would break: <32, 4, 4> spilled 66 values, reloaded from scratch at the top of every
// round BEHIND the X prefetch in the memory counter's order, i.e. every round waited for its youngest load.  NST: X stages per wave.
// LOSS = 1: the Kullback-Leibler updates (_nmf.py:556-591, 642-684) on the same decomposition.  Per round: (a) every wave
// reconstructs (W H) for ITS channels and the round's rows on the pipe (A = its block of H^T, B = the old rows of W, which the
// unit owners put in LDS one round ahead), Q = X / max(W H, eps), partial numerator^T = H_w Q_w^T; (c) the owners divide by
// rowsum(H) (column 0 of the per-block partial products the H update leaves), store W'; (e) every wave reconstructs (W' H) for
// its channels, Q' = X / max(W' H, eps) goes through the transposition stage instead of X, accA += W'^T Q'; the owners add up
// colsum(W') (column 0 of the record's second block).  Both operand layouts of H come from the wave's LDS block (HL implied).
template <typename real, int KP, int NQ, int RS, bool HL = false, int NST = 2, int LOSS = 0>
struct Big1Cfg {
  static constexpr int NKB = KP / 16, CW = 16 * NQ, NW = 8, NU = RS * NKB, SLOTS = (NU + NW - 1) / NW;
  static constexpr int SW = KP + 4, SX = CW + 4, ROWS = 16 * RS;
  static constexpr int UNIT = 256;                                       // values of one unit's partial numerator (16 rows x 16 components)
  static constexpr int PW0 = NU * UNIT > NST * 16 * SX ? NU * UNIT : NST * 16 * SX;  // per-wave region: the partials | the X stages
  static constexpr int RED = NU * 16 * KP;                              // W'^T W' partial block rows at the end of the slice
  static constexpr int PW = (NW * PW0 >= RED) ? PW0 : (RED + NW - 1) / NW;
  static constexpr int MAXCH = NW * CW;                                  // channels one workgroup covers
  static constexpr bool HLL = HL || LOSS == 1;
  static constexpr int HW = HLL ? KP * SX : 0;                           // per-wave block of H: [KP][CW + 4]
  static constexpr int WOLD = LOSS == 1 ? ROWS * SW : 0;                 // KL: the round's OLD rows of W, row-major
  __host__ __device__ static constexpr size_t smem_bytes() { return sizeof(real) * (size_t)(NW * PW + ROWS * SW + WOLD + NW * HW); }
  static_assert(KP % 16 == 0 && KP >= 16 && KP <= 64 && (NQ == 1 || NQ == 2 || NQ == 4) && RS % 2 == 0, "unsupported shape");
  static_assert(smem_bytes() <= 160 * 1024, "LDS");
};

// grid (S, B), 512 threads; dynamic LDS Big1Cfg::smem_bytes().  a.hht_part holds H H^T as per-block partial products
// (big_hht_part_kernel before the first iteration, big_hupdate_kernel afterwards), a.part receives the records.
template <typename real, int KP, int NQ, int RS, bool HL = false, int NST = 2, int LOSS = 0>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) big1_pass_kernel(BigArgs<real> a) {
  using C = Big1Cfg<real, KP, NQ, RS, HL, NST, LOSS>;
  constexpr bool HLL = C::HLL;
  using M = WideMma<real>;
  using acc = typename M::acc;
  constexpr int NKB = C::NKB, CW = C::CW, NW = C::NW, NU = C::NU, SLOTS = C::SLOTS, SW = C::SW, SX = C::SX, ROWS = C::ROWS,
                UNIT = C::UNIT, PW = C::PW;
  extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
  const int b = blockIdx.y;
  if (a.state && a.state[(long long)b * 8 + 3] != (real)0) return;
  real* const P = reinterpret_cast<real*>(big_smem);  // [NW][PW]
  real* const Wst = P + NW * PW;                      // [ROWS][SW]  the round's updated rows of W, row-major
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ar = M::arow(j);
  real* const Pw = P + wave * PW;
  real* const WoldSt = Wst + ROWS * SW;               // KL: [ROWS][SW] the round's old rows of W
  real* const sHw = WoldSt + C::WOLD + wave * C::HW;  // HLL: [KP][SX] this wave's block of H
  const real* __restrict__ Xb = a.X + (long long)b * a.x_bstride;
  real* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  const real* __restrict__ Hb = a.H + (long long)b * a.k * a.m;
  int row_begin, row_end;
  big_slice(a, row_begin, row_end);
  const int ch_w0 = wave * CW;
  const bool active = ch_w0 < a.MP;                   // this wave's channel block holds data
  const int nwa = (a.MP + CW - 1) / CW;               // waves whose partials are summed
  const bool upd = a.update_h != 0;
  const unsigned ldx_b = (unsigned)(a.ldx * (long long)sizeof(real));
  constexpr unsigned ldw_b = (unsigned)KP * (unsigned)sizeof(real);
  constexpr int V = 16 / (int)sizeof(real);           // elements per 16-byte piece

  // ---- operands that live in registers for the whole slice ---------------------------------------------------------
  // numerator's A operand: lane (i, g), k-step (q, r) <-> H[16 kb + arow(i)][ch_w0 + 16 q + 4 g + r]
  real hreg[HLL ? 1 : NKB][HLL ? 1 : NQ][4];
  if constexpr (HLL) {  // private to the wave: written and read by the same wave, no barrier
    for (int idx = lane; idx < KP * CW; idx += 64) {
      const int c = idx / CW, ch = ch_w0 + idx % CW;
      sHw[c * SX + idx % CW] = (c < a.k && ch < a.m) ? Hb[(long long)c * a.m + ch] : (real)0;
    }
    wide_wave_lds_fence();
  } else {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * kb + ar, ch = ch_w0 + 16 * q + 4 * g + r;
          hreg[kb][q][r] = (c < a.k && ch < a.m) ? Hb[(long long)c * a.m + ch] : (real)0;
        }
  }
  // the units this wave owns in phase c: v = wave + NW slot <-> (subtile v / NKB, component block v % NKB)
  int u_s[SLOTS], u_kb[SLOTS];
  real hha[SLOTS][NKB][4];  // denominator's A operand: lane (i, g), k-step r of block kbi <-> HHt[16 kbo + arow(i)][16 kbi + 4 g + r]
#pragma unroll
  for (int sl = 0; sl < SLOTS; ++sl) {
    const int v = wave + NW * sl;
    u_s[sl] = v < NU ? v / NKB : -1;
    u_kb[sl] = v < NU ? v % NKB : 0;
    if constexpr (LOSS == 1) {  // rowsum(H) of the unit's components 16 kbo + 4 g + r: column 0 of the partial products, block order
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const real* __restrict__ hp = a.hht_part + (long long)b * a.n_hblk * KP * KP + (16 * u_kb[sl] + 4 * g + r) * KP;
        real sH = hp[0];
        for (int blk = 1; blk < a.n_hblk; ++blk) sH += hp[(long long)blk * KP * KP];
        hha[sl][0][r] = sH;
      }
    } else
#pragma unroll
    for (int kbi = 0; kbi < NKB; ++kbi) {  // H H^T = the 64-channel blocks' partial products added in block order
      const real* __restrict__ hp = a.hht_part + (long long)b * a.n_hblk * KP * KP + (16 * u_kb[sl] + ar) * KP + 16 * kbi + 4 * g;
      wide_lds_read<real, 4>(hp, hha[sl][kbi]);
      for (int blk = 1; blk < a.n_hblk; ++blk) {
        real t4[4];
        wide_lds_read<real, 4>(hp + (long long)blk * KP * KP, t4);
#pragma unroll
        for (int r = 0; r < 4; ++r) hha[sl][kbi][r] += t4[r];
      }
    }
  }
  acc accA[NKB][NQ], accB[SLOTS][NKB];
  real wsum[SLOTS][4];  // KL: this lane's share of colsum(W') (its row, the unit's components)
#pragma unroll
  for (int sl = 0; sl < SLOTS; ++sl)
#pragma unroll
    for (int r = 0; r < 4; ++r) wsum[sl][r] = (real)0;
  const acc zero = {(real)0, (real)0, (real)0, (real)0};
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int q = 0; q < NQ; ++q) accA[kb][q] = zero;
#pragma unroll
  for (int sl = 0; sl < SLOTS; ++sl)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) accB[sl][kb] = zero;

  // ---- addressing: descriptors start at a subtile's first row and end with the slice, so a lane is in range exactly when its
  // row exists; pieces of a row beyond the data get the out-of-range sentinel once
  unsigned xvoff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int col = ch_w0 + 16 * q + 4 * g;
    xvoff[q] = (active && col / V < a.xchunks) ? (unsigned)j * ldx_b + (unsigned)col * (unsigned)sizeof(real) : OOB;
  }
  const char* const xbase = reinterpret_cast<const char*>(Xb);
  char* const wbase = reinterpret_cast<char*>(Wb);
  auto x_rsrc = [&](int row0) __attribute__((always_inline)) {
    const int rows = row0 < row_end ? row_end - row0 : 0;
    return make_rsrc(xbase + (long long)(rows > 0 ? row0 : row_begin) * ldx_b, (unsigned)rows * ldx_b);
  };
  auto w_rsrc = [&](int row0) __attribute__((always_inline)) {
    const int rows = row0 < row_end ? row_end - row0 : 0;
    return make_rsrc(wbase + (long long)(rows > 0 ? row0 : row_begin) * ldw_b, (unsigned)rows * ldw_b);
  };
  real x[RS][NQ][4];  // the round's rows of this wave's channel block: lane (row j, g) <-> X[16 s + j][ch_w0 + 16 q + 4 g + r]
  auto issue_x = [&](int s, int row0) __attribute__((always_inline)) {
    const rsrc_t xr = x_rsrc(row0 + 16 * s);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if constexpr (V == 4) {
        buf_load<real, 4, (HIPNMF_WIDE_X_AUX)>(xr, xvoff[q], 0u, x[s][q]);
      } else {  // fp64: two 16-byte pieces, the second one may lie beyond the data
        real lo[2], hi[2];
        const int col = ch_w0 + 16 * q + 4 * g;
        buf_load<real, 2, (HIPNMF_WIDE_X_AUX)>(xr, xvoff[q], 0u, lo);
        buf_load<real, 2, (HIPNMF_WIDE_X_AUX)>(xr, (xvoff[q] != OOB && (col + 2) / V < a.xchunks) ? xvoff[q] + 16u : OOB, 0u, hi);
        x[s][q][0] = lo[0], x[s][q][1] = lo[1], x[s][q][2] = hi[0], x[s][q][3] = hi[1];
      }
    }
  };

  if (active) {
#pragma unroll
    for (int s = 0; s < RS; ++s) issue_x(s, row_begin);
  }
  // (W H) block for the lane's row and the channel block q: A = H^T block (lane (channel i, g), k-step (kb, t) <->
  // H[16 kb + 4 g + t][ch_w0 + 16 q + arow(i)]), B = a row-major W fragment; D: lane (row j, g), register r <-> channel 16 q + 4 g + r
  auto wh_block = [&](const real (&w)[NKB][4], int q) __attribute__((always_inline)) -> acc {
    acc rec = zero;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t) rec = M::mma(sHw[(16 * kb + 4 * g + t) * SX + 16 * q + ar], w[kb][t], rec);
    return rec;
  };
  real wold1[SLOTS][4];  // KL: the unit's own block of the old rows (the denominator needs no other)
  if constexpr (LOSS == 1) {
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      const rsrc_t wr = w_rsrc(u_s[sl] >= 0 ? row_begin + 16 * u_s[sl] : row_end);
      buf_load<real, 4>(wr, (unsigned)j * ldw_b + (unsigned)((16 * u_kb[sl] + 4 * g) * (int)sizeof(real)), 0u, wold1[sl]);
      if (u_s[sl] >= 0) wide_lds_write<real, 4>(WoldSt + (16 * u_s[sl] + j) * SW + 16 * u_kb[sl] + 4 * g, wold1[sl]);
    }
    __syncthreads();
  }
#ifdef HIPNMF_BIG1_TIMING  // development build: shader-clock time per phase of this wave, printed by workgroup (0, 0)
  long long tphase[5] = {0, 0, 0, 0, 0}, tlast = (long long)__builtin_readcyclecounter();
#define BIG1_LAP(i)                                                  \
  {                                                                  \
    const long long now_ = (long long)__builtin_readcyclecounter();  \
    tphase[i] += now_ - tlast;                                       \
    tlast = now_;                                                    \
  }
#else
#define BIG1_LAP(i)
#endif
  // The round loop exists twice -- for waves whose channel block holds data and for those beyond the matrix (they only take part in
  // phase c and the barriers) -- so that no wave-uniform branch on `active` sits INSIDE it: such branches made the compiler keep two
  // register copies of the accumulators around every subtile (16 v_mov_b64 each way) and a second set of X registers.  Rows past
  // the slice need no branch either: their descriptors are empty, the prefetch moves nothing.
  auto rounds = [&](auto act_) __attribute__((always_inline)) {
    constexpr bool ACT = decltype(act_)::value;
    for (int t0 = row_begin; t0 < row_end; t0 += ROWS) {
      // the W fragments of this wave's units (B operand of the denominator: lane (row j, g) <-> components 16 kbi + 4 g .. + 3):
      // requested now, needed after the first barrier
      real wold[LOSS == 1 ? 1 : SLOTS][LOSS == 1 ? 1 : NKB][4];
      real wnext[SLOTS][4];  // KL: the unit's block of the NEXT round's old rows (into LDS in phase c)
#pragma unroll
      for (int sl = 0; sl < SLOTS; ++sl) {
        if constexpr (LOSS == 1) {
          const rsrc_t wr = w_rsrc(u_s[sl] >= 0 ? t0 + ROWS + 16 * u_s[sl] : row_end);
          buf_load<real, 4>(wr, (unsigned)j * ldw_b + (unsigned)((16 * u_kb[sl] + 4 * g) * (int)sizeof(real)), 0u, wnext[sl]);
        } else {
          const rsrc_t wr = w_rsrc(u_s[sl] >= 0 ? t0 + 16 * u_s[sl] : row_end);
#pragma unroll
          for (int kbi = 0; kbi < NKB; ++kbi) buf_load<real, 4>(wr, (unsigned)j * ldw_b + (unsigned)((16 * kbi + 4 * g) * (int)sizeof(real)), 0u, wold[sl][kbi]);
        }
      }
      // ---- a: partial numerators of the round over this wave's channels, two subtiles (2 NKB chains) at a time ------------
      if constexpr (ACT) {
#pragma unroll
        for (int s = 0; s < RS; s += 2) {
          acc num[2][NKB];
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) num[e][kb] = zero;
          real wB[LOSS == 1 ? 2 : 1][LOSS == 1 ? NKB : 1][4];  // KL: the old rows of the two subtiles, row-major fragments
          if constexpr (LOSS == 1) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
              for (int kb = 0; kb < NKB; ++kb) wide_lds_read<real, 4>(WoldSt + (16 * (s + e) + j) * SW + 16 * kb + 4 * g, wB[e][kb]);
          }
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            real xq[2][4];  // the B operand of the numerator: X, or Q = X / max(W H, eps) (_nmf.py:574-577)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              if constexpr (LOSS == 1) {
                const acc rec = wh_block(wB[e], q);
#pragma unroll
                for (int r = 0; r < 4; ++r) xq[e][r] = big_quot(x[s + e][q][r], kl_floor(rec[r]));
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) xq[e][r] = x[s + e][q][r];
              }
            }
            real ha[NKB][4];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
              if constexpr (HLL) {
                wide_lds_read<real, 4>(sHw + (16 * kb + ar) * SX + 16 * q + 4 * g, ha[kb]);
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) ha[kb][r] = hreg[kb][q][r];
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) num[e][kb] = M::mma(ha[kb][r], xq[e][r], num[e][kb]);
          }
          // D: lane (row j, g), register r <-> component 16 kb + 4 g + r: one 16-byte piece per lane and unit, stored at 4 * lane
          // (consecutive lanes on consecutive pieces: [row][component] order put eight lanes of a pass on two bank groups)
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
              real v4[4] = {num[e][kb][0], num[e][kb][1], num[e][kb][2], num[e][kb][3]};
              wide_lds_write<real, 4>(Pw + ((s + e) * NKB + kb) * UNIT + 4 * lane, v4);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // denominator^T = (H H^T) W^T of this wave's units: needs only the old rows, so it runs BEFORE the barrier (the pipe
      // is idle there while the waves arrive; after it the owners' chain is partials -> quotient -> stores only)
      acc denu[SLOTS];
      if constexpr (LOSS == 0) {
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
          // NOT under `if (u_s[sl] >= 0)`: a slot without a unit loaded zeros through an empty descriptor, and a load that is
          // consumed on one path only stays "pending" on the other in the compiler's wait-count model -- every later write to its
          // registers then waited for vmcnt(0), i.e. for the X prefetch issued just before: one HBM latency per SUBTILE of phase e
          // (profiles/r05_big1_phase_timing.txt: 10.3 k cycles for 5.1 k of MFMAs)
          denu[sl] = zero;
#pragma unroll
          for (int kbi = 0; kbi < NKB; ++kbi)
#pragma unroll
            for (int r = 0; r < 4; ++r) denu[sl] = M::mma(hha[sl][kbi][r], wold[sl][kbi][r], denu[sl]);
        }
      }
      BIG1_LAP(0)  // a: partial numerators (+ denominator)
      __syncthreads();
      BIG1_LAP(1)  // wait at the first barrier
      // ---- c: this wave's units: sum of the partials, W' ----------------------------------------------------------------
      // The LDS addresses of this phase are recomputed every round from an opaque copy of the lane id.  Hoisted out of the loop
      // they were the four values hipcc spilled at the 256-register cap, and a scratch reload HERE is expensive far beyond its
      // own latency: its destination register is reused by the first MFMA of phase e, so the compiler waits for vmcnt(0) there --
      // with the next round's X prefetch already in flight, i.e. for HBM latency, every round (measured: phase e 10.6 k cycles on
      // the older wave of a SIMD and 15 - 16 k on the younger one, against 5.1 k of matrix-pipe work each).
      int lane_r = lane;
      asm volatile("" : "+v"(lane_r));
      const int j_r = lane_r & 15, g_r = lane_r >> 4;
#pragma unroll
      for (int sl = 0; sl < SLOTS; ++sl) {
        if (u_s[sl] < 0) {
          if constexpr (LOSS == 1) {  // (the slot's loads must not stay pending on this path: see the denominator above)
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(wnext[sl][r]));
          }
          continue;
        }
        const int v = wave + NW * sl;
        real nsum[4];
        wide_lds_read<real, 4>(P + v * UNIT + 4 * lane_r, nsum);
        for (int w2 = 1; w2 < nwa; ++w2) {  // (channel blocks in wave order)
          real t4[4];
          wide_lds_read<real, 4>(P + w2 * PW + v * UNIT + 4 * lane_r, t4);
#pragma unroll
          for (int r = 0; r < 4; ++r) nsum[r] += t4[r];
        }
        acc den = zero;
        real wo[4], dd[4], qq[4], wn[4];
        if constexpr (LOSS == 1) {  // W *= ((X / WH) H^T) / rowsum(H)   (_nmf.py:577-581)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            den[r] = hha[sl][0][r];
            wo[r] = wold1[sl][r];
          }
        } else {
          den = denu[sl];
#pragma unroll
          for (int kbi = 0; kbi < NKB; ++kbi)
            if (kbi == u_kb[sl]) {
#pragma unroll
              for (int r = 0; r < 4; ++r) wo[r] = wold[sl][kbi][r];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          real d = den[r];
          if (a.l1w > (real)0) d = d + a.l1w;
          if (a.l2w > (real)0) d = d + a.l2w * wo[r];
          dd[r] = (d == (real)0) ? eps_val<real>() : d;
        }
        quotients<4>(nsum, dd, qq);
#pragma unroll
        for (int r = 0; r < 4; ++r) wn[r] = wo[r] * qq[r];
        const rsrc_t wr = w_rsrc(t0 + 16 * u_s[sl]);
        wide_store4<real>(wr, (unsigned)j * ldw_b + (unsigned)((16 * u_kb[sl] + 4 * g) * (int)sizeof(real)), 0u, wn);
        if (upd || LOSS == 1) wide_lds_write<real, 4>(Wst + (16 * u_s[sl] + j_r) * SW + 16 * u_kb[sl] + 4 * g_r, wn);
        if constexpr (LOSS == 1) {  // the next round's old rows (this round's were read before the barrier), colsum(W')
          wide_lds_write<real, 4>(WoldSt + (16 * u_s[sl] + j_r) * SW + 16 * u_kb[sl] + 4 * g_r, wnext[sl]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            wsum[sl][r] += wn[r];
            wold1[sl][r] = wnext[sl][r];
          }
        }
      }
      if (!upd) {  // nothing else needs X: the next round's rows may come
        if constexpr (ACT) {
#pragma unroll
          for (int s = 0; s < RS; ++s) issue_x(s, t0 + ROWS);
        }
        __syncthreads();  // (the partials are rewritten by the next round)
        continue;
      }
      BIG1_LAP(2)  // c: the owners' chain
      __syncthreads();
      BIG1_LAP(3)  // wait at the second barrier
      // ---- e: accA += W'^T X_w (contraction over the rows), W'^T W' by the unit owners ----------------------------------
      // Frobenius with two stages: subtile s + 1 is written into the other stage BEFORE the products of subtile s (its registers hold
      // the data already), so its write -> read round trip runs under those MFMAs instead of in front of its own
      constexpr bool AHEAD = LOSS == 0 && NST == 2;
      if constexpr (AHEAD) {
        if constexpr (ACT) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) wide_lds_write<real, 4>(Pw + j * SX + 16 * q + 4 * g, x[0][q]);
        }
      }
#pragma unroll
      for (int s = 0; s < RS; ++s) {
        real* const xst = Pw + (s % NST) * 16 * SX;  // this wave's stage (its partials are dead; nobody else touches the region now)
        if constexpr (ACT && !AHEAD) {
          if constexpr (LOSS == 1) {  // Q' = X / max(W' H, eps) with the updated rows (_nmf.py:660-663) takes X's place in the stage
            real wnB[NKB][4];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) wide_lds_read<real, 4>(Wst + (16 * s + j) * SW + 16 * kb + 4 * g, wnB[kb]);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
              const acc rec = wh_block(wnB, q);
              real qv[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) qv[r] = big_quot(x[s][q][r], kl_floor(rec[r]));
              wide_lds_write<real, 4>(xst + j * SX + 16 * q + 4 * g, qv);
            }
          } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) wide_lds_write<real, 4>(xst + j * SX + 16 * q + 4 * g, x[s][q]);
          }
        }
        wide_wave_lds_fence();
        if constexpr (AHEAD) {
          if constexpr (ACT) if (s + 1 < RS) {
            real* const nst = Pw + ((s + 1) % NST) * 16 * SX;  // (last read by subtile s - 1: in order behind those reads)
#pragma unroll
            for (int q = 0; q < NQ; ++q) wide_lds_write<real, 4>(nst + j * SX + 16 * q + 4 * g, x[s + 1][q]);
          }
        }
        if constexpr (ACT) issue_x(s, t0 + ROWS);  // the registers are free: the next round's subtile s
        // A: lane (c, g), k-step t <-> W'[row 4 g + t][16 kb + arow(c)];  B: lane (channel j, g) <-> X[row 4 g + t][16 q + j]
        real wa[NKB][4], wb[NKB][4];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            wa[kb][t] = Wst[(16 * s + 4 * g + t) * SW + 16 * kb + ar];
            if constexpr (sizeof(real) == 8)
              wb[kb][t] = Wst[(16 * s + 4 * g + t) * SW + 16 * kb + j];
            else
              wb[kb][t] = wa[kb][t];
          }
        if constexpr (ACT) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            real xb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) xb[t] = xst[(4 * g + t) * SX + 16 * q + j];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int kb = 0; kb < NKB; ++kb) accA[kb][q] = M::mma(wa[kb][t], xb[t], accA[kb][q]);
          }
        }
        if constexpr (LOSS == 0)
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl)
          if (u_s[sl] == s) {  // (wave-uniform)
#pragma unroll
            for (int kbo = 0; kbo < NKB; ++kbo)
              if (kbo == u_kb[sl]) {
#pragma unroll
                for (int kbi = 0; kbi < NKB; ++kbi)
#pragma unroll
                  for (int t = 0; t < 4; ++t) accB[sl][kbi] = M::mma(wa[kbo][t], wb[kbi][t], accB[sl][kbi]);
              }
          }
        wide_wave_lds_fence();
        __builtin_amdgcn_sched_barrier(0);  // one subtile at a time: hoisting the next one's LDS reads up here costs spills
      }
      BIG1_LAP(4)  // e: transposition + W'^T X (+ W'^T W')
    }
  };
  if (active)
    rounds(std::true_type{});
  else
    rounds(std::false_type{});
#ifdef HIPNMF_BIG1_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0)
    printf("big1 wave %d rounds %d: a %lld  wait1 %lld  c %lld  wait2 %lld  e %lld  (cycles per round)\n", wave, (row_end - row_begin + ROWS - 1) / ROWS,
           tphase[0] / ((row_end - row_begin + ROWS - 1) / ROWS), tphase[1] / ((row_end - row_begin + ROWS - 1) / ROWS),
           tphase[2] / ((row_end - row_begin + ROWS - 1) / ROWS), tphase[3] / ((row_end - row_begin + ROWS - 1) / ROWS),
           tphase[4] / ((row_end - row_begin + ROWS - 1) / ROWS));
#endif
#undef BIG1_LAP
  if (!upd) return;
  // ---- the slice's record.  W'^T X: every wave owns its channel block outright.  D: lane (j, g), register r <->
  // [component 16 kb + 4 g + r][channel ch_w0 + 16 q + j]
  const int rec = KP * a.MP + KP * KP;
  real* __restrict__ out = a.part + ((long long)b * a.S + blockIdx.x) * rec;
  if (active) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ch = ch_w0 + 16 * q + j;
        if (ch < a.MP) {
#pragma unroll
          for (int r = 0; r < 4; ++r) out[(16 * kb + 4 * g + r) * a.MP + ch] = accA[kb][q][r];
        }
      }
  }
  // W'^T W': block row kbo of unit v = (s, kbo) -> red[v][16][KP]; the units of one block row summed in subtile order
  __syncthreads();  // (the stages inside P are dead)
  real* const red = P;
  if constexpr (LOSS == 1) {  // colsum(W') in column 0 of the block (the 16 row lanes j of one g: a fixed butterfly), zeros elsewhere
    for (int idx = tid; idx < NU * 16 * KP; idx += 512) red[idx] = (real)0;
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      if (u_s[sl] < 0) continue;
      const int v = wave + NW * sl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        real sw = wsum[sl][r];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) sw += __shfl_xor(sw, off, 64);
        if (j == 0) red[(v * 16 + 4 * g + r) * KP] = sw;
      }
    }
  } else {
#pragma unroll
  for (int sl = 0; sl < SLOTS; ++sl) {
    if (u_s[sl] < 0) continue;
    const int v = wave + NW * sl;
#pragma unroll
    for (int kbi = 0; kbi < NKB; ++kbi)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(v * 16 + 4 * g + r) * KP + 16 * kbi + j] = accB[sl][kbi][r];
  }
  }
  __syncthreads();
  for (int idx = tid; idx < KP * KP; idx += 512) {
    const int c = idx / KP, c2 = idx % KP, kbo = c / 16, i = c % 16;
    real s = (real)0;
#pragma unroll
    for (int sb = 0; sb < RS; ++sb) s += red[((sb * NKB + kbo) * 16 + i) * KP + c2];
    out[KP * a.MP + idx] = s;
  }
}

// ---- the residual on the same decomposition: per-column sum((X - W H)^2) | sum(X^2) (| the Kullback-Leibler divergence per
// column) of one slice (_beta_divergence, _nmf.py:85-134, evaluated every check_every iterations by the stop rule :872-884 and
// once at the end; vaf, analysis.py:597-667).  Wave w owns the channels [w CW, (w + 1) CW) of ALL rows of the slice: its block of
// H^T sits in registers as the A operand, X_w streams through registers, the W fragment (the B operand, the same for every
// wave) comes from HBM / L2 -- no LDS, no barrier, every column's sum is produced by one wave: rows in order per lane, then a
// fixed butterfly over the 16 row lanes.  Replaces big_resid_kernel for fp32 (1.29 ms per evaluation at 64 x (512 x 10 000),
// k = 32 -- 2.4x the whole update pass -- because it re-staged H per channel block between workgroup barriers).
// grid (S, B), 512 threads, no dynamic LDS.
template <typename real, int KP, int NQ>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) big1_resid_kernel(BigArgs<real> a) {
  using M = WideMma<real>;
  using acc = typename M::acc;
  constexpr int NKB = KP / 16, CW = 16 * NQ;
  const int b = blockIdx.y;
  if (a.state && a.state[(long long)b * 8 + 3] != (real)0) return;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ar = M::arow(j);
  const int ch_w0 = wave * CW;
  if (ch_w0 >= a.MP) return;  // (no barrier in this kernel)
  const real* __restrict__ Xb = a.X + (long long)b * a.x_bstride;
  const real* __restrict__ Wb = a.W + (long long)b * a.w_bstride;
  const real* __restrict__ Hb = a.H + (long long)b * a.k * a.m;
  int row_begin, row_end;
  big_slice(a, row_begin, row_end);
  const unsigned ldx_b = (unsigned)(a.ldx * (long long)sizeof(real));
  constexpr unsigned ldw_b = (unsigned)KP * (unsigned)sizeof(real);
  constexpr int V = 16 / (int)sizeof(real);
  // A operand: lane (channel i, g), k-step (kb, s) <-> H[16 kb + 4 g + s][ch_w0 + 16 q + arow(i)]
  real hT[NKB][NQ][4];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int c = 16 * kb + 4 * g + s, ch = ch_w0 + 16 * q + ar;
        hT[kb][q][s] = (c < a.k && ch < a.m) ? Hb[(long long)c * a.m + ch] : (real)0;
      }
  unsigned xvoff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int col = ch_w0 + 16 * q + 4 * g;
    xvoff[q] = (col / V < a.xchunks) ? (unsigned)j * ldx_b + (unsigned)col * (unsigned)sizeof(real) : OOB;
  }
  const char* const xbase = reinterpret_cast<const char*>(Xb);
  const char* const wbase = reinterpret_cast<const char*>(Wb);
  struct Tile {
    real x[NQ][4];
    real w[NKB][4];
  };
  auto issue = [&](Tile& t, int row0) __attribute__((always_inline)) {
    const int rows = row0 < row_end ? row_end - row0 : 0;
    const rsrc_t xr = make_rsrc(xbase + (long long)(rows > 0 ? row0 : row_begin) * ldx_b, (unsigned)rows * ldx_b);
    const rsrc_t wr = make_rsrc(wbase + (long long)(rows > 0 ? row0 : row_begin) * ldw_b, (unsigned)rows * ldw_b);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if constexpr (V == 4) {
        buf_load<real, 4, (HIPNMF_WIDE_X_AUX)>(xr, xvoff[q], 0u, t.x[q]);
      } else {
        real lo[2], hi[2];
        const int col = ch_w0 + 16 * q + 4 * g;
        buf_load<real, 2, (HIPNMF_WIDE_X_AUX)>(xr, xvoff[q], 0u, lo);
        buf_load<real, 2, (HIPNMF_WIDE_X_AUX)>(xr, (xvoff[q] != OOB && (col + 2) / V < a.xchunks) ? xvoff[q] + 16u : OOB, 0u, hi);
        t.x[q][0] = lo[0], t.x[q][1] = lo[1], t.x[q][2] = hi[0], t.x[q][3] = hi[1];
      }
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) buf_load<real, 4>(wr, (unsigned)j * ldw_b + (unsigned)((16 * kb + 4 * g) * (int)sizeof(real)), 0u, t.w[kb]);
  };
  real sse[NQ][4], xsq[NQ][4], kld[NQ][4];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) sse[q][r] = xsq[q][r] = kld[q][r] = (real)0;
  const acc zero = {(real)0, (real)0, (real)0, (real)0};
  auto consume = [&](const Tile& t) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      acc rec = zero;  // (W H) block: D: lane (row j, g), register r <-> channel ch_w0 + 16 q + 4 g + r -- where the lane's X values sit
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s) rec = M::mma(hT[kb][q][s], t.w[kb][s], rec);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const real xv = t.x[q][r], d = xv - rec[r];
        sse[q][r] = fma_(d, d, sse[q][r]);
        xsq[q][r] = fma_(xv, xv, xsq[q][r]);
        if (a.kl) {  // x log(x / wh) - x + wh, zeros of X skipped, wh clamped (_nmf.py:140-161), as big_resid_kernel
          const real whv = rec[r];
          const real whc = whv < eps_val<real>() ? eps_val<real>() : whv;
          const real xs_ = xv > eps_val<real>() ? xv : eps_val<real>();
          const real lg = fma_(xv, log_(xs_ / whc), whv - xv);
          kld[q][r] += (xv > eps_val<real>()) ? lg : whv;
        }
      }
    }
  };
  Tile ta, tb;
  issue(ta, row_begin);
  __builtin_amdgcn_sched_barrier(0);
  issue(tb, row_begin + 16);
  __builtin_amdgcn_sched_barrier(0);
  for (int t0 = row_begin; t0 < row_end; t0 += 32) {  // (slices are whole multiples of 64 rows except the last; rows past the end read 0)
    consume(ta);
    __builtin_amdgcn_sched_barrier(0);
    issue(ta, t0 + 32);
    __builtin_amdgcn_sched_barrier(0);
    consume(tb);
    __builtin_amdgcn_sched_barrier(0);
    issue(tb, t0 + 48);
    __builtin_amdgcn_sched_barrier(0);
  }
  // the 16 row lanes j of one g hold partial sums of the same channels: a fixed butterfly over the low four lane bits
  const int NC = big_ncol(a);
  real* __restrict__ out = a.colpart + ((long long)b * a.S + blockIdx.x) * NC;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      real s1 = sse[q][r], s2 = xsq[q][r], s3 = kld[q][r];
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
        if (a.kl) s3 += __shfl_xor(s3, off, 64);
      }
      const int ch = ch_w0 + 16 * q + 4 * g + r;
      if (j == 0 && ch < a.MP) {
        out[ch] = s1;
        out[a.MP + ch] = s2;
        if (a.kl) out[2 * a.MP + ch] = s3;
      }
    }
}

}  // namespace hipnmf
