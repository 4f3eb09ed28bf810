// Wave-level building blocks of the blocked Cholesky factorisations (dense: cx_cholesky.hip, tile-sparse:
// cx_sparse_chol.hip): the 32x32 diagonal factor + inverse by one wavefront, and the panel solve
// X = U_kk^-T W(k, .) as an fp64 MFMA product whose result registers are the operands of the trailing update.
#ifndef CX_CHOL_BLOCKS_H_
#define CX_CHOL_BLOCKS_H_

#include "cx_internal.h"

namespace cxchol {

constexpr int NB = 32;  // panel width: the wave-level diagonal factorisation keeps a 32x32 block in registers

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kPotrfLds = NB * NB + NB + 2 * 16 * 16;  // LDS doubles of potrf_inverse_block: U, 1 / diag(U), V11 | V22

// v at lane `l` (compile-time constant), as a wave-uniform value: two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane(unsigned(u), l);
  const unsigned hi = __builtin_amdgcn_readlane(unsigned(u >> 32), l);
  return __longlong_as_double((static_cast<unsigned long long>(hi) << 32) | lo);
}

// Second half of the diagonal-block routines below: U (upper, row-major, zeros below the diagonal) and 1 / diag(U) are in
// LDS (lds[r * NB + c], lds[NB * NB + j]); store U into the factor and form U^-1 (see potrf_inverse_block).
// uinv_lds (optional, NB * NB doubles of LDS outside `lds`): a second copy of the inverse for a caller that goes on to use it.
// TF: the scalar type the factor is stored in (double; float for the fp32 tile pool of use_mixed_precision_solves)
template <typename TF>
__device__ __forceinline__ void potrf_tail(TF* __restrict__ Fblk, int ldf, int kb, double* __restrict__ uinv, double* __restrict__ lds,
                                           double* __restrict__ uinv_lds = nullptr) {
  const int lane = threadIdx.x & 63;
  // (single wavefront: its LDS writes are ordered before its LDS reads, no barrier needed)
  // U into the factor from the LDS copy, two rows per store instruction (all 64 lanes) instead of one
#pragma unroll
  for (int it = 0; it < NB / 2; ++it) {
    const int r = 2 * it + (lane >> 5), c = lane & 31;
    const double v = lds[r * NB + c];
    if (r < kb && c < kb && c >= r) Fblk[size_t(r) * ldf + c] = TF(v);
  }
  // V = U^-1 by blocks of 16: U = [U11 U12; 0 U22]  =>  V = [V11  -V11 U12 V22; 0  V22].
  //  * V11 and V22 side by side: lanes 0-15 own the columns of V11, lanes 16-31 those of V22 (16 registers each),
  //    right-looking back substitution: 240 fp64 FMAs of this one wavefront instead of the 496 of the unblocked
  //    inverse (an fp64 FMA costs a lone wavefront 8 cycles);
  //  * the off-diagonal block as two 16x16x16 products on the matrix cores (8 v_mfma_f64_16x16x4_f64); the result
  //    registers of T = U12 V22 are already the B operands of V11 T.
  const int half = (lane >> 4) & 1, lc = lane & 15, o = 16 * half;  // lanes 32-63 mirror 0-31 and store nothing
  double V[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) V[r] = (r == lc) ? 1.0 : 0.0;
#pragma unroll
  for (int r = 15; r >= 0; --r) {
    V[r] = (lc >= r) ? V[r] * lds[NB * NB + o + r] : 0.0;
#pragma unroll
    for (int q = 0; q < r; ++q) V[q] -= lds[(o + q) * NB + o + r] * V[r];
  }
  double* __restrict__ Vs = lds + NB * NB + NB;  // V11 | V22, 16 x 16 row-major each
  if (lane < NB) {
#pragma unroll
    for (int r = 0; r < 16; ++r) Vs[half * 256 + r * 16 + lc] = V[r];
  }
  {
    const int li = lane & 15, lk = lane >> 4;
    double4_t Tm = double4_t{0.0, 0.0, 0.0, 0.0}, X = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)  // T = U12 V22: A(i, k) = U(i, 16 + k), B(k, j) = V22(k, j)
      Tm = __builtin_amdgcn_mfma_f64_16x16x4f64(lds[li * NB + 16 + 4 * s4 + lk], Vs[256 + (4 * s4 + lk) * 16 + li], Tm, 0, 0, 0);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)  // X = V11 T: register s4 of T holds T(4 s4 + lk, li), the B operand of step s4
      X = __builtin_amdgcn_mfma_f64_16x16x4f64(Vs[li * 16 + 4 * s4 + lk], Tm[s4], X, 0, 0, 0);
    // the whole inverse, 4 rows x 16 columns per store instruction: [V11 | -X] over [0 | V22]
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int r = lk + 4 * g;
      const double v11 = Vs[r * 16 + li], v12 = -X[g], v22 = Vs[256 + r * 16 + li];
      uinv[r * NB + li] = v11;
      uinv[r * NB + 16 + li] = v12;
      uinv[(16 + r) * NB + li] = 0.0;
      uinv[(16 + r) * NB + 16 + li] = v22;
      if (uinv_lds) {
        uinv_lds[r * NB + li] = v11;
        uinv_lds[r * NB + 16 + li] = v12;
        uinv_lds[(16 + r) * NB + li] = 0.0;
        uinv_lds[(16 + r) * NB + 16 + li] = v22;
      }
    }
  }
}

// One wavefront: factor the kb x kb diagonal block of W at k0 (upper, U'U), store U_kk into the factor
// F (ld n) and U_kk^-1 (NB x NB row-major, identity-padded) into uinv.  Lane c owns column c of the block
// in NB registers; right-looking, fully unrolled, pivot row broadcast by v_readlane (no LDS round trip
// or barrier in the 32-step chain).  The inverse is then formed column by column from an LDS copy of U
// (broadcast reads), lds: kPotrfLds doubles.
// Measured, MI355X: 8.9 us per call warm with the updates pinned (see the inner loop; 13.2 us before, when the
// optimiser had turned the loop into a left-looking factorisation with 700 SGPR spills -- Dubrovnik-356 Cholesky 3.15 ->
// 2.97 ms, tile-sparse Final 125.1 -> 120.6 ms on the same box).  Of the 13.2 us (tools/potrf_bench.hip): 20 k of its 31 k cycles in the 496 broadcast + fp64
// FMA pairs of the factorisation -- an fp64 FMA costs a lone wavefront 8 cycles --, 6 k in its 64 narrow stores,
// 5 k in the blocked inverse below; 14.4 us with the unblocked inverse: one column per lane, 496 FMAs), ~18 us cold
// inside a step kernel (23 us before).  In place the blocked inverse is worth 4 %: Dubrovnik-356 Cholesky 3.32 ->
// 3.17 ms, tile-sparse Final 130.5 -> 125.2 ms (same box).  A/B: blocking the FACTORISATION by 16 as well (two 16-column
// factorisations, U12 and the Schur update of the second block on v_mfma_f64_16x16x4_f64, a third of the
// instructions) ran 9.3 us warm but LOST in place on the same box:
// Dubrovnik-356 reduced solve 4.18 -> 4.29 ms, tile-sparse Final 175 -> 186 ms; it is not kept.  Broadcasting the pivot
// row through LDS (one ds_read per row instead of two v_readlane) was slower both ways: 20.1 us warm, Dubrovnik-356
// 3.29 -> 3.82 ms.
// Pointer form: Wblk / Fblk address the block's (0, 0) entry, ldw / ldf are the row strides.
__device__ __forceinline__ void potrf_inverse_block_scalar(const double* __restrict__ Wblk, int ldw, double* __restrict__ Fblk, int ldf,
                                                           int kb, double* __restrict__ uinv, int* __restrict__ not_pd,
                                                           double* __restrict__ lds) {
  const int lane = threadIdx.x & 63;
  double T[NB];
  {
    // branch-free: every lane loads from a clamped (always valid) address, padding is selected afterwards
    const int cc = min(lane, kb - 1);
    const double* __restrict__ base = Wblk + cc;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      const double v = base[size_t(min(r, kb - 1)) * ldw];
      const bool in = r < kb && lane < kb && lane >= r;
      T[r] = in ? v : ((r == lane) ? 1.0 : 0.0);
    }
  }
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const double d = readlane_f64(T[j], j);
    ok = ok && (d > 0.0);
    // 1 / sqrt(d) from the hardware estimate and two Newton steps (a correctly rounded sqrt and a
    // division cost ~45 instructions of this single wavefront's 32-step serial chain)
    // 1 / sqrt(d): hardware estimate and ONE Halley step (cubic) instead of two Newton steps (quadratic each):
    // half the dependent fp64 operations of the 32-step chain
    double rs = __builtin_amdgcn_rsq(d);
    {
      const double e = fma(-d * rs, rs, 1.0);            // 1 - d rs^2
      rs = fma(rs * e, fma(0.375, e, 0.5), rs);          // rs (1 + e / 2 + 3 e^2 / 8)
    }
    T[j] *= rs;  // the whole row; on the diagonal lane this is d * rs = sqrt(d) (no select: 32 lane masks kept in
                 // SGPRs next to the step's broadcasts made the compiler spill ~22 SGPRs per step)
    if (lane == 0) lds[NB * NB + j] = rs;  // 1 / U(j, j) for the inverse below
    const double uj = T[j];
    // entries below the diagonal (lane < i) are updated too; they are never read.  Four rows at a time: four
    // broadcasts into distinct SGPR pairs, then four FMAs (back to back on one SGPR pair every FMA waits two
    // cycles for its v_readlane), then the four results are pinned -- an empty asm on the updated registers: left
    // alone the optimiser sinks all updates of row i down to step i (a left-looking factorisation), which keeps every
    // broadcast U(j, i) alive until then: 700 SGPR spills to VGPR lanes
#pragma unroll
    for (int i = j + 1; i < NB; i += 4) {
      double bc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bc[q] = (i + q < NB) ? readlane_f64(T[j], (i + q < NB) ? i + q : NB - 1) : 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (i + q < NB) T[i + q] -= bc[q] * uj;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (i + q < NB) asm volatile("" : "+v"(T[i + q]));
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the broadcasts of step j + 1 out of step j (SGPR pressure)
  }
  if (!ok && lane == 0) *not_pd = 1;
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    if (lane < NB) lds[r * NB + lane] = (lane >= r) ? T[r] : 0.0;  // U(r, lane)
  }
  potrf_tail(Fblk, ldf, kb, uinv, lds);
}

// ---------------------------------------------------------------------------------------------------------------
// The same routine with the factorisation on the matrix cores.  The block lives in the C layout of
// v_mfma_f64_16x16x4_f64 -- tile (a, b) register g of lane l is element (16 a + 4 g + (l >> 4), 16 b + (l & 15)) -- and is
// factored right-looking in groups of FOUR columns, four being the K of that instruction:
//   A  the 4 x 4 diagonal block D of the group (10 broadcasts), U_D = chol(D) and V = U_D^-1 in every lane;
//   B  the group's rows of the factor, X = U_D^-T A(rows, :) = V' A(rows, :), as ONE product per column tile: the pivot
//      rows ARE register g0 of the tiles (a0, b) in B-operand layout, the A operand holds V' in its first four rows;
//   C  the trailing update A(r, c) -= sum_k X(k, r) X(k, c): X in that same register is both the A operand (rows of the
//      group, columns of tile a -- masked to the rows behind the group) and the B operand: one product per tile.
// No cross-lane data movement besides the ten broadcasts of step A; 25 matrix instructions and ~700 vector
// instructions replace the 496 broadcast + FMA pairs (~1 500 instructions, 20 k cycles) of potrf_inverse_block.
__device__ __forceinline__ double rsqrt_halley(double d) {
  const double rs = __builtin_amdgcn_rsq(d);
  const double e = fma(-d * rs, rs, 1.0);      // 1 - d rs^2
  return fma(rs * e, fma(0.375, e, 0.5), rs);  // rs (1 + e / 2 + 3 e^2 / 8)
}

template <int J0>
__device__ __forceinline__ void potrf_group4(double4_t (&T)[2][2], bool& ok, double* __restrict__ lds_rs) {
  constexpr int A0 = J0 / 16, G0 = (J0 % 16) / 4, C0 = J0 % 16;
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  // A: D(r, c), r <= c, sits in tile (A0, A0), register G0, lane 16 r + C0 + c
  const double src = T[A0][A0][G0];
  double u[4][4], rs[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double p = readlane_f64(src, 16 * k + C0 + k);
#pragma unroll
    for (int m = 0; m < k; ++m) p = fma(-u[m][k], u[m][k], p);
    ok = ok && (p > 0.0);
    rs[k] = rsqrt_halley(p);
    u[k][k] = p * rs[k];
#pragma unroll
    for (int c = k + 1; c < 4; ++c) {
      double t = readlane_f64(src, 16 * k + C0 + c);
#pragma unroll
      for (int m = 0; m < k; ++m) t = fma(-u[m][k], u[m][c], t);
      u[k][c] = t * rs[k];
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) lds_rs[J0 + k] = rs[k];
  }
  // V = U_D^-1 (upper)
  double v[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k][k] = rs[k];
#pragma unroll
  for (int j = 1; j < 4; ++j)
#pragma unroll
    for (int i = j - 1; i >= 0; --i) {
      double t = 0.0;
#pragma unroll
      for (int k = i + 1; k <= j; ++k) t = fma(u[i][k], v[k][j], t);
      v[i][j] = -t * rs[i];
    }
  // B: A operand of X = V' A(rows, :): lane (l & 15 = i, l >> 4 = k) supplies V(k, i) for k <= i < 4, else 0
  double aop = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = k; i < 4; ++i) aop = (lk == k && li == i) ? v[k][i] : aop;
  double x[2] = {0.0, 0.0};
#pragma unroll
  for (int b = A0; b < 2; ++b) {
    double4_t z = double4_t{0.0, 0.0, 0.0, 0.0};
    z = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, T[A0][b][G0], z, 0, 0, 0);
    x[b] = z[0];          // rows 0 .. 3 of the product: register 0, lane (k, c)
    T[A0][b][G0] = z[0];  // the group's rows of the factor
  }
  // C: trailing update; rows up to the group's last one must see a zero A operand
#pragma unroll
  for (int a = A0; a < 2; ++a) {
    if (a == A0 && C0 == 12) continue;  // the group closes its tile row
    const double xa = (a == A0) ? ((li > C0 + 3) ? -x[a] : 0.0) : -x[a];
#pragma unroll
    for (int b = a; b < 2; ++b) T[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, x[b], T[a][b], 0, 0, 0);
  }
}

// the eight groups, then U (zeros below the diagonal) into LDS, row-major, for potrf_tail
__device__ __forceinline__ void potrf_core_mfma(double4_t (&T)[2][2], int* __restrict__ not_pd, double* __restrict__ lds) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  bool ok = true;
  double* lds_rs = lds + NB * NB;
  potrf_group4<0>(T, ok, lds_rs);
  potrf_group4<4>(T, ok, lds_rs);
  potrf_group4<8>(T, ok, lds_rs);
  potrf_group4<12>(T, ok, lds_rs);
  potrf_group4<16>(T, ok, lds_rs);
  potrf_group4<20>(T, ok, lds_rs);
  potrf_group4<24>(T, ok, lds_rs);
  potrf_group4<28>(T, ok, lds_rs);
  if (!ok && lane == 0) *not_pd = 1;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = 16 * a + 4 * g + lk, c = 16 * b + li;
        lds[r * NB + c] = (b >= a && c >= r) ? T[a][b][g] : 0.0;
      }
}

// The block is handed over IN REGISTERS (C layout, tile (1, 0) ignored; entries outside the kb x kb block and below the
// diagonal must hold the padding: unit diagonal, zeros) -- for a caller that has just computed it (the look-ahead of a
// step kernel: no store, fence and reload between the trailing update and the next factorisation).
template <typename TF>
__device__ __forceinline__ void potrf_inverse_regs(double4_t (&T)[2][2], TF* __restrict__ Fblk, int ldf, int kb,
                                                   double* __restrict__ uinv, int* __restrict__ not_pd, double* __restrict__ lds,
                                                   double* __restrict__ uinv_lds = nullptr) {
  potrf_core_mfma(T, not_pd, lds);
  potrf_tail(Fblk, ldf, kb, uinv, lds, uinv_lds);
}

__device__ __forceinline__ void potrf_inverse_block_mfma(const double* __restrict__ Wblk, int ldw, double* __restrict__ Fblk, int ldf,
                                                         int kb, double* __restrict__ uinv, int* __restrict__ not_pd,
                                                         double* __restrict__ lds) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  double4_t T[2][2];  // tile (1, 0) is never touched
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = a; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = 16 * a + 4 * g + lk, c = 16 * b + li;
        // branch-free: a clamped (always valid) address, padding (unit diagonal) selected afterwards
        const double val = Wblk[size_t(min(r, kb - 1)) * ldw + min(c, kb - 1)];
        const bool in = r < kb && c < kb && c >= r;
        T[a][b][g] = in ? val : ((r == c) ? 1.0 : 0.0);
      }
  T[1][0] = double4_t{0.0, 0.0, 0.0, 0.0};
  potrf_inverse_regs(T, Fblk, ldf, kb, uinv, not_pd, lds);
}

// The routine the factorisations call: the matrix-core variant unless the library is built with -DCX_POTRF_SCALAR (A/B).
__device__ __forceinline__ void potrf_inverse_block(const double* __restrict__ Wblk, int ldw, double* __restrict__ Fblk, int ldf,
                                                    int kb, double* __restrict__ uinv, int* __restrict__ not_pd,
                                                    double* __restrict__ lds) {
#ifdef CX_POTRF_SCALAR
  potrf_inverse_block_scalar(Wblk, ldw, Fblk, ldf, kb, uinv, not_pd, lds);
#else
  potrf_inverse_block_mfma(Wblk, ldw, Fblk, ldf, kb, uinv, not_pd, lds);
#endif
}

// X = U_kk^-T W(k-rows, c0 .. c0 + 32) by MFMA: X[m][c] = sum_r Uinv[r][m] W[k0 + r][c], m, r < NB.
// Pointer form: Wrow addresses W(k0, c0); local columns >= ncols and rows >= kb read as zero.
// Result tile (mt, nt) register g of lane l is X[16 mt + (l >> 4) + 4 g][c0 + 16 nt + (l & 15)] -- which is
// exactly the operand layout of the trailing update (K index m = 4 (4 mt + g) + (l >> 4)), so X feeds
// the next MFMA without leaving the registers.
__device__ __forceinline__ void panel_x(const double* __restrict__ Wrow, int ldw, const double* __restrict__ uinv, int kb,
                                        int ncols, double4_t (&X)[2][2]) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  double bop[8][2], aop[8][2];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int r = 4 * s + lk;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int c = 16 * nt + li;
      bop[s][nt] = (r < kb && c < ncols) ? Wrow[size_t(r) * ldw + c] : 0.0;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) aop[s][mt] = uinv[r * NB + 16 * mt + li];  // A(m, r) = Uinv[r][m]
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) X[mt][nt] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) X[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[s][mt], bop[s][nt], X[mt][nt], 0, 0, 0);
}

}  // namespace cxchol

#endif
