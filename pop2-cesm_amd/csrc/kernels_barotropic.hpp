// kernels_barotropic.hpp -- HIP kernels of the barotropic driver, the PCG / ChronGear solvers,
// halo gather/scatter and the deterministic (b4b-style) global reductions.
//
// All 2-D work is tiny compared with the 3-D phases (O(1/km)); what matters here is launch
// count and latency, so the solver fuses every vector update with the matvec and the
// per-workgroup partial dot products, keeps all scalars on the device, and only returns to
// the host every convergenceCheckFreq iterations.
#pragma once
#include "kernels_common.hpp"

namespace pop {

#define POP_RED_THREADS 256

// ---- barotropic.F90:417-571: RHS assembly, part 1 (pointwise, whole array) ---------------
struct BtropArgs {
  const double *ZX, *ZY, *GXC, *GXO, *GYC, *GYO, *UBO, *VBO, *PCUR, *FW, *PGUESS;
  double *UH, *VH, *W3, *W4, *RHS, *centerWgt, *PNEW, *GXN, *GYN, *UBN, *VBN;
  const double *GXR, *GYR;    // reference gradient for the velocity update (old on leapfrog, cur otherwise)
  const double *scal;         // device scalars: [0] = xcheck
  double rcheck, rconst;
};
__global__ void k_btrop_rhs1(DevGrid g, StepParams sp, BtropArgs a) {
  Col c;
  if (!col_setup(g, c, false)) return;
  const long long q = c.q2;
  double W3, W4;
  if (sp.leapfrogts) {
    W3 = sp.c2dtp * (a.ZX[q] - sp.gamma * a.GXC[q] - (1.0 - sp.gamma) * a.GXO[q]);
    W4 = sp.c2dtp * (a.ZY[q] - sp.gamma * a.GYC[q] - (1.0 - sp.gamma) * a.GYO[q]);
  } else {
    W3 = sp.c2dtp * (a.ZX[q] - a.GXC[q]);
    W4 = sp.c2dtp * (a.ZY[q] - a.GYC[q]);
  }
  double uh, vh;
  if (sp.impcor) {
    const double W1 = sp.c2dtp * sp.beta * g.FCOR[q];
    const double W2 = 1.0 / (1.0 + W1 * W1);
    uh = W2 * (W3 + W1 * W4) + a.UBO[q];
    vh = W2 * (W4 - W1 * W3) + a.VBO[q];
  } else { uh = W3 + a.UBO[q]; vh = W4 + a.VBO[q]; }
  a.UH[q] = uh; a.VH[q] = vh;
  const double gx = sp.leapfrogts ? a.GXO[q] : a.GXC[q], gy = sp.leapfrogts ? a.GYO[q] : a.GYC[q];
  a.W3[q] = g.HU[q] * (uh + sp.beta * sp.c2dtp * gx);
  a.W4[q] = g.HU[q] * (vh + sp.beta * sp.c2dtp * gy);
}
// part 2: div (operators.F90:101-113), diagonal term, operator centre weight, initial guess
__global__ void k_btrop_rhs2(DevGrid g, StepParams sp, BtropArgs a) {
  Col c;
  if (!col_setup(g, c, false)) return;
  const long long q = c.q2;
  const int nxb = g.nxb;
  double d = 0.0;
  if (c.i >= 1 && c.j >= 1 && 1 <= g.KMT[q])
    d = 0.5 * (a.W3[q] * g.DYU[q] + a.W3[q - nxb] * g.DYU[q - nxb] - a.W3[q - 1] * g.DYU[q - 1] - a.W3[q - 1 - nxb] * g.DYU[q - 1 - nxb] +
               a.W4[q] * g.DXU[q] + a.W4[q - 1] * g.DXU[q - 1] - a.W4[q - nxb] * g.DXU[q - nxb] - a.W4[q - 1 - nxb] * g.DXU[q - 1 - nxb]);
  double rhs = d / (sp.beta * sp.c2dtp);
  const double dc = (g.KMT[q] >= 1) ? g.TAREA[q] / (sp.beta * sp.c2dtp * sp.dtp * sp.grav) : 0.0;
  rhs = rhs - dc * a.PCUR[q] - a.FW[q] * g.TAREA[q] / (sp.beta * sp.c2dtp);
  a.RHS[q] = rhs;
  a.centerWgt[q] = g.WC0[q] - dc;
  a.PNEW[q] = a.PGUESS[q];
}
// barotropic.F90:622-679: null-space removal (whole array)
__global__ void k_btrop_fin1(DevGrid g, BtropArgs a) {
  Col c;
  if (!col_setup(g, c, false)) return;
  const long long q = c.q2;
  const double xcheck = a.scal[0];
  a.PNEW[q] = a.PNEW[q] + g.CONSTNT[q] * a.rcheck * xcheck - g.CHECKER[q] * a.rconst * xcheck;
}
// grad of the new surface pressure (operators.F90:178-187) and new barotropic velocity
__global__ void k_btrop_fin2(DevGrid g, StepParams sp, BtropArgs a) {
  Col c;
  if (!col_setup(g, c, false)) return;
  const long long q = c.q2;
  const int nxb = g.nxb;
  double gx = 0.0, gy = 0.0;
  if (c.i < g.nxb - 1 && c.j < g.nyb - 1 && 1 <= g.KMU[q]) {
    const double f00 = a.PNEW[q], f10 = a.PNEW[q + 1], f01 = a.PNEW[q + nxb], f11 = a.PNEW[q + nxb + 1];
    gx = g.DXUR[q] * 0.5 * (f11 - f00 - f01 + f10);
    gy = g.DYUR[q] * 0.5 * (f11 - f00 + f01 - f10);
  }
  a.GXN[q] = gx; a.GYN[q] = gy;
  a.UBN[q] = a.UH[q] - sp.beta * sp.c2dtp * (gx - a.GXR[q]);
  a.VBN[q] = a.VH[q] - sp.beta * sp.c2dtp * (gy - a.GYR[q]);
}

// ---- operators.F90 as stand-alone entry points (pop_operator): grad :126-192, div :49-119, zcurl :199-272 on one
// horizontal slab at level k (the time step itself has them inlined in its kernels)
__global__ void k_operator(DevGrid g, int op, int k, const double *__restrict__ A, const double *__restrict__ Bf,
                           double *__restrict__ O1, double *__restrict__ O2, long long blk_stride_in, long long blk_stride_out, int b0) {
  // b0: first block of the launch (pop_operator_host works on one block: grid y = 1, strides 0, b0 = that block)
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y + b0;
  if (p2 >= g.n2) return;
  const int i = p2 % g.nxb, j = p2 / g.nxb, nxb = g.nxb;
  const long long q = (long long)b * g.n2 + p2, qi = (long long)b * blk_stride_in + p2, qo = (long long)b * blk_stride_out + p2;
  if (op == 0) {
    double gx = 0.0, gy = 0.0;
    if (i < nxb - 1 && j < g.nyb - 1 && k <= g.KMU[q]) {
      const double f00 = A[qi], f10 = A[qi + 1], f01 = A[qi + nxb], f11 = A[qi + nxb + 1];
      gx = g.DXUR[q] * 0.5 * (f11 - f00 - f01 + f10);
      gy = g.DYUR[q] * 0.5 * (f11 - f00 + f01 - f10);
    }
    O1[qo] = gx; O2[qo] = gy;
    return;
  }
  double r = 0.0;
  if (i >= 1 && j >= 1 && k <= g.KMT[q]) {
    const double dy00 = g.DYU[q], dy0m = g.DYU[q - nxb], dym0 = g.DYU[q - 1], dymm = g.DYU[q - 1 - nxb];
    const double dx00 = g.DXU[q], dx0m = g.DXU[q - nxb], dxm0 = g.DXU[q - 1], dxmm = g.DXU[q - 1 - nxb];
    if (op == 1)
      r = 0.5 * (A[qi] * dy00 + A[qi - nxb] * dy0m - A[qi - 1] * dym0 - A[qi - 1 - nxb] * dymm +
                 Bf[qi] * dx00 + Bf[qi - 1] * dxm0 - Bf[qi - nxb] * dx0m - Bf[qi - 1 - nxb] * dxmm);
    else
      r = 0.5 * (Bf[qi] * dy00 + Bf[qi - nxb] * dy0m - Bf[qi - 1] * dym0 - Bf[qi - 1 - nxb] * dymm -
                 A[qi] * dx00 - A[qi - 1] * dxm0 + A[qi - nxb] * dx0m + A[qi - 1 - nxb] * dxmm);
  }
  O1[qo] = r;
}

// ---- deterministic reductions --------------------------------------------------------------
// Stage 1 is fused into the producing kernels: each workgroup reduces its 256 cells with a
// fixed LDS tree and writes partial[(blk*nchunk + chunk)*nfields + f].  Stage 2 (one workgroup)
// adds the partials of every POP block in fixed order -> block sums (the b4b block-sum vector of
// mpi/POP_ReductionsMod.F90:348-383), then the block sums in global block-id order.
// levels 32 .. 1 of the fixed tree inside the first wavefront (r3): lane t < s adds the value of lane t + s -- the operands and the
// order of sh[t] = sh[t] + sh[t + s] -- through the cross-lane network instead of six LDS round trips with a barrier each
__device__ __forceinline__ double tree_tail64(double x) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) x = x + __shfl_down(x, s, 64);
  return x;
}
template <int NF>
__device__ __forceinline__ void wg_reduce_store(double (&v)[NF], double *partial, int slot) {
  __shared__ double sh[NF][POP_RED_THREADS];
  const int t = threadIdx.x;
#pragma unroll
  for (int f = 0; f < NF; ++f) sh[f][t] = v[f];
  __syncthreads();
  // the fixed tree (LDS steps 128, 64, then the shuffles of tree_tail64) without a barrier inside it (round 4): lane l of ONE wave forms what the two
  // LDS steps leave in element l, (v[l] + v[l+128]) + (v[l+64] + v[l+192]) -- the same additions in the same order
  if (t < 64) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const double x = tree_tail64((sh[f][t] + sh[f][t + 128]) + (sh[f][t + 64] + sh[f][t + 192]));
      if (t == 0) partial[(long long)slot * NF + f] = x;
    }
  }
}
// Stage 2: blocksum[gid[b]*NF+f] = ordered sum of partials of local block b.  One workgroup per
// (local block); threads take strided subsets sequentially, then a fixed tree.
// ordered sum of the nchunk partials of one block (all NF fields) by one workgroup: thread-strided
// left-to-right sums, then a fixed tree; result valid in thread 0
template <int NF, int DEPTH = 16>
__device__ __forceinline__ void block_sum_ordered(const double *__restrict__ pb, int nchunk, double (&out)[NF]) {
  __shared__ double sh[NF][POP_RED_THREADS];
  const int t = threadIdx.x;
  double v[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) v[f] = 0.0;
  // same left-to-right order as a plain strided loop; the loads of DEPTH terms are issued before their
  // adds so the chain costs one memory latency per DEPTH terms instead of one per term (16 inside the solver kernels that
  // recompute the total themselves; 64 in the one-workgroup k_block_sums launches of large grids, whose 130 terms per thread
  // then take three round trips instead of nine)
  int cidx = t;
  for (; cidx + (DEPTH - 1) * POP_RED_THREADS < nchunk; cidx += DEPTH * POP_RED_THREADS) {
    double w[DEPTH][NF];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
#pragma unroll
      for (int f = 0; f < NF; ++f) w[u][f] = pb[((long long)cidx + u * POP_RED_THREADS) * NF + f];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
#pragma unroll
      for (int f = 0; f < NF; ++f) v[f] = v[f] + w[u][f];
  }
  if (DEPTH > 16) {
    // the remainder (fewer than DEPTH terms; r4): full batches of 16 while there are that many, then ONE batch of 8 in which every load is
    // requested -- at a clamped address where the term does not exist -- before the first add, and a term that does not exist is not added
    // (adding +0.0 could turn a sum of -0.0 into +0.0); repeated only if more than 8 terms are left.  Until round 4 the last terms went one
    // load -> one add at a time (tx0.1v3: 33 844 chunks, 132 terms per thread = four dependent round trips behind the two batches of 64); a
    // predicated batch of all 64 was tried first and cost more than it saved (11.2 against 9.6 us per launch inside the step).
    for (; cidx + 15 * POP_RED_THREADS < nchunk; cidx += 16 * POP_RED_THREADS) {
      double w[16][NF];
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int f = 0; f < NF; ++f) w[u][f] = pb[((long long)cidx + u * POP_RED_THREADS) * NF + f];
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int f = 0; f < NF; ++f) v[f] = v[f] + w[u][f];
    }
    for (; cidx < nchunk; cidx += 8 * POP_RED_THREADS) {
      double w[8][NF];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long c = (long long)cidx + u * POP_RED_THREADS;
#pragma unroll
        for (int f = 0; f < NF; ++f) w[u][f] = pb[(c < nchunk ? c : (long long)nchunk - 1) * NF + f];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool there = (long long)cidx + u * POP_RED_THREADS < nchunk;
#pragma unroll
        for (int f = 0; f < NF; ++f) v[f] = there ? v[f] + w[u][f] : v[f];
      }
    }
  } else {
    for (; cidx < nchunk; cidx += POP_RED_THREADS)
#pragma unroll
      for (int f = 0; f < NF; ++f) v[f] = v[f] + pb[(long long)cidx * NF + f];
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) sh[f][t] = v[f];
  __syncthreads();
  // out is used by thread 0 only (every caller stores from threadIdx.x == 0); the tree without a barrier inside it (see wg_reduce_store)
#pragma unroll
  for (int f = 0; f < NF; ++f) out[f] = (t < 64) ? tree_tail64((sh[f][t] + sh[f][t + 128]) + (sh[f][t + 64] + sh[f][t + 192])) : 0.0;
}
template <int NF>
__global__ void __launch_bounds__(POP_RED_THREADS) k_block_sums(const double *__restrict__ partial, int nchunk, const int *__restrict__ gid,
                             double *__restrict__ blocksum) {
  const int b = blockIdx.x;
  double r[NF];
  block_sum_ordered<NF, (NF == 1 ? 64 : 32)>(partial + (long long)b * nchunk * NF, nchunk, r);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int f = 0; f < NF; ++f) blocksum[(long long)gid[b] * NF + f] = r[f];
  }
}
// The same sum by 1024 threads (large grids, r4).  The order of block_sum_ordered -- 256 accumulators, accumulator a adds the partials
// a, a + 256, a + 512, ... strictly left to right, then the fixed tree over the accumulators -- bounds a 256-thread workgroup by the 63
// loads a wave can have in flight: 132 terms per thread (tx0.1v3) are three memory round trips.  Here every accumulator is shared by
// FOUR threads, thread (a, h) holding the h-th quarter of its terms: all 1024 threads request their <= LMAX terms at once (ONE round
// trip), then the quarters are added in turn -- thread (a, 0) from +0.0, thread (a, h) from what thread (a, h - 1) left in LDS.  The
// same additions in the same order; a term that does not exist is not added.
constexpr int POP_RELAY_SLACK = 4096;
template <int NF, int LMAX>
__global__ void __launch_bounds__(1024) k_block_sums_relay(const double *__restrict__ partial, int nchunk, const int *__restrict__ gid,
                                                            double *__restrict__ blocksum) {
  __shared__ double relay[POP_RED_THREADS];
  const int b = blockIdx.x, f = blockIdx.y, T = threadIdx.x, a = T & (POP_RED_THREADS - 1), h = T >> 8;   // one workgroup per (block, field)
  const double *__restrict__ pb = partial + (long long)b * nchunk * NF + f;
  const int terms = (nchunk + POP_RED_THREADS - 1) / POP_RED_THREADS, L = (terms + 3) / 4;   // L <= LMAX (host)
  // term u of this thread: partial a + 256 (h L + u) = (a wave-uniform base that advances with u) + (one offset per thread): a scalar
  // base and ONE address register for all the loads.  Unpredicated: the host allocates POP_RELAY_SLACK doubles behind the partials, a
  // value read from there is not added.
  const unsigned toff = ((unsigned)a + (unsigned)POP_RED_THREADS * (unsigned)(h * L)) * NF;
  double w[LMAX];
#pragma unroll
  for (int u = 0; u < LMAX; ++u) {
    const double *__restrict__ pu = pb + (size_t)POP_RED_THREADS * u * NF;
    w[u] = (u < L) ? pu[toff] : 0.0;   // (uniform condition)
  }
#pragma unroll 1
  for (int hh = 0; hh < 4; ++hh) {
    if (h == hh) {
      double v = (hh == 0) ? 0.0 : relay[a];
#pragma unroll
      for (int u = 0; u < LMAX; ++u) {
        const bool there = u < L && (long long)a + (long long)POP_RED_THREADS * (h * L + u) < nchunk;
        v = there ? v + w[u] : v;
      }
      relay[a] = v;
    }
    __syncthreads();
  }
  if (T < 64) {
    const double x = tree_tail64((relay[T] + relay[T + 128]) + (relay[T + 64] + relay[T + 192]));
    if (T == 0) blocksum[(long long)gid[b] * NF + f] = x;
  }
}
// the whole block-sum vector of the decomposition in one launch (grid = nblocks_tot): own blocks get their
// ordered sum, the others 0, ready for the all-reduce (replaces memset + k_block_sums)
template <int NF>
__global__ void __launch_bounds__(POP_RED_THREADS) k_block_sums_global(const double *__restrict__ partial, int nchunk, const int *__restrict__ local_of_gid,
                                    double *__restrict__ blocksum) {
  const int bg = blockIdx.x, lb = local_of_gid[bg];
  double r[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) r[f] = 0.0;
  if (lb >= 0) block_sum_ordered<NF, (NF == 1 ? 64 : 32)>(partial + (long long)lb * nchunk * NF, nchunk, r);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int f = 0; f < NF; ++f) blocksum[(long long)bg * NF + f] = r[f];
  }
}

// Solver scalars kept on the device
struct SolverScalars {
  double eta0, eta1, alpha, beta_cg, rr, sum0, sum1, xcheck;
  double rho_old, sigma;   // ChronGear
  double rho2[2], sigma2[2];   // fused ChronGear: (rho_old, sigma) ping-pong between iterations (read [par], write [1-par])
  // fused solvers with one check interval of look-ahead: `stop` is raised by the check that meets the convergence
  // criterion and turns the kernels of the interval already enqueued behind it into no-ops; `icnt` counts the checks
  int stop, icnt;
};
enum { FIN_PCG_RZ = 1, FIN_PCG_SQ = 2, FIN_RR = 3, FIN_XCHECK = 4, FIN_CG_INIT = 5, FIN_CG_ITER = 6, FIN_PLAIN = 7, FIN_TRIPOLE = 8 };
// Stage 3: global sum over the block-sum vector in block-id order + scalar recurrences
// (POP_SolversMod.F90:1365-1420 for pcg, :2159-2170 for ChronGear).  Single thread.
template <int NF>
__global__ void k_finalize(const double *__restrict__ blocksum, int nblocks_tot, SolverScalars *s, int mode) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double g[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) g[f] = 0.0;
  for (int b = 0; b < nblocks_tot; ++b)
#pragma unroll
    for (int f = 0; f < NF; ++f) g[f] = g[f] + blocksum[(long long)b * NF + f];
  s->sum0 = g[0];
  if (NF > 1) s->sum1 = g[NF > 1 ? 1 : 0];
  if (mode == FIN_TRIPOLE && NF > 1) {   // field 1 = redundant top-row points: taken out of each block sum before the blocks are added
    double t = 0.0;
    for (int b = 0; b < nblocks_tot; ++b) t = t + (blocksum[(long long)b * NF] - blocksum[(long long)b * NF + 1]);
    s->sum0 = t;
  }
  switch (mode) {
    case FIN_PCG_RZ: s->eta1 = g[0]; s->beta_cg = s->eta1 / s->eta0; break;               // s = z + s*(eta1/eta0)
    case FIN_PCG_SQ: s->eta0 = s->eta1; s->eta1 = s->eta0 / g[0]; s->alpha = s->eta1; break;
    case FIN_RR: s->rr = g[0]; break;
    case FIN_XCHECK: s->xcheck = g[0]; break;
    case FIN_CG_INIT: s->rho_old = g[0]; s->sigma = g[NF > 1 ? 1 : 0]; s->alpha = s->rho_old / s->sigma;
      s->rho2[0] = s->rho_old; s->sigma2[0] = s->sigma; break;
    case FIN_CG_ITER: {
      const double rho = g[0], delta = g[NF > 1 ? 1 : 0];
      s->beta_cg = rho / s->rho_old;
      s->sigma = delta - (s->beta_cg * s->beta_cg) * s->sigma;
      s->alpha = rho / s->sigma;
      s->rho_old = rho;
    } break;
    default: break;
  }
}

// POP_GlobalMaxval / Minval / Maxloc / Minloc (mpi/POP_ReductionsMod.F90:2670-3223, 4002-4400): per workgroup the
// extreme value of the physical cells selected by the mask (non-zero = selected) and the smallest cell index that
// attains it; partial[2*slot] = value, partial[2*slot+1] = cell index within the rank (-1: none selected)
template <bool ISMAX>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_extreme_partial(DevGrid g, const double *__restrict__ A, const double *__restrict__ M, double *__restrict__ partial) {
  __shared__ double sv[POP_RED_THREADS], si[POP_RED_THREADS];
  const int p2 = red_cell(g), b = blockIdx.y, t = threadIdx.x;
  double v = 0.0, idx = -1.0;
  if (p2 < g.n2) {
    const int i = p2 % g.nxb, j = p2 / g.nxb;
    const long long q = (long long)b * g.n2 + p2;
    if (i + 1 >= g.ib && i + 1 <= blk_ie(g, b) && j + 1 >= g.jb && j + 1 <= blk_je(g, b) && (!M || M[q] != 0.0)) { v = A[q]; idx = (double)q; }
  }
  sv[t] = v; si[t] = idx;
  __syncthreads();
  for (int s = POP_RED_THREADS / 2; s > 0; s >>= 1) {
    if (t < s) {
      const double v2 = sv[t + s], i2 = si[t + s];
      const bool take = i2 >= 0.0 && (si[t] < 0.0 || (ISMAX ? v2 > sv[t] : v2 < sv[t]) || (v2 == sv[t] && i2 < si[t]));
      if (take) { sv[t] = v2; si[t] = i2; }
    }
    __syncthreads();
  }
  if (t == 0) { const long long slot = (long long)b * gridDim.x + red_chunk(g); partial[2 * slot] = sv[0]; partial[2 * slot + 1] = si[0]; }
}
// POP_GlobalCount (:2062-2207): number of non-zero physical cells, as a sum of exact ones
__global__ void __launch_bounds__(POP_RED_THREADS)
k_count_partial(DevGrid g, const double *__restrict__ A, const double *__restrict__ DUP, double *__restrict__ partial) {
  const int p2 = red_cell(g), b = blockIdx.y;
  double v[1] = {0.0};
  if (p2 < g.n2) {
    const int i = p2 % g.nxb, j = p2 / g.nxb;
    const long long q = (long long)b * g.n2 + p2;
    if (i + 1 >= g.ib && i + 1 <= blk_ie(g, b) && j + 1 >= g.jb && j + 1 <= blk_je(g, b) && A[q] != 0.0 && !(DUP && DUP[q] != 0.0)) v[0] = 1.0;
  }
  wg_reduce_store<1>(v, partial, b * gridDim.x + red_chunk(g));
}

// partial of (a*mask, a*mask*dup) -- tripole global sums of N-face / NE-corner fields
__global__ void __launch_bounds__(POP_RED_THREADS)
k_dot_partial_dup(DevGrid g, const double *__restrict__ A, const double *__restrict__ M, const double *__restrict__ DUP,
                  double *__restrict__ partial) {
  const int p2 = red_cell(g), b = blockIdx.y;
  double v[2] = {0.0, 0.0};
  if (p2 < g.n2) {
    const int i = p2 % g.nxb, j = p2 / g.nxb;
    if (i + 1 >= g.ib && i + 1 <= blk_ie(g, b) && j + 1 >= g.jb && j + 1 <= blk_je(g, b)) {
      const long long q = (long long)b * g.n2 + p2;
      double x = A[q];
      if (M) x = x * M[q];
      v[0] = x; v[1] = x * DUP[q];
    }
  }
  wg_reduce_store<2>(v, partial, b * gridDim.x + red_chunk(g));
}

// Convergence check of the fused solvers in ONE launch: ordered block sums of the (r,r) partials (blocks of the
// view in block-id order, same rule as k_block_sums + k_finalize), result to the device scalars and straight
// into pinned host memory -- replaces three stream operations (block sums, finalize, copy) by one.
__global__ void __launch_bounds__(POP_RED_THREADS)
k_rr_total(const double *__restrict__ partial, int nchunk, int nblocks, SolverScalars *s, double *host_ring, double criterion) {
  if (s->stop) return;
  double total = 0.0;
  for (int b = 0; b < nblocks; ++b) {
    double r[1];
    block_sum_ordered<1>(partial + (long long)b * nchunk, nchunk, r);
    total = total + r[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) {   // result of check number icnt into the pinned ring the host reads after the interval's event
    s->sum0 = total; s->rr = total;
    host_ring[s->icnt & 7] = total;
    s->icnt = s->icnt + 1;
    if (total < criterion) s->stop = 1;
  }
}

// the same check for blocks spread over ranks: the all-reduced block-sum vector added in block-id order
__global__ void k_rr_blocks(const double *__restrict__ blocksum, int nblocks_tot, SolverScalars *s, double *host_ring, double criterion) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || s->stop) return;
  double total = 0.0;
  for (int b = 0; b < nblocks_tot; ++b) total = total + blocksum[b];
  s->sum0 = total; s->rr = total;
  host_ring[s->icnt & 7] = total;
  s->icnt = s->icnt + 1;
  if (total < criterion) s->stop = 1;
}

// generic masked product sum over the physical domain: partial of a*b*mask (b, mask optional)
__global__ void __launch_bounds__(POP_RED_THREADS)
k_dot_partial(DevGrid g, const double *__restrict__ A, const double *__restrict__ Bv, const double *__restrict__ M,
              double *__restrict__ partial) {
  const int p2 = red_cell(g), b = blockIdx.y;
  double v[1] = {0.0};
  if (p2 < g.n2) {
    const int i = p2 % g.nxb, j = p2 / g.nxb;
    if (i + 1 >= g.ib && i + 1 <= blk_ie(g, b) && j + 1 >= g.jb && j + 1 <= blk_je(g, b)) {
      const long long q = (long long)b * g.n2 + p2;
      double x = A[q];
      if (Bv) x = x * Bv[q];
      if (M) x = x * M[q];
      v[0] = x;
    }
  }
  wg_reduce_store<1>(v, partial, b * gridDim.x + red_chunk(g));
}

// POP_SolversDiagonal (POP_SolversMod.F90:1110-1151): centre weight of one block = time-independent part - correction
__global__ void k_solver_diagonal(const double *__restrict__ WC0, const double *__restrict__ corr, double *__restrict__ C, int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) C[p] = WC0[p] - corr[p];
}

// ---- btropOperator (POP_SolversMod.F90:2414-2426) at one point -----------------------------
__device__ __forceinline__ double btrop_op(const DevGrid &g, const double *__restrict__ C, const double *__restrict__ X,
                                           long long q, int nxb) {
  return C[q] * X[q] + g.WNo[q] * X[q + nxb] + g.WNo[q - nxb] * X[q - nxb] + g.WEa[q] * X[q + 1] + g.WEa[q - 1] * X[q - 1] +
         g.WNE[q] * X[q + nxb + 1] + g.WNE[q - nxb] * X[q - nxb + 1] + g.WNE[q - 1] * X[q + nxb - 1] + g.WNE[q - 1 - nxb] * X[q - nxb - 1];
}
__device__ __forceinline__ bool op_range(const DevGrid &g, int i, int j) {   // i,j = 2..n-1 (1-based)
  return i >= 1 && i <= g.nxb - 2 && j >= 1 && j <= g.nyb - 2;
}
__device__ __forceinline__ bool interior(const DevGrid &g, int b, int i, int j) {
  return i + 1 >= g.ib && i + 1 <= blk_ie(g, b) && j + 1 >= g.jb && j + 1 <= blk_je(g, b);
}

struct SolverArgs {
  double *X, *R, *S0, *S1, *Q, *Z, *AZ;
  const double *Bv, *C;      // right-hand side, centre weight
  double *partial;
  SolverScalars *sc;
};
// r = b - A x  (whole array; :1273-1283).  WITH_RR: also partial (r,r) (:1452-1458, :2188-2199)
template <bool WITH_RR>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_residual(DevGrid g, SolverArgs a) {
  const int p2 = red_cell(g), b = blockIdx.y;
  double v[1] = {0.0};
  if (p2 < g.n2) {
    const int i = p2 % g.nxb, j = p2 / g.nxb;
    const long long q = (long long)b * g.n2 + p2;
    const double ax = op_range(g, i, j) ? btrop_op(g, a.C, a.X, q, g.nxb) : 0.0;
    const double r = a.Bv[q] - ax;
    a.R[q] = r;
    if (WITH_RR && interior(g, b, i, j)) v[0] = (r * r) * g.mMask[q];
  }
  if (WITH_RR) wg_reduce_store<1>(v, a.partial, b * gridDim.x + red_chunk(g));
}
// PCG step A (:1320-1345, :1433-1438 of the previous iteration): x += alpha s; r -= alpha q (if
// UPDATE); z = r/diag; partial (r,z)
template <bool UPDATE>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_pcg_a(DevGrid g, SolverArgs a) {
  const int p2 = red_cell(g), b = blockIdx.y;
  double v[1] = {0.0};
  if (p2 < g.n2) {
    const int i = p2 % g.nxb, j = p2 / g.nxb;
    const long long q = (long long)b * g.n2 + p2;
    double r = a.R[q];
    if (UPDATE) {
      const double al = a.sc->alpha;
      a.X[q] = a.X[q] + al * a.S0[q];
      r = r - al * a.Q[q];
      a.R[q] = r;
    }
    const double cw = a.C[q];
    const double z = (cw != 0.0) ? r / cw : 0.0;
    a.Z[q] = z;
    if (interior(g, b, i, j)) v[0] = (r * z) * g.mMask[q];
  }
  wg_reduce_store<1>(v, a.partial, b * gridDim.x + red_chunk(g));
}
// x,r update only (before a convergence check)
__global__ void k_pcg_xr(DevGrid g, SolverArgs a) {
  const int p2 = red_cell(g), b = blockIdx.y;
  if (p2 >= g.n2) return;
  const long long q = (long long)b * g.n2 + p2;
  const double al = a.sc->alpha;
  a.X[q] = a.X[q] + al * a.S0[q];
  a.R[q] = a.R[q] - al * a.Q[q];
}
// PCG step B (:1377-1399): s_new = z + s_old*(eta1/eta0) evaluated at the 9 stencil points from
// the old direction buffer (no race: S0 is read, S1 written); q = A s_new; partial (q,s)
__global__ void __launch_bounds__(POP_RED_THREADS)
k_pcg_b(DevGrid g, SolverArgs a) {
  const int p2 = red_cell(g), b = blockIdx.y;
  double v[1] = {0.0};
  if (p2 < g.n2) {
    const int i = p2 % g.nxb, j = p2 / g.nxb, nxb = g.nxb;
    const long long q = (long long)b * g.n2 + p2;
    const double bt = a.sc->beta_cg;
    auto sn = [&](long long qq) { return a.Z[qq] + a.S0[qq] * bt; };
    const double s = sn(q);
    a.S1[q] = s;
    double aq = 0.0;
    if (op_range(g, i, j)) {
      aq = a.C[q] * s + g.WNo[q] * sn(q + nxb) + g.WNo[q - nxb] * sn(q - nxb) + g.WEa[q] * sn(q + 1) + g.WEa[q - 1] * sn(q - 1) +
           g.WNE[q] * sn(q + nxb + 1) + g.WNE[q - nxb] * sn(q - nxb + 1) + g.WNE[q - 1] * sn(q + nxb - 1) + g.WNE[q - 1 - nxb] * sn(q - nxb - 1);
    }
    a.Q[q] = aq;
    if (interior(g, b, i, j)) v[0] = (aq * s) * g.mMask[q];
  }
  wg_reduce_store<1>(v, a.partial, b * gridDim.x + red_chunk(g));
}

// ---- ChronGear (POP_SolversMod.F90:2040-2210) -----------------------------------------------
// init: z = r*A0R (EXTZ: z already in a.Z, halo included -- EVP preconditioner); s = z; q = A s; partial (r,z), (s,q)
template <bool EXTZ>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_cg_init(DevGrid g, SolverArgs a) {
  const int p2 = red_cell(g), b = blockIdx.y;
  double v[2] = {0.0, 0.0};
  if (p2 < g.n2) {
    const int i = p2 % g.nxb, j = p2 / g.nxb, nxb = g.nxb;
    const long long q = (long long)b * g.n2 + p2;
    auto zf = [&](long long qq) { if (EXTZ) return a.Z[qq]; const double cw = a.C[qq]; return a.R[qq] * ((cw != 0.0) ? 1.0 / cw : 0.0); };
    const double z = zf(q);
    if (!EXTZ) a.Z[q] = z;
    a.S0[q] = z;
    double aq = 0.0;
    if (op_range(g, i, j))
      aq = a.C[q] * z + g.WNo[q] * zf(q + nxb) + g.WNo[q - nxb] * zf(q - nxb) + g.WEa[q] * zf(q + 1) + g.WEa[q - 1] * zf(q - 1) +
           g.WNE[q] * zf(q + nxb + 1) + g.WNE[q - nxb] * zf(q - nxb + 1) + g.WNE[q - 1] * zf(q + nxb - 1) + g.WNE[q - 1 - nxb] * zf(q - nxb - 1);
    a.Q[q] = aq;
    if (interior(g, b, i, j)) { v[0] = (a.R[q] * z) * g.mMask[q]; v[1] = (z * aq) * g.mMask[q]; }
  }
  wg_reduce_store<2>(v, a.partial, b * gridDim.x + red_chunk(g));
}
// z = r*A0R (whole array; the halo of Z follows)
__global__ void k_cg_z(DevGrid g, SolverArgs a) {
  const int p2 = red_cell(g), b = blockIdx.y;
  if (p2 >= g.n2) return;
  const long long q = (long long)b * g.n2 + p2;
  const double cw = a.C[q];
  a.Z[q] = a.R[q] * ((cw != 0.0) ? 1.0 / cw : 0.0);
}
// az = A z; partial (r,z), (az,z)
__global__ void __launch_bounds__(POP_RED_THREADS)
k_cg_az(DevGrid g, SolverArgs a) {
  const int p2 = red_cell(g), b = blockIdx.y;
  double v[2] = {0.0, 0.0};
  if (p2 < g.n2) {
    const int i = p2 % g.nxb, j = p2 / g.nxb;
    const long long q = (long long)b * g.n2 + p2;
    const double az = op_range(g, i, j) ? btrop_op(g, a.C, a.Z, q, g.nxb) : 0.0;
    a.AZ[q] = az;
    if (interior(g, b, i, j)) { const double z = a.Z[q]; v[0] = (a.R[q] * z) * g.mMask[q]; v[1] = (az * z) * g.mMask[q]; }
  }
  wg_reduce_store<2>(v, a.partial, b * gridDim.x + red_chunk(g));
}
// s = z + beta s; q = az + beta q; x += alpha s; r -= alpha q   (FIRST: x,r only with the init alpha)
template <bool FIRST>
__global__ void k_cg_update(DevGrid g, SolverArgs a) {
  const int p2 = red_cell(g), b = blockIdx.y;
  if (p2 >= g.n2) return;
  const long long q = (long long)b * g.n2 + p2;
  const double al = a.sc->alpha;
  double s = a.S0[q], qq = a.Q[q];
  if (!FIRST) {
    const double bt = a.sc->beta_cg;
    s = a.Z[q] + bt * s;
    qq = a.AZ[q] + bt * qq;
    a.S0[q] = s; a.Q[q] = qq;
  }
  a.X[q] = a.X[q] + al * s;
  a.R[q] = a.R[q] - al * qq;
}

// ---- halo update kernels (index lists from the plan) ---------------------------------------
// ghost <- interior copies between blocks of this rank, and fill values; one thread per
// (cell, level); field layout (n2 cells, nz levels, blocks): cell index = blk*n2 + p2
__global__ void k_halo_local(double *__restrict__ F, const int *__restrict__ dst, const int *__restrict__ src, int ncopy,
                             const int *__restrict__ filld, int nfill, double fill, int nz, int n2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (t < ncopy) {
    const int d = dst[t], s = src[t];
    F[((long long)(d / n2) * nz + k) * n2 + d % n2] = F[((long long)(s / n2) * nz + k) * n2 + s % n2];
  } else if (t < ncopy + nfill) {
    const int d = filld[t - ncopy];
    F[((long long)(d / n2) * nz + k) * n2 + d % n2] = fill;
  }
}
// pack cells into a message buffer laid out [level][cell]
__global__ void k_halo_pack(const double *__restrict__ F, const int *__restrict__ src, int n, double *__restrict__ buf, int nz, int n2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (t >= n) return;
  const int s = src[t];
  buf[(long long)k * n + t] = F[((long long)(s / n2) * nz + k) * n2 + s % n2];
}
__global__ void k_halo_unpack(double *__restrict__ F, const int *__restrict__ dst, int n, const double *__restrict__ buf, int nz, int n2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (t >= n) return;
  const int d = dst[t];
  F[((long long)(d / n2) * nz + k) * n2 + d % n2] = buf[(long long)k * n + t];
}

// tripole northern boundary (HaloPlan::tripole): phase 1 evaluates every entry into a buffer, phase 2 stores
__global__ void k_tripole_eval(const double *__restrict__ F, const int *__restrict__ A, const int *__restrict__ Bc, int n,
                               double *__restrict__ buf, double isign, int nz, int n2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (t >= n) return;
  const int ca = A[t], cb = Bc[t];
  const double x = F[((long long)(ca / n2) * nz + k) * n2 + ca % n2];
  double r;
  if (cb < 0) r = isign * x;
  else {
    const double y = F[((long long)(cb / n2) * nz + k) * n2 + cb % n2];
    const double m = 0.5 * (fabs(x) + fabs(y));
    r = (x < 0.0) ? -m : m;
  }
  buf[(long long)k * n + t] = r;
}
__global__ void k_tripole_store(double *__restrict__ F, const int *__restrict__ D, int n, const double *__restrict__ buf, int nz, int n2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (t >= n) return;
  const int d = D[t];
  F[((long long)(d / n2) * nz + k) * n2 + d % n2] = buf[(long long)k * n + t];
}

// all peers in one launch: element t of the concatenated list belongs to the message that starts at
// start[t] cells into the buffer and has cnt[t] cells per level (message = level-major, as above)
__global__ void k_halo_pack_all(const double *__restrict__ F, const int *__restrict__ src, const int *__restrict__ start,
                                const int *__restrict__ cnt, int n, double *__restrict__ buf, int nz, int n2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (t >= n) return;
  const int s = src[t], st = start[t], m = cnt[t];
  buf[(long long)st * nz + (long long)k * m + (t - st)] = F[((long long)(s / n2) * nz + k) * n2 + s % n2];
}
__global__ void k_halo_unpack_all(double *__restrict__ F, const int *__restrict__ dst, const int *__restrict__ start,
                                  const int *__restrict__ cnt, int n, const double *__restrict__ buf, int nz, int n2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (t >= n) return;
  const int d = dst[t], st = start[t], m = cnt[t];
  F[((long long)(d / n2) * nz + k) * n2 + d % n2] = buf[(long long)st * nz + (long long)k * m + (t - st)];
}

// Several fields in ONE halo update (the seven updates that end a step, step_mod.F90:467-560, travel as one message
// per neighbour).  blockIdx.y = level within the concatenation of the fields' levels; the message of a peer is
// [field][level][cell], i.e. level-major over the concatenation.
struct HaloFields { double *F[8]; int nz[8], lev0[8]; int nf, nztot; };
__device__ __forceinline__ int halo_field_of(const HaloFields &H, int kk) {
  int f = 0;
  while (f + 1 < H.nf && kk >= H.lev0[f + 1]) ++f;
  return f;
}
__global__ void k_halo_pack_many(HaloFields H, const int *__restrict__ src, const int *__restrict__ start, const int *__restrict__ cnt,
                                 int n, double *__restrict__ buf, int n2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, kk = blockIdx.y;
  if (t >= n) return;
  const int f = halo_field_of(H, kk), k = kk - H.lev0[f], nz = H.nz[f];
  const int s = src[t], st = start[t], m = cnt[t];
  buf[(long long)st * H.nztot + (long long)kk * m + (t - st)] = H.F[f][((long long)(s / n2) * nz + k) * n2 + s % n2];
}
__global__ void k_halo_unpack_many(HaloFields H, const int *__restrict__ dst, const int *__restrict__ start, const int *__restrict__ cnt,
                                   int n, const double *__restrict__ buf, int n2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, kk = blockIdx.y;
  if (t >= n) return;
  const int f = halo_field_of(H, kk), k = kk - H.lev0[f], nz = H.nz[f];
  const int d = dst[t], st = start[t], m = cnt[t];
  H.F[f][((long long)(d / n2) * nz + k) * n2 + d % n2] = buf[(long long)st * H.nztot + (long long)kk * m + (t - st)];
}
__global__ void k_halo_local_many(HaloFields H, const int *__restrict__ dst, const int *__restrict__ src, int ncopy,
                                  const int *__restrict__ filld, int nfill, double fill, int n2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, kk = blockIdx.y;
  const int f = halo_field_of(H, kk), k = kk - H.lev0[f], nz = H.nz[f];
  double *F = H.F[f];
  if (t < ncopy) {
    const int d = dst[t], s = src[t];
    F[((long long)(d / n2) * nz + k) * n2 + d % n2] = F[((long long)(s / n2) * nz + k) * n2 + s % n2];
  } else if (t < ncopy + nfill) {
    const int d = filld[t - ncopy];
    F[((long long)(d / n2) * nz + k) * n2 + d % n2] = fill;
  }
}

}  // namespace pop

// =============================================================================================
// Fused single-rank solver path.
//
// When every ghost cell of the rank has a local source (one rank owns all blocks) the halo
// update of the solver vectors is folded into the consuming kernel: a neighbour that is a ghost
// cell is read at its SOURCE cell (srcmap), which is bit-identical to reading the ghost after
// POP_HaloUpdate (closed-boundary ghosts read as the fill value 0).  The final stage of each dot
// product (ordered sum of the workgroup partials, block sums in block-id order -- the same rule as
// k_block_sums + k_finalize) is recomputed by every workgroup of the NEXT kernel, so an iteration
// is two launches for pcg and the scalars never leave the device.
// =============================================================================================
namespace pop {

struct FusedArgs {
  double *X, *R, *Z, *S0, *S1, *Q;
  double *AZ; const double *A0R;   // fused ChronGear: A z, 1/diag
  const double *Bv, *C;
  double *partA, *partB;      // workgroup partials of (r,z) and (s,q)
  SolverScalars *sc;
  const int *srcmap;          // per cell: own index (interior), source cell (ghost), -1 (fill)
  int nchunk, nblocks;
  // large grids: the per-block ordered sums of the partials are formed by k_block_sums between the
  // solver kernels (bsA/bsB) instead of being recomputed by every workgroup (O(#workgroups^2) reads)
  const double *bsA, *bsB;
  int presummed;
  // blocks spread over ranks: the one halo exchange per iteration (z) is packed by the kernel that produces z and
  // read in place from the receive buffer by the kernel that consumes it -- no pack / unpack launches.
  //   sendmap[q] >= 0: cell q is entry e of (send_off, send_slot): its value goes to sendbuf[send_slot[send_off[e] .. send_off[e+1])]
  //   rmap[q]    >= 0: ghost cell q is owned by another rank; its value is rbuf[rmap[q]] (message order of the peer lists)
  const int *sendmap, *send_off, *send_slot;
  double *sendbuf;
  const int *rmap;
  const double *rbuf;
};
// cells within NGHOST of the edge of the physical domain: the only ones a neighbour block's ghosts can copy
__device__ __forceinline__ bool send_band(const DevGrid &g, int b, int i, int j) {
  return i + 1 < g.ib + NGHOST || i + 1 > blk_ie(g, b) - NGHOST || j + 1 < g.jb + NGHOST || j + 1 > blk_je(g, b) - NGHOST;
}
__device__ __forceinline__ void pack_cell(const FusedArgs &a, long long q, double val) {
  const int e = a.sendmap[q];
  if (e < 0) return;
  for (int n = a.send_off[e]; n < a.send_off[e + 1]; ++n) a.sendbuf[a.send_slot[n]] = val;
}
// value of z at stencil source cell m (>= 0): from the receive buffer when m is a ghost owned by another rank
__device__ __forceinline__ double z_at(const FusedArgs &a, int m) {
  if (a.rmap) { const int rs = a.rmap[m]; if (rs >= 0) return a.rbuf[rs]; }
  return a.Z[m];
}

// A value another launch left in memory and nothing in THIS launch changes (the solver scalars eta0 / eta1 / alpha / stop, the block
// sums): read through the constant address space it becomes a scalar (SMEM) load -- no VGPR, no vmcnt, so it neither queues
// behind the operand loads in flight nor forces a full wait on them.  (r4, from the ISA of k_fpcg_b2: `alpha`, the block sum and
// `eta0` were three VECTOR loads with uniform addresses behind the 24 operand loads, each followed by s_waitcnt vmcnt(0): two more
// dependent round trips per wave.)  The scalar cache is invalidated at every kernel start, so the previous launch's stores are seen.
template <class T>
__device__ __forceinline__ T sld(const T *p) { return *(const __attribute__((address_space(4))) T *)p; }
// presummed form of the total: the block sums (already ordered sums, k_block_sums) added in block order; scalar loads
__device__ __forceinline__ double presummed_total(const double *__restrict__ bs, int nblocks) {
  double total = 0.0;
  for (int b = 0; b < nblocks; ++b) total = total + sld(bs + b);
  return total;
}
// ordered total of workgroup partials: per POP block a thread-strided sequential sum + fixed tree,
// block sums added in block order.  Every thread returns the same value.  (r4: the levels 32 .. 1 of the tree inside the first
// wavefront, as in wg_reduce_store -- the operands and the order of shf[t] + shf[t + s] -- and, later in r4, the two LDS levels formed by the
// lanes of that wavefront themselves: two barriers per block.)
__device__ __forceinline__ double fused_total(const double *__restrict__ partial, int nchunk, int nblocks,
                                              const double *__restrict__ bs, int presummed) {
  __shared__ double shf[POP_RED_THREADS];
  __shared__ double shf_tot;
  const int t = threadIdx.x;
  double total = 0.0;
  if (presummed) return presummed_total(bs, nblocks);   // block sums already formed (same ordered rule): add in block order
  for (int b = 0; b < nblocks; ++b) {
    double v = 0.0;
    for (int c = t; c < nchunk; c += POP_RED_THREADS) v = v + partial[(long long)b * nchunk + c];
    shf[t] = v;
    __syncthreads();
    if (t < 64) { const double x = tree_tail64((shf[t] + shf[t + 128]) + (shf[t + 64] + shf[t + 192])); if (t == 0) shf_tot = x; }
    __syncthreads();
    total = total + shf_tot;
  }
  return total;
}

// "not written yet" in a memory word that is its own flag (kernels_pcg_persist.hpp): a NaN bit pattern no arithmetic produces
constexpr unsigned long long POP_SPIN_EMPTY = 0x7FF8DEADBEEF0001ULL;

// step A: [r -= alpha q]; z = r/diag; partial (r,z).  The cell's operands are loaded before the ordered total of the
// previous partials is formed, so memory latency overlaps it.  The other half of the pending update, x += alpha s, is
// done by step B of the same iteration (round 3): B holds s of the previous iteration in registers anyway, so x costs it one
// load and one store, and A no longer reads x and s -- one word per point and iteration less, the same operations on the
// same operands (alpha is published by workgroup (0,0) of A and read by B after the launch boundary).
template <bool UPDATE>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_fpcg_a(DevGrid g, FusedArgs a) {
  const int stop = sld(&a.sc->stop);   // an earlier check has converged: this launch belongs to the look-ahead interval (branch below, after the loads are issued)
  if (red_land_out<1>(g, a.partA, a.nchunk, a.sendmap != nullptr)) return;
  const int p2 = red_cell(g), b = blockIdx.y;
  const bool live = p2 < g.n2;
  const long long q = (long long)b * g.n2 + (live ? p2 : 0);
  double r = a.R[q], qq = 0.0;
  if (UPDATE) qq = a.Q[q];
  const double cw = a.C[q], mk = (double)g.mMask8[q];
  if (stop) return;
  double alpha = 0.0;
  if (UPDATE) {
    const double sq = fused_total(a.partB, a.nchunk, a.nblocks, a.bsB, a.presummed);
    const double rz = sld(&a.sc->eta1);
    alpha = rz / sq;                                   // eta1 = eta0/(s,q), POP_SolversMod.F90:1419
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { a.sc->eta0 = rz; a.sc->alpha = alpha; }
  }
  double v[1] = {0.0};
  if (live) {
    const int i = p2 % g.nxb, j = p2 / g.nxb;
    if (UPDATE) {
      r = r - alpha * qq;
      a.R[q] = r;
    }
    const double z = (cw != 0.0) ? r / cw : 0.0;
    a.Z[q] = z;
    if (interior(g, b, i, j)) {
      v[0] = (r * z) * mk;
      if (a.sendmap && send_band(g, b, i, j)) pack_cell(a, q, z);
    }
  }
  wg_reduce_store<1>(v, a.partA, b * a.nchunk + red_chunk(g));
}

// step A with TWO chunks per workgroup (compacted launches on large grids, round 3): the kernel is bound by the length of a
// workgroup's dependency chain -- list entry -> operands -> total of the previous partials -> reduction tree -> partial -- times the
// number of workgroups a CU has to run one after the other (85 of them at tx0.1v3), not by bytes (3 words per point take 45 us, 6 take
// 68).  A thread takes the same cell of two chunks that are neighbours in the launch order of the same XCD band (list entries e
// and e + 8); both sets of operands are requested together and the two reduction trees share their barriers.  Each chunk's
// 256 products are reduced by the tree of wg_reduce_store into the chunk's own slot: bitwise the one-chunk kernel.
template <bool UPDATE>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_fpcg_a_pair(DevGrid g, FusedArgs a) {
  __shared__ double sh[2][POP_RED_THREADS];
  const int stop = sld(&a.sc->stop);
  const int b = blockIdx.y, t = threadIdx.x;
  const int e0 = ((int)blockIdx.x >> 3) * 16 + ((int)blockIdx.x & 7);
  int ch[2]; bool land[2]; long long q[2]; bool live[2];
  double r[2], qq[2] = {0.0, 0.0}, cw[2], mk[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = e0 + 8 * u;
    ch[u] = g.red_act[(long long)b * g.red_nact + e];
    // position of entry e in the sorted sequence (ocean chunks first); workgroup (0,0) publishes the scalars and never skips
    land[u] = g.skip && !(e == 0 && b == 0) && ((e & 7) * (g.red_nact >> 3) + (e >> 3)) >= g.red_cnt[b];
    const long long p2 = (long long)ch[u] * POP_RED_THREADS + t;
    live[u] = p2 < g.n2;
    q[u] = (long long)b * g.n2 + (live[u] ? p2 : 0);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    r[u] = a.R[q[u]];
    if (UPDATE) qq[u] = a.Q[q[u]];
    cw[u] = a.C[q[u]]; mk[u] = (double)g.mMask8[q[u]];
  }
  if (stop) return;
  double alpha = 0.0;
  if (UPDATE) {
    const double sq = fused_total(a.partB, a.nchunk, a.nblocks, a.bsB, 1);
    const double rz = sld(&a.sc->eta1);
    alpha = rz / sq;
    if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0) { a.sc->eta0 = rz; a.sc->alpha = alpha; }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    double v = 0.0;
    if (live[u] && !land[u]) {
      const int p2 = (int)(q[u] - (long long)b * g.n2);
      const int i = p2 % g.nxb, j = p2 / g.nxb;
      double rr = r[u];
      if (UPDATE) { rr = rr - alpha * qq[u]; a.R[q[u]] = rr; }
      const double z = (cw[u] != 0.0) ? rr / cw[u] : 0.0;
      a.Z[q[u]] = z;
      if (interior(g, b, i, j)) v = (rr * z) * mk[u];
    }
    sh[u][t] = v;
  }
  __syncthreads();
  if (t < 128) {   // the two chunks' trees in two waves, no barrier inside a tree (see wg_reduce_store)
    const int u = t >> 6, l = t & 63;
    const double x = tree_tail64((sh[u][l] + sh[u][l + 128]) + (sh[u][l + 64] + sh[u][l + 192]));
    if (l == 0) a.partA[(long long)b * a.nchunk + ch[u]] = x;
  }
}

// step B: s_new = z + s_old*(eta1/eta0) at the 9 stencil points; q = A s_new; partial (q,s).
// Only cells on the rim of the physical domain can have ghost neighbours: they read them through
// srcmap (bit-identical to reading the ghost after a halo update); all other cells index directly.
// XUPD: the pending x += alpha s of the previous iteration (see step A), on every cell step A updated it on
template <bool XUPD>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_fpcg_b(DevGrid g, FusedArgs a) {
  if (sld(&a.sc->stop)) return;   // an earlier check has converged: this launch belongs to the look-ahead interval
  if (red_land_out<1>(g, a.partB, a.nchunk, a.sendmap != nullptr)) return;
  const int p2 = red_cell(g), b = blockIdx.y;
  const bool live = p2 < g.n2;
  const int pp = live ? p2 : 0;
  const int i = pp % g.nxb, j = pp / g.nxb, nxb = g.nxb;
  const long long q = (long long)b * g.n2 + pp;
  const bool inner = live && interior(g, b, i, j);
  const bool rim = inner && (i + 1 == g.ib || i + 1 == blk_ie(g, b) || j + 1 == g.jb || j + 1 == blk_je(g, b));
  // gather operands first (independent loads in flight while the total is formed)
  const int off[9] = {0, nxb, -nxb, 1, -1, nxb + 1, -nxb + 1, nxb - 1, -nxb - 1};
  double zv[9], sv[9], wv[9];
  zv[0] = (inner || !live) ? a.Z[q] : z_at(a, (int)q); sv[0] = a.S0[q];
  const double xold = XUPD ? a.X[q] : 0.0;
#pragma unroll
  for (int t = 1; t < 9; ++t) { zv[t] = 0.0; sv[t] = 0.0; }
  if (inner) {
    if (!rim) {
#pragma unroll
      for (int t = 1; t < 9; ++t) { zv[t] = a.Z[q + off[t]]; sv[t] = a.S0[q + off[t]]; }
    } else {
#pragma unroll
      for (int t = 1; t < 9; ++t) {
        const int m = a.srcmap[q + off[t]];
        if (m >= 0) { zv[t] = z_at(a, m); sv[t] = a.S0[m]; }
      }
    }
    wv[0] = a.C[q]; wv[1] = g.WNo[q]; wv[2] = g.WNo[q - nxb]; wv[3] = g.WEa[q]; wv[4] = g.WEa[q - 1];
    wv[5] = g.WNE[q]; wv[6] = g.WNE[q - nxb]; wv[7] = g.WNE[q - 1]; wv[8] = g.WNE[q - 1 - nxb];
  }
  const double mk = (double)g.mMask8[q];
  const double rz = fused_total(a.partA, a.nchunk, a.nblocks, a.bsA, a.presummed);
  const double bt = rz / sld(&a.sc->eta0);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { a.sc->eta1 = rz; a.sc->beta_cg = bt; }
  double v[1] = {0.0};
  if (live) {
    if (XUPD) a.X[q] = xold + sld(&a.sc->alpha) * sv[0];
    const double s = zv[0] + sv[0] * bt;
    a.S1[q] = s;
    double aq = 0.0;
    if (inner) {
      aq = wv[0] * s;
#pragma unroll
      for (int t = 1; t < 9; ++t) aq = aq + wv[t] * (zv[t] + sv[t] * bt);
      v[0] = (aq * s) * mk;
    }
    a.Q[q] = aq;
  }
  wg_reduce_store<1>(v, a.partB, b * a.nchunk + red_chunk(g));
}

// ---- fused ChronGear (POP_SolversMod.F90:2100-2210): two launches per iteration ----------------------------
// ordered totals of two interleaved partial fields (same rule as fused_total, both fields in one pass)
__device__ __forceinline__ void fused_total2(const double *__restrict__ partial, int nchunk, int nblocks,
                                             const double *__restrict__ bs, int presummed, double &t0, double &t1) {
  __shared__ double shf[2][POP_RED_THREADS];
  const int t = threadIdx.x;
  t0 = 0.0; t1 = 0.0;
  if (presummed) {
    for (int b = 0; b < nblocks; ++b) { t0 = t0 + bs[2 * b]; t1 = t1 + bs[2 * b + 1]; }
    return;
  }
  for (int b = 0; b < nblocks; ++b) {
    double v0 = 0.0, v1 = 0.0;
    for (int c = t; c < nchunk; c += POP_RED_THREADS) {
      const double2 w = *reinterpret_cast<const double2 *>(partial + 2 * ((long long)b * nchunk + c));
      v0 = v0 + w.x; v1 = v1 + w.y;
    }
    shf[0][t] = v0; shf[1][t] = v1;
    __syncthreads();
    for (int s = POP_RED_THREADS / 2; s > 0; s >>= 1) {
      if (t < s) { shf[0][t] = shf[0][t] + shf[0][t + s]; shf[1][t] = shf[1][t] + shf[1][t + s]; }
      __syncthreads();
    }
    t0 = t0 + shf[0][0]; t1 = t1 + shf[1][0];
    __syncthreads();
  }
}
// z = r*A0R at stencil source cell m; a ghost owned by another rank has its owner's z in the receive buffer
__device__ __forceinline__ double cg_z_at(const FusedArgs &a, int m) {
  if (a.rmap) { const int rs = a.rmap[m]; if (rs >= 0) return a.rbuf[rs]; }
  return a.R[m] * a.A0R[m];
}
// step A (:2117-2157): z = r*A0R at the nine stencil points (ghost neighbours at their source cell, which is what
// the halo update of z delivers), az = A z, partial (r,z), (az,z)
__global__ void __launch_bounds__(POP_RED_THREADS)
k_fcg_a(DevGrid g, FusedArgs a) {
  if (a.sc->stop) return;   // an earlier check has converged: this launch belongs to the look-ahead interval
  if (red_land_out<2>(g, a.partA, a.nchunk, a.sendmap != nullptr)) return;
  const int p2 = red_cell(g), b = blockIdx.y;
  const bool live = p2 < g.n2;
  const int pp = live ? p2 : 0;
  const int i = pp % g.nxb, j = pp / g.nxb, nxb = g.nxb;
  const long long q = (long long)b * g.n2 + pp;
  const bool inner = live && interior(g, b, i, j);
  double v[2] = {0.0, 0.0};
  if (inner) {
    const bool rim = (i + 1 == g.ib || i + 1 == blk_ie(g, b) || j + 1 == g.jb || j + 1 == blk_je(g, b));
    const int off[9] = {0, nxb, -nxb, 1, -1, nxb + 1, -nxb + 1, nxb - 1, -nxb - 1};
    double zv[9], wv[9];
    const double r = a.R[q];
    zv[0] = r * a.A0R[q];
    if (!rim) {
#pragma unroll
      for (int t = 1; t < 9; ++t) zv[t] = a.R[q + off[t]] * a.A0R[q + off[t]];
    } else {
#pragma unroll
      for (int t = 1; t < 9; ++t) {
        const int m = a.srcmap[q + off[t]];
        zv[t] = (m >= 0) ? cg_z_at(a, m) : 0.0;
      }
    }
    wv[0] = a.C[q]; wv[1] = g.WNo[q]; wv[2] = g.WNo[q - nxb]; wv[3] = g.WEa[q]; wv[4] = g.WEa[q - 1];
    wv[5] = g.WNE[q]; wv[6] = g.WNE[q - nxb]; wv[7] = g.WNE[q - 1]; wv[8] = g.WNE[q - 1 - nxb];
    double az = wv[0] * zv[0];
#pragma unroll
    for (int t = 1; t < 9; ++t) az = az + wv[t] * zv[t];
    a.Z[q] = zv[0]; a.AZ[q] = az;
    const double mk = (double)g.mMask8[q];
    v[0] = (r * zv[0]) * mk; v[1] = (az * zv[0]) * mk;
  }
  wg_reduce_store<2>(v, a.partA, b * a.nchunk + red_chunk(g));
}
// step A with two horizontally adjacent cells per thread (large grids, even row pitch; see k_fpcg_b2): the pair shares
// its three stencil rows of r and A0R.  Same operations per cell, same reduction tree: bitwise equal to k_fcg_a.
__global__ void __launch_bounds__(POP_RED_THREADS / 2)
k_fcg_a2(DevGrid g, FusedArgs a) {
  if (a.sc->stop) return;   // an earlier check has converged: this launch belongs to the look-ahead interval
  if (red_land_out<2>(g, a.partA, a.nchunk, a.sendmap != nullptr)) return;
  __shared__ double sh[2][POP_RED_THREADS];
  const int b = blockIdx.y, t = threadIdx.x, nxb = g.nxb;
  const long long p0 = (long long)red_chunk(g) * POP_RED_THREADS + 2 * t;
  const bool live0 = p0 < g.n2, live1 = p0 + 1 < g.n2;
  const int pp = live0 ? (int)p0 : 0;
  const int i = pp % nxb, j = pp / nxb;
  const long long q = (long long)b * g.n2 + pp;
  const bool fast = live1 && i + 1 > g.ib && i + 2 < blk_ie(g, b) && j + 1 > g.jb && j + 1 < blk_je(g, b);
  double v[2][2] = {{0.0, 0.0}, {0.0, 0.0}};   // [cell][field]
  if (fast) {
    double zr[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const long long o = q + (long long)(r - 1) * nxb;
      const double2 rc = *reinterpret_cast<const double2 *>(a.R + o), ac = *reinterpret_cast<const double2 *>(a.A0R + o);
      zr[r][0] = a.R[o - 1] * a.A0R[o - 1]; zr[r][1] = rc.x * ac.x; zr[r][2] = rc.y * ac.y; zr[r][3] = a.R[o + 2] * a.A0R[o + 2];
    }
    const double2 rr = *reinterpret_cast<const double2 *>(a.R + q);
    const double2 cc = *reinterpret_cast<const double2 *>(a.C + q);
    const double2 no0 = *reinterpret_cast<const double2 *>(g.WNo + q), nom = *reinterpret_cast<const double2 *>(g.WNo + q - nxb);
    const double2 ea0 = *reinterpret_cast<const double2 *>(g.WEa + q);
    const double eaw = g.WEa[q - 1];
    const double2 ne0 = *reinterpret_cast<const double2 *>(g.WNE + q), nem = *reinterpret_cast<const double2 *>(g.WNE + q - nxb);
    const double ne0w = g.WNE[q - 1], nemw = g.WNE[q - 1 - nxb];
    double azA = cc.x * zr[1][1];
    azA = azA + no0.x * zr[2][1]; azA = azA + nom.x * zr[0][1]; azA = azA + ea0.x * zr[1][2]; azA = azA + eaw * zr[1][0];
    azA = azA + ne0.x * zr[2][2]; azA = azA + nem.x * zr[0][2]; azA = azA + ne0w * zr[2][0]; azA = azA + nemw * zr[0][0];
    double azB = cc.y * zr[1][2];
    azB = azB + no0.y * zr[2][2]; azB = azB + nom.y * zr[0][2]; azB = azB + ea0.y * zr[1][3]; azB = azB + ea0.x * zr[1][1];
    azB = azB + ne0.y * zr[2][3]; azB = azB + nem.y * zr[0][3]; azB = azB + ne0.x * zr[2][1]; azB = azB + nem.x * zr[0][1];
    *reinterpret_cast<double2 *>(a.Z + q) = make_double2(zr[1][1], zr[1][2]);
    *reinterpret_cast<double2 *>(a.AZ + q) = make_double2(azA, azB);
    const double mk0 = (double)g.mMask8[q], mk1 = (double)g.mMask8[q + 1];
    v[0][0] = (rr.x * zr[1][1]) * mk0; v[0][1] = (azA * zr[1][1]) * mk0;
    v[1][0] = (rr.y * zr[1][2]) * mk1; v[1][1] = (azB * zr[1][2]) * mk1;
  } else {
    // rim cells, straight-line (round 4; see k_fpcg_b2): source map, then the neighbours, every load unconditional at a clamped address
    const int off[9] = {0, nxb, -nxb, 1, -1, nxb + 1, -nxb + 1, nxb - 1, -nxb - 1};
    const long long qsafe = (long long)b * g.n2 + nxb + 1;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bool live = (e == 0 ? live0 : live1);
      const int p2 = live ? (int)(p0 + e) : 0, ii = p2 % nxb, jj = p2 / nxb;
      const long long qq = (long long)b * g.n2 + p2;
      const bool in_e = live && interior(g, b, ii, jj);
      const long long qn = in_e ? qq : qsafe;
      int m[9];
#pragma unroll
      for (int n = 1; n < 9; ++n) m[n] = a.srcmap[qn + off[n]];
      const double r = a.R[qn];
      const double z0 = r * a.A0R[qn];
      const double wv[9] = {a.C[qn], g.WNo[qn], g.WNo[qn - nxb], g.WEa[qn], g.WEa[qn - 1], g.WNE[qn], g.WNE[qn - nxb], g.WNE[qn - 1], g.WNE[qn - 1 - nxb]};
      const double mk = (double)g.mMask8[qn];
      double zn[9];
#pragma unroll
      for (int n = 1; n < 9; ++n) zn[n] = cg_z_at(a, (m[n] >= 0) ? m[n] : (int)qn);
      double az = wv[0] * z0;
#pragma unroll
      for (int n = 1; n < 9; ++n) az = az + wv[n] * ((m[n] >= 0) ? zn[n] : 0.0);
      if (in_e) {
        a.Z[qq] = z0; a.AZ[qq] = az;
        v[e][0] = (r * z0) * mk; v[e][1] = (az * z0) * mk;
      }
    }
  }
  // the tree of wg_reduce_store<2> over the 256 cells of the chunk
#pragma unroll
  for (int f = 0; f < 2; ++f) { sh[f][2 * t] = v[0][f]; sh[f][2 * t + 1] = v[1][f]; }
  __syncthreads();
  for (int s = POP_RED_THREADS / 2; s > 0; s >>= 1) {
    if (t < s) { sh[0][t] = sh[0][t] + sh[0][t + s]; sh[1][t] = sh[1][t] + sh[1][t + s]; }
    __syncthreads();
  }
  if (t == 0) {
    const long long slot = (long long)b * a.nchunk + red_chunk(g);
    a.partA[2 * slot] = sh[0][0]; a.partA[2 * slot + 1] = sh[1][0];
  }
}
// step B (:2159-2186): scalar recurrences from the two totals, then s = z + beta s; q = az + beta q; x += alpha s;
// r -= alpha q on the physical cells (ghost values of these vectors are never read in the fused form)
__global__ void __launch_bounds__(POP_RED_THREADS)
k_fcg_b(DevGrid g, FusedArgs a, int par) {
  if (a.sc->stop) return;   // an earlier check has converged: this launch belongs to the look-ahead interval
  if (red_land(g, a.sendmap != nullptr)) return;
  const int p2 = red_cell(g), b = blockIdx.y;
  const bool live = p2 < g.n2;
  const int pp = live ? p2 : 0;
  const int i = pp % g.nxb, j = pp / g.nxb;
  const long long q = (long long)b * g.n2 + pp;
  const bool inner = live && interior(g, b, i, j);
  double z = 0.0, az = 0.0, s = 0.0, qq = 0.0, x = 0.0, r = 0.0;
  if (inner) { z = a.Z[q]; az = a.AZ[q]; s = a.S0[q]; qq = a.Q[q]; x = a.X[q]; r = a.R[q]; }
  double rho, delta;
  fused_total2(a.partA, a.nchunk, a.nblocks, a.bsA, a.presummed, rho, delta);
  const double bt = rho / a.sc->rho2[par];
  const double sigma = delta - (bt * bt) * a.sc->sigma2[par];
  const double al = rho / sigma;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    a.sc->rho2[1 - par] = rho; a.sc->sigma2[1 - par] = sigma; a.sc->alpha = al; a.sc->beta_cg = bt; a.sc->rho_old = rho; a.sc->sigma = sigma;
  }
  if (inner) {
    s = z + bt * s;
    qq = az + bt * qq;
    a.S0[q] = s; a.Q[q] = qq;
    a.X[q] = x + al * s;
    const double rn = r - al * qq;
    a.R[q] = rn;
    if (a.sendmap && send_band(g, b, i, j)) pack_cell(a, q, rn * a.A0R[q]);   // next iteration's z for the neighbours' ghosts
  } else if (live && a.rmap) {
    // ghost owned by another rank: its z of this iteration is in the receive buffer; the search direction and the
    // solution there are advanced with the owner's arithmetic (needed by r = b - A x at the checks and as the ghost
    // values of the solution), so they never travel
    const int rs = a.rmap[q];
    if (rs >= 0) {
      const double sg = a.rbuf[rs] + bt * a.S0[q];
      a.S0[q] = sg;
      a.X[q] = a.X[q] + al * sg;
    }
  }
}

// step B with two horizontally adjacent cells per thread (large grids with an even row pitch): the chunk of 256
// consecutive cells is handled by 128 threads, the pair (q, q+1) shares its three stencil rows of z and s, which are
// read as (q-1), (q, q+1) as one 16-byte load, (q+2): 14 load instructions per cell instead of 28.  Same operations
// in the same order per cell, and the 256 products are reduced by the same tree as in k_fpcg_b: bitwise equal.
template <bool XUPD>
__global__ void __launch_bounds__(POP_RED_THREADS / 2)
k_fpcg_b2(DevGrid g, FusedArgs a) {
  const int stop = sld(&a.sc->stop);   // an earlier check has converged: this launch belongs to the look-ahead interval (branch below, after the loads are issued)
  // the scalars of the iteration first (scalar loads, see sld): presummed block sums only (host selects this kernel for large grids)
  const double alpha = XUPD ? sld(&a.sc->alpha) : 0.0;
  const double eta0 = sld(&a.sc->eta0);
  const double rz = presummed_total(a.bsA, a.nblocks);
  if (red_land_out<1>(g, a.partB, a.nchunk, a.sendmap != nullptr)) return;
  __shared__ double sh[POP_RED_THREADS];
  const int b = blockIdx.y, t = threadIdx.x, nxb = g.nxb;
  const long long p0 = (long long)red_chunk(g) * POP_RED_THREADS + 2 * t;     // first cell of the pair (even)
  const bool live0 = p0 < g.n2, live1 = p0 + 1 < g.n2;
  const int pp = live0 ? (int)p0 : 0;
  const int i = pp % nxb, j = pp / nxb;
  const long long q = (long long)b * g.n2 + pp;
  // fast path: both cells strictly inside the physical domain of the same row (no ghost neighbour, no srcmap)
  const bool fast = live1 && i + 1 > g.ib && i + 2 < blk_ie(g, b) && j + 1 > g.jb && j + 1 < blk_je(g, b);
  double v0 = 0.0, v1 = 0.0;
  double zr[3][4], sr[3][4];
  double2 cc, xx, x0, xm, y0, ym;
  double x0w = 0.0, xmw = 0.0, y0w = 0.0, ymw = 0.0, mk0 = 0.0, mk1 = 0.0;
  if (fast) {   // operands first: the loads are in flight while the total is formed
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const long long o = q + (long long)(r - 1) * nxb;
      const double2 zc = *reinterpret_cast<const double2 *>(a.Z + o), sc = *reinterpret_cast<const double2 *>(a.S0 + o);
      zr[r][0] = a.Z[o - 1]; zr[r][1] = zc.x; zr[r][2] = zc.y; zr[r][3] = a.Z[o + 2];
      sr[r][0] = a.S0[o - 1]; sr[r][1] = sc.x; sr[r][2] = sc.y; sr[r][3] = a.S0[o + 2];
    }
    cc = *reinterpret_cast<const double2 *>(a.C + q);
    // the off-centre weights from their two U-point terms (rows j and j-1, columns q-1 .. q+1): two fields instead of three
    x0 = *reinterpret_cast<const double2 *>(g.XW + q); xm = *reinterpret_cast<const double2 *>(g.XW + q - nxb);
    y0 = *reinterpret_cast<const double2 *>(g.YW + q); ym = *reinterpret_cast<const double2 *>(g.YW + q - nxb);
    x0w = g.XW[q - 1]; xmw = g.XW[q - 1 - nxb]; y0w = g.YW[q - 1]; ymw = g.YW[q - 1 - nxb];
    mk0 = (double)g.mMask8[q]; mk1 = (double)g.mMask8[q + 1];
    if (XUPD) xx = *reinterpret_cast<const double2 *>(a.X + q);
  }
  if (stop) return;
  const double bt = rz / eta0;
  if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0) { a.sc->eta1 = rz; a.sc->beta_cg = bt; }
  if (fast) {
    // rows: 0 = j-1, 1 = j, 2 = j+1; columns: 0 = q-1, 1 = q, 2 = q+1, 3 = q+2
    if (XUPD) *reinterpret_cast<double2 *>(a.X + q) = make_double2(xx.x + alpha * sr[1][1], xx.y + alpha * sr[1][2]);
    // WNE = xne + yne, WEa = xne + xse - yne - yse, WNo = yne + ynw - xne - xnw (host_setup.cpp, the same additions in the same order)
    double2 no0, nom, ea0, ne0, nem;
    const double ne0w = x0w + y0w, nemw = xmw + ymw;
    ne0.x = x0.x + y0.x; ne0.y = x0.y + y0.y; nem.x = xm.x + ym.x; nem.y = xm.y + ym.y;
    const double eaw = x0w + xmw - y0w - ymw;
    ea0.x = x0.x + xm.x - y0.x - ym.x; ea0.y = x0.y + xm.y - y0.y - ym.y;
    no0.x = y0.x + y0w - x0.x - x0w; no0.y = y0.y + y0.x - x0.y - x0.x;
    nom.x = ym.x + ymw - xm.x - xmw; nom.y = ym.y + ym.x - xm.y - xm.x;
#define SN(r, c) (zr[r][c] + sr[r][c] * bt)
    const double sA = SN(1, 1), sB = SN(1, 2);
    double aqA = cc.x * sA;
    aqA = aqA + no0.x * SN(2, 1); aqA = aqA + nom.x * SN(0, 1); aqA = aqA + ea0.x * SN(1, 2); aqA = aqA + eaw * SN(1, 0);
    aqA = aqA + ne0.x * SN(2, 2); aqA = aqA + nem.x * SN(0, 2); aqA = aqA + ne0w * SN(2, 0); aqA = aqA + nemw * SN(0, 0);
    double aqB = cc.y * sB;
    aqB = aqB + no0.y * SN(2, 2); aqB = aqB + nom.y * SN(0, 2); aqB = aqB + ea0.y * SN(1, 3); aqB = aqB + ea0.x * SN(1, 1);
    aqB = aqB + ne0.y * SN(2, 3); aqB = aqB + nem.y * SN(0, 3); aqB = aqB + ne0.x * SN(2, 1); aqB = aqB + nem.x * SN(0, 1);
#undef SN
    *reinterpret_cast<double2 *>(a.S1 + q) = make_double2(sA, sB);
    *reinterpret_cast<double2 *>(a.Q + q) = make_double2(aqA, aqB);
    v0 = (aqA * sA) * mk0; v1 = (aqB * sB) * mk1;
  } else {
    // generic path, cell by cell (rim of the physical domain, ghost cells, ragged end of the block).  Straight-line since round 4: a wave
    // that holds ONE such lane runs this path for all of them, and with the loads inside `if (in_e)` / `if (m >= 0)` it was ~17 dependent
    // round trips (srcmap -> z, s per neighbour); now every load is unconditional at a clamped address -- three rounds (cell, source map,
    // neighbours) -- and the conditions select values.  Same operations on the same operands.
    const int off[9] = {0, nxb, -nxb, 1, -1, nxb + 1, -nxb + 1, nxb - 1, -nxb - 1};
    const long long qsafe = (long long)b * g.n2 + nxb + 1;            // a cell whose eight neighbours exist in the array
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bool live = (e == 0 ? live0 : live1);
      const int p2 = live ? (int)(p0 + e) : 0, ii = p2 % nxb, jj = p2 / nxb;
      const long long qq = (long long)b * g.n2 + p2;
      const bool in_e = live && interior(g, b, ii, jj);
      const long long qn = in_e ? qq : qsafe;                         // where the stencil operands are read
      const double sold = a.S0[qq], xold = XUPD ? a.X[qq] : 0.0;
      const double zown = in_e ? a.Z[qq] : z_at(a, (int)qq);
      int m[9];
#pragma unroll
      for (int n = 1; n < 9; ++n) m[n] = a.srcmap[qn + off[n]];
      const double wv[9] = {a.C[qn], g.WNo[qn], g.WNo[qn - nxb], g.WEa[qn], g.WEa[qn - 1], g.WNE[qn], g.WNE[qn - nxb], g.WNE[qn - 1], g.WNE[qn - 1 - nxb]};
      const double mk = (double)g.mMask8[qn];
      double zn[9], sn0[9];
#pragma unroll
      for (int n = 1; n < 9; ++n) { const int mm = (m[n] >= 0) ? m[n] : (int)qn; zn[n] = z_at(a, mm); sn0[n] = a.S0[mm]; }
      const double s = zown + sold * bt;
      double aq = 0.0, vv = 0.0;
      if (in_e) {
        aq = wv[0] * s;
#pragma unroll
        for (int n = 1; n < 9; ++n) {
          const double sn = (m[n] >= 0) ? zn[n] + sn0[n] * bt : 0.0 + 0.0 * bt;
          aq = aq + wv[n] * sn;
        }
        vv = (aq * s) * mk;
      }
      if (live) {
        if (XUPD) a.X[qq] = xold + alpha * sold;
        a.S1[qq] = s; a.Q[qq] = aq;
      }
      if (e == 0) v0 = vv; else v1 = vv;
    }
  }
  // the tree of wg_reduce_store<1> over the 256 cells of the chunk
  sh[2 * t] = v0; sh[2 * t + 1] = v1;
  __syncthreads();
  if (t < 64) {   // (one wave, no barrier inside the tree: see wg_reduce_store)
    const double x = tree_tail64((sh[t] + sh[t + 128]) + (sh[t + 64] + sh[t + 192]));
    if (t == 0) a.partB[(long long)b * a.nchunk + red_chunk(g)] = x;
  }
}

// pending x,r update before a convergence check
__global__ void __launch_bounds__(POP_RED_THREADS)
k_fpcg_xr(DevGrid g, FusedArgs a) {
  if (a.sc->stop) return;   // an earlier check has converged: this launch belongs to the look-ahead interval
  if (red_land(g, a.sendmap != nullptr)) return;
  const int p2 = red_cell(g), b = blockIdx.y;
  const double sq = fused_total(a.partB, a.nchunk, a.nblocks, a.bsB, a.presummed);
  const double rz = a.sc->eta1;
  const double alpha = rz / sq;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { a.sc->eta0 = rz; a.sc->alpha = alpha; }
  if (p2 >= g.n2) return;
  const long long q = (long long)b * g.n2 + p2;
  a.X[q] = a.X[q] + alpha * a.S0[q];
  // r -= alpha q is not formed: every use of this kernel is followed by r = b - A x (convergence check) or by the
  // end of the solve, so the reference's update (:1433-1438) would be overwritten or never read
}

// r = b - A x on the physical domain with ghost neighbours of x read at their source; partial (r,r)
template <bool WITH_RR>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_fresidual(DevGrid g, FusedArgs a) {
  if (a.sc->stop) return;   // an earlier check has converged: this launch belongs to the look-ahead interval
  if (WITH_RR ? red_land_out<1>(g, a.partA, a.nchunk, a.sendmap != nullptr) : red_land(g, a.sendmap != nullptr)) return;
  const int p2 = red_cell(g), b = blockIdx.y;
  double v[1] = {0.0};
  if (p2 < g.n2) {
    const int i = p2 % g.nxb, j = p2 / g.nxb, nxb = g.nxb;
    const long long q = (long long)b * g.n2 + p2;
    if (interior(g, b, i, j)) {
      auto xs = [&](long long qq) { const int m = a.srcmap[qq]; return (m < 0) ? 0.0 : a.X[m]; };
      const double ax = a.C[q] * a.X[q] + g.WNo[q] * xs(q + nxb) + g.WNo[q - nxb] * xs(q - nxb) + g.WEa[q] * xs(q + 1) + g.WEa[q - 1] * xs(q - 1) +
                        g.WNE[q] * xs(q + nxb + 1) + g.WNE[q - nxb] * xs(q - nxb + 1) + g.WNE[q - 1] * xs(q + nxb - 1) + g.WNE[q - 1 - nxb] * xs(q - nxb - 1);
      const double r = a.Bv[q] - ax;
      a.R[q] = r;
      if (WITH_RR) v[0] = (r * r) * (double)g.mMask8[q];
      // distributed ChronGear: the z = r*A0R the next iteration's neighbours need travels from here
      if (a.sendmap && a.A0R && send_band(g, b, i, j)) pack_cell(a, q, r * a.A0R[q]);
    }
  }
  if (WITH_RR) wg_reduce_store<1>(v, a.partA, b * a.nchunk + red_chunk(g));
}

// r = b - A x with two horizontally adjacent cells per thread (large grids, even row pitch; see k_fpcg_b2): bitwise
// equal to k_fresidual
template <bool WITH_RR>
__global__ void __launch_bounds__(POP_RED_THREADS / 2)
k_fresidual2(DevGrid g, FusedArgs a) {
  if (a.sc->stop) return;   // an earlier check has converged: this launch belongs to the look-ahead interval
  if (WITH_RR ? red_land_out<1>(g, a.partA, a.nchunk, a.sendmap != nullptr) : red_land(g, a.sendmap != nullptr)) return;
  __shared__ double sh[POP_RED_THREADS];
  const int b = blockIdx.y, t = threadIdx.x, nxb = g.nxb;
  const long long p0 = (long long)red_chunk(g) * POP_RED_THREADS + 2 * t;
  const bool live0 = p0 < g.n2, live1 = p0 + 1 < g.n2;
  const int pp = live0 ? (int)p0 : 0;
  const int i = pp % nxb, j = pp / nxb;
  const long long q = (long long)b * g.n2 + pp;
  const bool fast = live1 && i + 1 > g.ib && i + 2 < blk_ie(g, b) && j + 1 > g.jb && j + 1 < blk_je(g, b);
  double v0 = 0.0, v1 = 0.0;
  if (fast) {
    double xr[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const long long o = q + (long long)(r - 1) * nxb;
      const double2 xc = *reinterpret_cast<const double2 *>(a.X + o);
      xr[r][0] = a.X[o - 1]; xr[r][1] = xc.x; xr[r][2] = xc.y; xr[r][3] = a.X[o + 2];
    }
    const double2 cc = *reinterpret_cast<const double2 *>(a.C + q);
    const double2 no0 = *reinterpret_cast<const double2 *>(g.WNo + q), nom = *reinterpret_cast<const double2 *>(g.WNo + q - nxb);
    const double2 ea0 = *reinterpret_cast<const double2 *>(g.WEa + q);
    const double eaw = g.WEa[q - 1];
    const double2 ne0 = *reinterpret_cast<const double2 *>(g.WNE + q), nem = *reinterpret_cast<const double2 *>(g.WNE + q - nxb);
    const double ne0w = g.WNE[q - 1], nemw = g.WNE[q - 1 - nxb];
    const double2 bv = *reinterpret_cast<const double2 *>(a.Bv + q);
    const double axA = cc.x * xr[1][1] + no0.x * xr[2][1] + nom.x * xr[0][1] + ea0.x * xr[1][2] + eaw * xr[1][0] +
                       ne0.x * xr[2][2] + nem.x * xr[0][2] + ne0w * xr[2][0] + nemw * xr[0][0];
    const double axB = cc.y * xr[1][2] + no0.y * xr[2][2] + nom.y * xr[0][2] + ea0.y * xr[1][3] + ea0.x * xr[1][1] +
                       ne0.y * xr[2][3] + nem.y * xr[0][3] + ne0.x * xr[2][1] + nem.x * xr[0][1];
    const double rA = bv.x - axA, rB = bv.y - axB;
    *reinterpret_cast<double2 *>(a.R + q) = make_double2(rA, rB);
    if (WITH_RR) { v0 = (rA * rA) * (double)g.mMask8[q]; v1 = (rB * rB) * (double)g.mMask8[q + 1]; }
    if (a.sendmap && a.A0R) {   // the second ring of the send band lies inside the fast region
      if (send_band(g, b, i, j)) pack_cell(a, q, rA * a.A0R[q]);
      if (send_band(g, b, i + 1, j)) pack_cell(a, q + 1, rB * a.A0R[q + 1]);
    }
  } else {
    // rim cells, straight-line (round 4; see k_fpcg_b2): source map, then the neighbours, every load unconditional at a clamped address
    const int off[8] = {nxb, -nxb, 1, -1, nxb + 1, -nxb + 1, nxb - 1, -nxb - 1};
    const long long qsafe = (long long)b * g.n2 + nxb + 1;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bool live = (e == 0 ? live0 : live1);
      const int p2 = live ? (int)(p0 + e) : 0, ii = p2 % nxb, jj = p2 / nxb;
      const long long qq = (long long)b * g.n2 + p2;
      const bool in_e = live && interior(g, b, ii, jj);
      const long long qn = in_e ? qq : qsafe;
      int m[8];
#pragma unroll
      for (int n = 0; n < 8; ++n) m[n] = a.srcmap[qn + off[n]];
      const double wc = a.C[qn], xc = a.X[qn], bvv = a.Bv[qn], mk = (double)g.mMask8[qn];
      const double wv[8] = {g.WNo[qn], g.WNo[qn - nxb], g.WEa[qn], g.WEa[qn - 1], g.WNE[qn], g.WNE[qn - nxb], g.WNE[qn - 1], g.WNE[qn - 1 - nxb]};
      double xn[8];
#pragma unroll
      for (int n = 0; n < 8; ++n) { const double x = a.X[(m[n] >= 0) ? m[n] : (int)qn]; xn[n] = (m[n] < 0) ? 0.0 : x; }
      const double ax = wc * xc + wv[0] * xn[0] + wv[1] * xn[1] + wv[2] * xn[2] + wv[3] * xn[3] + wv[4] * xn[4] + wv[5] * xn[5] + wv[6] * xn[6] + wv[7] * xn[7];
      const double r = bvv - ax;
      if (in_e) {
        a.R[qq] = r;
        if (WITH_RR) { const double vv = (r * r) * mk; if (e == 0) v0 = vv; else v1 = vv; }
        if (a.sendmap && a.A0R && send_band(g, b, ii, jj)) pack_cell(a, qq, r * a.A0R[qq]);
      }
    }
  }
  if (WITH_RR) {
    sh[2 * t] = v0; sh[2 * t + 1] = v1;
    __syncthreads();
    for (int s = POP_RED_THREADS / 2; s > 0; s >>= 1) {
      if (t < s) sh[t] = sh[t] + sh[t + s];
      __syncthreads();
    }
    if (t == 0) a.partA[(long long)b * a.nchunk + red_chunk(g)] = sh[0];
  }
}

// view of the 2-D system the fused solver works on
struct SolveView {
  DevGrid g;
  double *X, *R, *Z, *S0, *S1, *Q, *RHS, *C, *partial, *blocksum;
  int *srcmap, *gid;
  int nchunk, nblocks_tot;
};
// ghost cells <- source cells / fill value (every source is an interior cell of the same array)
__global__ void k_halo_srcmap(double *__restrict__ X, const int *__restrict__ srcmap, long long n) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int m = srcmap[q];
  if (m != (int)q) X[q] = (m < 0) ? 0.0 : X[m];
}
// operator centre weight on all blocks (barotropic.F90:535-554, POP_SolversMod.F90:1144)
__global__ void k_center_all(DevGrid g, StepParams sp, const double *__restrict__ TAREA, const int *__restrict__ KMT,
                             double *__restrict__ C, long long n) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const double dc = (KMT[q] >= 1) ? TAREA[q] / (sp.beta * sp.c2dtp * sp.dtp * sp.grav) : 0.0;
  C[q] = g.WC0[q] - dc;
}

}  // namespace pop
