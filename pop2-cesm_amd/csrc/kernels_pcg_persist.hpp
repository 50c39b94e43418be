// kernels_pcg_persist.hpp -- the whole preconditioned conjugate-gradient solve (POP_SolversMod.F90:1200-1503, diagonal
// preconditioner) of a SMALL 2-D system as ONE resident launch (round 4; pop_tuning.pcg_persist).
//
// On grids whose ten solver vectors are a few megabytes (gx1v7: 122 880 points) the fused two-launch iteration of
// kernels_barotropic.hpp sits at the floor of a dependent launch: 2 x ~5 us of launch boundary + the in-kernel total of the
// previous launch's partials = 14 us per iteration for ~1 us of arithmetic, and at N > 1 that solve is replicated on every rank, so
// it is the part of the step that does not shard.  Here at most 128 workgroups stay resident for the whole solve.  A workgroup owns
// CP consecutive 256-cell chunks (the chunks and partial slots of the fused kernels); x, s and z of its cells AND of every cell its
// stencils read (its "window": own cells, then the halo cells, then one cell that holds zeros for the fill value of closed
// boundaries) live in LDS, r, q, the nine weights and the window indices of the eight neighbours in registers, for all iterations.
// What crosses workgroups per iteration: the chunk partials of the two inner products and z of the cells in somebody's halo.
// The search direction and the solution at halo cells are advanced by the workgroup itself with the owner's arithmetic (the rule of
// the distributed solvers), so they never travel.
//
// Every exchanged word is its own flag: a 64-bit relaxed agent-scope atomic store of the VALUE into a slot that holds
// POP_SPIN_EMPTY (a NaN bit pattern no arithmetic produces) until then; readers re-load until they see something else.  No fence,
// no ticket, no read-modify-write, no ordering assumption between two words.  The slots of phase n live in buffer n mod 3; a
// workgroup resets its own slots of buffer (n+1) mod 3 before it writes phase n -- that buffer was last read in phase n-2, and
// nobody can be in phase n before everybody has written phase n-1, i.e. finished reading phase n-2.  Collecting all partials of a
// phase is therefore also the grid-wide barrier of the phase.
//
// Same numbers as the fused launches, bit for bit (tests/test_gpu_parity.py::test_persistent_pcg_is_bitwise_the_fused_pcg): the
// chunk partials are formed by the tree of wg_reduce_store, their total by the rule of fused_total (thread-strided left-to-right
// sums per block, fixed tree, blocks in order), the cell arithmetic in the order of k_fpcg_a / k_fpcg_b / k_fpcg_xr / k_fresidual.
// A wait that does not end (a workgroup that died) gives up after ~0.5 s, marks the workgroup dead -- no later wait spins -- and
// the solve reports it (status word), so a bug here is a loud failure and never a hung GPU.
#pragma once
#include "kernels_barotropic.hpp"

namespace pop {

constexpr int POP_PERSIST_MAXP = 8;      // partial slots one thread collects per phase: nblocks * ceil(nchunk / 256) must not exceed it
constexpr int POP_PERSIST_MAXH = 8;      // halo cells one thread fetches per iteration: ceil(nhalo / 256)

struct PersistArgs {
  double *X; const double *Bv, *C, *WNo, *WEa, *WNE; const unsigned char *mMask8;
  int nxb, nchunk, nblocks, nslots;      // partial slot = block * nchunk + chunk
  long long ncell;                       // n2 * nblocks
  const int *own_q;                      // [nwg * CP * 256] cell of own position L = u * 256 + t; -1: not a physical (interior) cell
  const unsigned short *nbr;             // [nwg * CP * 256 * 8] window index of the eight stencil neighbours (order of k_fpcg_b)
  const int *halo_off, *halo_q;          // halo cells of workgroup w: halo_q[halo_off[w] .. halo_off[w+1]), window index CP * 256 + h
  unsigned long long *P, *Zb;            // [3][nslots] partials, [3][ncell] z
  int max_iter, freq;
  double criterion;
  double *out;                           // pinned: [0] iterations, [1] last (r,r), [2] 0 ok / 1 a wait gave up, [3] checks done
};

__device__ __forceinline__ unsigned long long ld_word(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_word(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int CP>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_pcg_persist(PersistArgs a) {
  extern __shared__ double lds[];                          // Xw | Sw | Zw, nwin doubles each
  __shared__ double sh[CP][POP_RED_THREADS];               // the chunk trees; row 0 also serves the total's tree
  __shared__ double sh_x;
  __shared__ int dead;
  constexpr int NT = POP_RED_THREADS, NOWN = CP * NT;
  const int t = threadIdx.x, w = blockIdx.x;
  const int h0 = a.halo_off[w], nhalo = a.halo_off[w + 1] - h0, nwin = NOWN + nhalo + 1;
  double *Xw = lds, *Sw = lds + nwin, *Zw = lds + 2 * nwin;
  if (t == 0) dead = 0;

  // ---- own cells: operands that never change, in registers
  int q[CP]; bool inner[CP];
  double cw[CP], mk[CP], wv[CP][8], r[CP], qq[CP];
  unsigned short nb[CP][8];
#pragma unroll
  for (int u = 0; u < CP; ++u) {
    const long long L = (long long)w * NOWN + u * NT + t;
    q[u] = a.own_q[L];
    inner[u] = q[u] >= 0;
    const long long qc = inner[u] ? q[u] : 0;
    const int nxb = a.nxb;
    cw[u] = a.C[qc]; mk[u] = (double)a.mMask8[qc];
    // the neighbours of a cell at the edge of the array are never read for a cell that is not physical: clamp the addresses
    const long long qs = (qc - nxb - 1 >= 0) ? qc : (long long)nxb + 1;
    wv[u][0] = a.WNo[qs]; wv[u][1] = a.WNo[qs - nxb]; wv[u][2] = a.WEa[qs]; wv[u][3] = a.WEa[qs - 1];
    wv[u][4] = a.WNE[qs]; wv[u][5] = a.WNE[qs - nxb]; wv[u][6] = a.WNE[qs - 1]; wv[u][7] = a.WNE[qs - 1 - nxb];
#pragma unroll
    for (int n = 0; n < 8; ++n) nb[u][n] = a.nbr[L * 8 + n];
    r[u] = 0.0; qq[u] = 0.0;
  }
  // ---- window: solution from the first guess, search direction 0 (POP_SolversMod.F90:1290-1300)
  for (int L = t; L < nwin; L += NT) {
    int qL = -1;
    if (L < NOWN) qL = a.own_q[(long long)w * NOWN + L];
    else if (L < NOWN + nhalo) qL = a.halo_q[h0 + L - NOWN];
    Xw[L] = qL >= 0 ? a.X[qL] : 0.0; Sw[L] = 0.0; Zw[L] = 0.0;
  }
  __syncthreads();

  int phase = 0;                                           // number of the next exchange: its slots are in buffer phase % 3
  // ---- one exchange: the chunk partials of v[] out, everybody's partials in, their ordered total back (every thread the same
  //      value); with_z: also z of the halo cells of iteration m
  auto exchange = [&](double (&v)[CP], bool with_z, int m) -> double {
#pragma unroll
    for (int u = 0; u < CP; ++u) sh[u][t] = v[u];
    __syncthreads();
    for (int s = NT / 2; s >= 64; s >>= 1) {
      if (t < s) {
#pragma unroll
        for (int u = 0; u < CP; ++u) sh[u][t] = sh[u][t] + sh[u][t + s];
      }
      __syncthreads();
    }
    unsigned long long *Pn = a.P + (long long)(phase % 3) * a.nslots, *Pr = a.P + (long long)((phase + 1) % 3) * a.nslots;
    if (t < 64) {
#pragma unroll
      for (int u = 0; u < CP; ++u) {
        const double x = tree_tail64(sh[u][t]);
        const int slot = w * CP + u;
        if (t == 0 && slot < a.nslots) { st_word(Pr + slot, POP_SPIN_EMPTY); st_word(Pn + slot, (unsigned long long)__double_as_longlong(x)); }
      }
    }
    // what this thread collects: slots b * nchunk + c, c = t, t + 256, ... of every block, and (with_z) its halo cells
    unsigned long long pv[POP_PERSIST_MAXP], zv[POP_PERSIST_MAXH];
    const unsigned long long *Zn = a.Zb + (long long)(m % 3) * a.ncell;
    const int per_b = (a.nchunk - t + NT - 1) / NT;        // chunks of one block this thread adds (<= 0: none)
    for (int tries = 0;; ++tries) {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < POP_PERSIST_MAXP; ++k) {
        const int b = per_b > 0 ? k / per_b : a.nblocks, c = per_b > 0 ? t + (k % per_b) * NT : 0;
        pv[k] = (b < a.nblocks) ? ld_word(Pn + (long long)b * a.nchunk + c) : 0ULL;
      }
      if (with_z) {
#pragma unroll
        for (int k = 0; k < POP_PERSIST_MAXH; ++k) {
          const int hh = t + k * NT;
          zv[k] = (hh < nhalo) ? ld_word(Zn + a.halo_q[h0 + hh]) : 0ULL;
        }
      }
#pragma unroll
      for (int k = 0; k < POP_PERSIST_MAXP; ++k) ok = ok && pv[k] != POP_SPIN_EMPTY;
      if (with_z) {
#pragma unroll
        for (int k = 0; k < POP_PERSIST_MAXH; ++k) ok = ok && zv[k] != POP_SPIN_EMPTY;
      }
      if (ok) break;
      if (tries > (1 << 20) || *(volatile int *)&dead) { dead = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (with_z) {
#pragma unroll
      for (int k = 0; k < POP_PERSIST_MAXH; ++k) { const int hh = t + k * NT; if (hh < nhalo) Zw[NOWN + hh] = __longlong_as_double((long long)zv[k]); }
    }
    // the rule of fused_total: per block the thread-strided left-to-right sum, the fixed tree, blocks in order
    double total = 0.0;
    for (int b = 0; b < a.nblocks; ++b) {
      double x = 0.0;
#pragma unroll
      for (int k = 0; k < POP_PERSIST_MAXP; ++k)
        if (per_b > 0 && k / per_b == b) x = x + __longlong_as_double((long long)pv[k]);
      __syncthreads();
      sh[0][t] = x;
      __syncthreads();
      for (int s = NT / 2; s >= 64; s >>= 1) {
        if (t < s) sh[0][t] = sh[0][t] + sh[0][t + s];
        __syncthreads();
      }
      if (t < 64) { const double y = tree_tail64(sh[0][t]); if (t == 0) sh_x = y; }
      __syncthreads();
      total = total + sh_x;
    }
    __syncthreads();
    ++phase;
    return total;
  };
  // A x at own cell u from the window array W (x or the new search direction): the order of btropOperator as the fused kernels
  // evaluate it -- centre, N, S, E, W, NE, SE, NW, SW
  auto apply = [&](const double *W, int u) -> double {
    double ax = cw[u] * W[u * NT + t];
#pragma unroll
    for (int n = 0; n < 8; ++n) ax = ax + wv[u][n] * W[nb[u][n]];
    return ax;
  };
  auto residual = [&]() {                                  // r = b - A x on the physical cells (k_fresidual)
#pragma unroll
    for (int u = 0; u < CP; ++u) r[u] = inner[u] ? a.Bv[q[u]] - apply(Xw, u) : 0.0;
  };
  auto advance_x = [&](double alpha) {                     // x += alpha s on every window cell (k_fpcg_b XUPD / k_fpcg_xr)
    for (int L = t; L < nwin - 1; L += NT) Xw[L] = Xw[L] + alpha * Sw[L];
  };

  residual();
  double eta0 = 1.0, eta1 = 0.0, sq = 0.0, rr = 0.0;
  bool pending = false;
  int m = 0, nchecks = 0, converged = 0;
  double v[CP];
  while (m < a.max_iter) {
    ++m;
    // ---- step A (k_fpcg_a): [r -= alpha q]; z = r / diag; partial (r, z)
    double alpha = 0.0;
    if (pending) { alpha = eta1 / sq; eta0 = eta1; }
    unsigned long long *Zn = a.Zb + (long long)(m % 3) * a.ncell, *Zr = a.Zb + (long long)((m + 1) % 3) * a.ncell;
#pragma unroll
    for (int u = 0; u < CP; ++u) {
      v[u] = 0.0;
      if (inner[u]) {
        if (pending) r[u] = r[u] - alpha * qq[u];
        const double z = (cw[u] != 0.0) ? r[u] / cw[u] : 0.0;
        Zw[u * NT + t] = z;
        st_word(Zr + q[u], POP_SPIN_EMPTY);
        st_word(Zn + q[u], (unsigned long long)__double_as_longlong(z));
        v[u] = (r[u] * z) * mk[u];
      }
    }
    const double rz = exchange(v, true, m);
    // ---- step B (k_fpcg_b): [x += alpha s]; s = z + s beta at every window cell; q = A s; partial (q, s)
    const double bt = rz / eta0;
    eta1 = rz;
    if (pending) advance_x(alpha);
    for (int L = t; L < nwin - 1; L += NT) Sw[L] = Zw[L] + Sw[L] * bt;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CP; ++u) {
      v[u] = 0.0; qq[u] = 0.0;
      if (inner[u]) {
        const double s = Sw[u * NT + t];
        const double aq = apply(Sw, u);
        qq[u] = aq;
        v[u] = (aq * s) * mk[u];
      }
    }
    sq = exchange(v, false, m);
    pending = true;
    if (m % a.freq == 0) {
      // ---- convergence check (k_fpcg_xr, k_fresidual<true>, k_rr_total): the pending x update, r = b - A x, (r, r)
      alpha = eta1 / sq; eta0 = eta1;
      advance_x(alpha);
      pending = false;
      __syncthreads();
      residual();
#pragma unroll
      for (int u = 0; u < CP; ++u) v[u] = inner[u] ? (r[u] * r[u]) * mk[u] : 0.0;
      rr = exchange(v, false, m);
      ++nchecks;
      if (rr < a.criterion) { converged = 1; break; }
    }
  }
  if (pending) { const double alpha = eta1 / sq; advance_x(alpha); }   // iterations past the last check (max_iter not a multiple of freq)
  __syncthreads();
#pragma unroll
  for (int u = 0; u < CP; ++u) if (inner[u]) a.X[q[u]] = Xw[u * NT + t];
  if (w == 0 && t == 0) { a.out[0] = converged ? (double)m : (double)a.max_iter; a.out[1] = rr; a.out[3] = (double)nchecks; }
  if (t == 0 && dead) a.out[2] = 1.0;
}

// every exchange word empty before a solve
__global__ void k_fill_words(unsigned long long *p, long long n, unsigned long long v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace pop
