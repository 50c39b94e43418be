// kernels_pcg_persist.hpp -- the whole preconditioned conjugate-gradient solve (POP_SolversMod.F90:1200-1503, diagonal
// preconditioner) of a SMALL 2-D system as ONE resident launch (round 4; pop_tuning.pcg_persist); the ChronGear and the P-CSI iterations likewise (k_cg_persist, k_pcsi_persist, below).
//
// On grids whose ten solver vectors are a few megabytes (gx1v7: 122 880 points) the fused two-launch iteration of
// kernels_barotropic.hpp sits at the floor of a dependent launch: 2 x ~5 us of launch boundary + the in-kernel total of the
// previous launch's partials = 14 us per iteration for ~1 us of arithmetic, and at N > 1 that solve is replicated on every rank, so
// it is the part of the step that does not shard.  Here at most 250 workgroups (one per CU) stay resident for the whole solve.  A workgroup owns
// CP consecutive 256-cell chunks (the chunks and partial slots of the fused kernels); x, s and z of its cells AND of every cell its
// stencils read (its "window": own cells, then the halo cells, then one cell that holds zeros for the fill value of closed
// boundaries) live in LDS, r, q, the nine weights and the window indices of the eight neighbours in registers, for all iterations.
// What crosses workgroups per iteration: the chunk partials of the two inner products and z of the cells in somebody's halo.
// The search direction and the solution at halo cells are advanced by the workgroup itself with the owner's arithmetic (the rule of
// the distributed solvers), so they never travel.
//
// Every exchanged item is ONE aligned 16-byte word {value, tag} written by one 16-byte store and read by one 16-byte load, both at
// agent scope (sc1: through to / from memory, no cache in between): the tag is the number of the exchange (the launch's epoch in the
// high half, the phase or iteration in the low half) and never repeats, so a reader takes a value only if its tag is the one it waits
// for -- a stale word of an earlier phase, iteration or solve cannot be mistaken for it, nothing has to be reset, and no ordering
// between two different words is assumed.  (The first form of this kernel used 8-byte words that were their own flag, reset to an
// "empty" pattern one phase ahead: correct only if the reset became visible before a later reader polled -- an ordering between two
// words that relaxed accesses do not promise.  A 16-byte aligned access of one lane is one memory transaction on this hardware.)
// The words of phase n live in buffer n mod 2: a workgroup can be in phase n + 1 only after everybody has written phase n, i.e. has
// finished reading phase n - 1.  Collecting all partials of a phase is therefore also the grid-wide barrier of the phase.
//
// Same numbers as the fused launches, bit for bit (tests/test_gpu_parity.py::test_persistent_pcg_is_bitwise_the_fused_pcg): the
// chunk partials are formed by the tree of wg_reduce_store, their total by the rule of fused_total (thread-strided left-to-right
// sums per block, fixed tree, blocks in order), the cell arithmetic in the order of k_fpcg_a / k_fpcg_b / k_fpcg_xr / k_fresidual.
// A wait that does not end gives up after PersistArgs::wait_ticks of wall-clock time (2 s), marks the workgroup dead -- no later wait
// of it spins -- and the solve reports it (status word): never a hung GPU.  The host then restores the first guess, solves with the
// two-launch form, says so on stderr and does not use the resident form again in that model (solver_pcg_fused).  Seen so far only
// with several PROCESSES sharing one GPU (the multi-rank rehearsals of tests/), where the residency of all workgroups at once that
// the waits assume is not this kernel's to guarantee.
#pragma once
#include "kernels_barotropic.hpp"

namespace pop {

constexpr int POP_PERSIST_MAXP = 8;      // partial slots one thread collects per phase: nblocks * ceil(nchunk / 256) must not exceed it
constexpr int POP_PERSIST_MAXH = 6;      // halo cells one thread fetches per iteration: ceil(nhalo / 256)  (8 + 6 loads and their operands fit one asm block)

struct alignas(16) PWord { double v; unsigned long long tag; };

struct PersistArgs {
  double *X; const double *Bv, *C, *WNo, *WEa, *WNE; const unsigned char *mMask8;
  int nxb, nchunk, nblocks, nslots;      // partial slot = block * nchunk + chunk
  long long ncell;                       // n2 * nblocks
  const int *own_q;                      // [nwg * CP * 256] cell of own position L = u * 256 + t; -1: not a physical (interior) cell
  const unsigned short *nbr;             // [nwg * CP * 256 * 8] window index of the eight stencil neighbours (order of k_fpcg_b)
  const int *halo_off, *halo_q;          // halo cells of workgroup w: halo_q[halo_off[w] .. halo_off[w+1]), window index CP * 256 + h
  PWord *W;                              // one buffer: [2][nslots] partials, then [2][ncell] z (byte offsets fit 32 bits)
  unsigned long long epoch;              // high half of every tag of this launch
  int max_iter, freq;
  double criterion;
  unsigned long long wait_ticks;         // a wait for another workgroup gives up after this many ticks of the 100 MHz wall clock
  double *out;                           // pinned: [0] iterations, [1] last (r,r), [2] 0 ok / 1 a wait gave up, [3] checks done
};

typedef unsigned pword4 __attribute__((ext_vector_type(4)));
// one 16-byte agent-scope store of {value, tag}
__device__ __forceinline__ void st_pword(PWord *p, double v, unsigned long long tag) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  pword4 x;
  x[0] = (unsigned)b; x[1] = (unsigned)(b >> 32); x[2] = (unsigned)tag; x[3] = (unsigned)(tag >> 32);
  // (s_nop: a store of more than 8 bytes must not have its data registers overwritten in the next cycles -- the compiler's hazard
  // recogniser inserts that wait state for its own stores but does not look inside an asm block)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" : : "v"(p), "v"(x) : "memory");
}
// fourteen 16-byte agent-scope loads at base + 32-bit byte offsets, requested together, one wait
__device__ __forceinline__ void ld_pwords14(const PWord *base, const unsigned (&off)[14], pword4 (&o)[14]) {
  asm volatile(
      "global_load_dwordx4 %0, %14, %28 sc1\n\tglobal_load_dwordx4 %1, %15, %28 sc1\n\tglobal_load_dwordx4 %2, %16, %28 sc1\n\t"
      "global_load_dwordx4 %3, %17, %28 sc1\n\tglobal_load_dwordx4 %4, %18, %28 sc1\n\tglobal_load_dwordx4 %5, %19, %28 sc1\n\t"
      "global_load_dwordx4 %6, %20, %28 sc1\n\tglobal_load_dwordx4 %7, %21, %28 sc1\n\tglobal_load_dwordx4 %8, %22, %28 sc1\n\t"
      "global_load_dwordx4 %9, %23, %28 sc1\n\tglobal_load_dwordx4 %10, %24, %28 sc1\n\tglobal_load_dwordx4 %11, %25, %28 sc1\n\t"
      "global_load_dwordx4 %12, %26, %28 sc1\n\tglobal_load_dwordx4 %13, %27, %28 sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]),
        "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13])
      : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]), "v"(off[8]), "v"(off[9]),
        "v"(off[10]), "v"(off[11]), "v"(off[12]), "v"(off[13]), "s"(base)
      : "memory");
}
__device__ __forceinline__ double pword_value(const pword4 &x) { return __longlong_as_double((long long)(((unsigned long long)x[1] << 32) | x[0])); }
__device__ __forceinline__ unsigned long long pword_tag(const pword4 &x) { return ((unsigned long long)x[3] << 32) | x[2]; }
// Polls fourteen words (ld_pwords14) until every one a thread needs carries the tag it waits for.  The bound is wall-clock time (100 MHz
// counter), not a number of polls: after `ticks` the workgroup is marked dead (sticky, in LDS) and neither this nor any later wait of it spins.
__device__ __forceinline__ void wait_pwords14(const PWord *W, const unsigned (&off)[14], const bool (&need)[14], const unsigned long long (&want)[14],
                                              pword4 (&got)[14], int *dead, unsigned long long ticks) {
  unsigned long long t0 = 0;
  for (int tries = 0;; ++tries) {
    ld_pwords14(W, off, got);
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 14; ++k) ok = ok && (!need[k] || pword_tag(got[k]) == want[k]);
    if (ok) break;
    if (*(volatile int *)dead) break;
    if ((tries & 255) == 0) {
      const unsigned long long now = wall_clock64();
      if (tries == 0) t0 = now;
      else if (now - t0 > ticks) { *dead = 1; break; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int CP>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_pcg_persist(PersistArgs a) {
  extern __shared__ double lds[];                          // Xw | Sw | Zw, nwin doubles each
  // the chunk trees and the trees of the blocks' totals in arrays of their own, the totals double-buffered by the parity of the exchange:
  // three barriers per exchange (values in, block sums in, totals out) -- a value is overwritten only two or more barriers after its last read
  __shared__ double sh[CP][POP_RED_THREADS];
  __shared__ double shb[POP_PERSIST_MAXP][POP_RED_THREADS];
  __shared__ double sh_tot[2][POP_PERSIST_MAXP];
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

  int phase = 0;                                           // number of the next exchange: its slots are in buffer phase % 2
  // ---- one exchange: the chunk partials of v[] out, everybody's partials in, their ordered total back (every thread the same
  //      value); with_z: also z of the halo cells of iteration m
  auto exchange = [&](double (&v)[CP], bool with_z, int m) -> double {
#pragma unroll
    for (int u = 0; u < CP; ++u) sh[u][t] = v[u];
    __syncthreads();
    const unsigned long long ptag = a.epoch | (unsigned long long)(unsigned)phase, ztag = a.epoch | (unsigned long long)(unsigned)m | 0x80000000ULL;
    const long long pbase = (long long)(phase & 1) * a.nslots;                       // in PWords from a.W
    const long long zbase = 2LL * a.nslots + (long long)(m & 1) * a.ncell;
    // the tree of wg_reduce_store over the 256 values of a chunk, one WAVE per chunk: lane l forms what the two LDS steps leave in
    // element l -- (v[l] + v[l+128]) + (v[l+64] + v[l+192]) -- and the wave finishes with tree_tail64: no barrier between the steps,
    // and the chunks of a workgroup (the blocks of a view, below) go through their trees side by side instead of one after the other
    const int wave = t >> 6, lane = t & 63;
    for (int u = wave; u < CP; u += 4) {
      const double x = tree_tail64((sh[u][lane] + sh[u][lane + 128]) + (sh[u][lane + 64] + sh[u][lane + 192]));
      const int slot = w * CP + u;
      if (lane == 0 && slot < a.nslots) st_pword(a.W + pbase + slot, x, ptag);
    }
    // what this thread collects: slots b * nchunk + c, c = t, t + 256, ... of every block, and (with_z) its halo cells; a load it does
    // not need is aimed at the first word of the phase and accepted whatever its tag.  (Forming these once before the iteration loop
    // instead of in every exchange -- 22 more live registers -- was measured 1 us per iteration SLOWER: profiles/r04_ab_persist_shape.txt.)
    const int per_b = (a.nchunk - t + NT - 1) / NT;        // chunks of one block this thread adds (<= 0: none)
    unsigned off[14]; bool need[14]; unsigned long long want[14];
    int kb[POP_PERSIST_MAXP];
#pragma unroll
    for (int k = 0; k < POP_PERSIST_MAXP; ++k) {
      const int b = per_b > 0 ? k / per_b : a.nblocks, c = per_b > 0 ? t + (k % per_b) * NT : 0;
      need[k] = b < a.nblocks; want[k] = ptag; kb[k] = need[k] ? b : -1;
      off[k] = (unsigned)((pbase + (need[k] ? (long long)b * a.nchunk + c : 0)) * (long long)sizeof(PWord));
    }
    int halo_cell[POP_PERSIST_MAXH];
#pragma unroll
    for (int k = 0; k < POP_PERSIST_MAXH; ++k) {
      const int hh = t + k * NT;
      need[POP_PERSIST_MAXP + k] = with_z && hh < nhalo; want[POP_PERSIST_MAXP + k] = ztag;
      halo_cell[k] = hh < nhalo ? 0 : -1;
      off[POP_PERSIST_MAXP + k] = (unsigned)((need[POP_PERSIST_MAXP + k] ? zbase + a.halo_q[h0 + hh] : pbase) * (long long)sizeof(PWord));
    }
    pword4 got[14];
    wait_pwords14(a.W, off, need, want, got, &dead, a.wait_ticks);
    unsigned long long pv[POP_PERSIST_MAXP];
#pragma unroll
    for (int k = 0; k < POP_PERSIST_MAXP; ++k) pv[k] = need[k] ? (unsigned long long)__double_as_longlong(pword_value(got[k])) : 0ULL;
    if (with_z) {
#pragma unroll
      for (int k = 0; k < POP_PERSIST_MAXH; ++k) if (halo_cell[k] >= 0) Zw[NOWN + t + k * NT] = pword_value(got[POP_PERSIST_MAXP + k]);
    }
    // the rule of fused_total: per block the thread-strided left-to-right sum, the fixed tree, blocks in order -- the trees of all blocks
    // side by side (one set of barriers; with the eight bands of an 8-rank decomposition in one view, eight trees one after the other
    // were 5 us of every exchange)
    const int nb = a.nblocks, par = phase & 1;
    for (int b = 0; b < nb; ++b) {
      double x = 0.0;
#pragma unroll
      for (int k = 0; k < POP_PERSIST_MAXP; ++k)
        if (kb[k] == b) x = x + __longlong_as_double((long long)pv[k]);
      shb[b][t] = x;
    }
    __syncthreads();
    for (int b = wave; b < nb; b += 4) {
      const double y = tree_tail64((shb[b][lane] + shb[b][lane + 128]) + (shb[b][lane + 64] + shb[b][lane + 192]));
      if (lane == 0) sh_tot[par][b] = y;
    }
    __syncthreads();
    double total = 0.0;
    for (int b = 0; b < nb; ++b) total = total + sh_tot[par][b];
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
    PWord *const Zn = a.W + 2LL * a.nslots + (long long)(m & 1) * a.ncell;
    const unsigned long long ztag = a.epoch | (unsigned long long)(unsigned)m | 0x80000000ULL;
#pragma unroll
    for (int u = 0; u < CP; ++u) {
      v[u] = 0.0;
      if (inner[u]) {
        if (pending) r[u] = r[u] - alpha * qq[u];
        const double z = (cw[u] != 0.0) ? r[u] / cw[u] : 0.0;
        Zw[u * NT + t] = z;
        st_pword(Zn + q[u], z, ztag);
        v[u] = (r[u] * z) * mk[u];
      }
    }
    const double rz = exchange(v, true, m);
    // ---- step B (k_fpcg_b): [x += alpha s]; s = z + s beta at every window cell; q = A s; partial (q, s)
    const double bt = rz / eta0;
    eta1 = rz;
    for (int L = t; L < nwin - 1; L += NT) {               // one pass over the window: the pending x += alpha s, then the new s
      const double so = Sw[L];
      if (pending) Xw[L] = Xw[L] + alpha * so;
      Sw[L] = Zw[L] + so * bt;
    }
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


// ---- ChronGear (POP_SolversMod.F90:1960-2266, diagonal preconditioner) of a small 2-D system as one resident launch (round 4) ------------
// The iterations of solver_chrongear_fused (k_fcg_a | block sums | k_fcg_b per iteration, k_fresidual + k_rr_total per check) after its
// start-up pass, which stays as it is: this kernel takes over x, r, s, q, A0R and (rho, sigma) of the start-up.  Same windows, tags and
// buffers as k_pcg_persist.  Per iteration: z = r A0R of the own cells is published for the neighbours' halos and the neighbours' z
// collected (a wait for the NEIGHBOURS only), az = A z, the chunk partials of (r, z) and (az, z) are exchanged (the one grid-wide wait of
// the iteration; pcg has two), then s = z + beta s, q = az + beta q, x += alpha s, r -= alpha q with s and x advanced at the halo cells
// too (the owner's arithmetic).  Totals by the rule of fused_total2, chunk partials by the tree of wg_reduce_store<2>: bitwise the fused form.
struct CgPersistArgs {
  PersistArgs p;                         // X, Bv, C, weights, mask, plan, W (partials: [2][2 * nslots], then z), epoch, limits, out
  double *R, *S, *Q;                     // r, s, q of the start-up pass (not written back: nothing reads them after the solve)
  const double *A0R;
  const SolverScalars *sc;               // rho2[0], sigma2[0] of the start-up pass
};
constexpr int POP_CGP_MAXP = 7;          // partial slots per field one thread collects: 7 + 7 words in one request

template <int CP>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_cg_persist(CgPersistArgs ca) {
  const PersistArgs &a = ca.p;
  extern __shared__ double lds[];                          // Xw | Sw | Zw, nwin doubles each
  __shared__ double sh[2][CP][POP_RED_THREADS];            // (arrays and barriers as in k_pcg_persist)
  __shared__ double shb[2][POP_CGP_MAXP][POP_RED_THREADS];
  __shared__ double sh_tot[2][2][POP_CGP_MAXP];
  __shared__ int dead;
  constexpr int NT = POP_RED_THREADS, NOWN = CP * NT;
  const int t = threadIdx.x, w = blockIdx.x;
  const int h0 = a.halo_off[w], nhalo = a.halo_off[w + 1] - h0, nwin = NOWN + nhalo + 1;
  double *Xw = lds, *Sw = lds + nwin, *Zw = lds + 2 * nwin;
  if (t == 0) dead = 0;
  int q[CP]; bool inner[CP];
  double cw[CP], mk[CP], wv[CP][8], r[CP], sq_[CP], qv[CP], a0r[CP];
  unsigned short nb[CP][8];
#pragma unroll
  for (int u = 0; u < CP; ++u) {
    const long long L = (long long)w * NOWN + u * NT + t;
    q[u] = a.own_q[L];
    inner[u] = q[u] >= 0;
    const long long qc = inner[u] ? q[u] : 0;
    const int nxb = a.nxb;
    cw[u] = a.C[qc]; mk[u] = (double)a.mMask8[qc];
    const long long qs = (qc - nxb - 1 >= 0) ? qc : (long long)nxb + 1;
    wv[u][0] = a.WNo[qs]; wv[u][1] = a.WNo[qs - nxb]; wv[u][2] = a.WEa[qs]; wv[u][3] = a.WEa[qs - 1];
    wv[u][4] = a.WNE[qs]; wv[u][5] = a.WNE[qs - nxb]; wv[u][6] = a.WNE[qs - 1]; wv[u][7] = a.WNE[qs - 1 - nxb];
#pragma unroll
    for (int n = 0; n < 8; ++n) nb[u][n] = a.nbr[L * 8 + n];
    r[u] = inner[u] ? ca.R[qc] : 0.0; qv[u] = inner[u] ? ca.Q[qc] : 0.0; a0r[u] = ca.A0R[qc];
    sq_[u] = 0.0;
  }
  // window: x and s of the own and the halo cells as the start-up pass left them (ghost values there are copies of their sources)
  for (int L = t; L < nwin; L += NT) {
    int qL = -1;
    if (L < NOWN) qL = a.own_q[(long long)w * NOWN + L];
    else if (L < NOWN + nhalo) qL = a.halo_q[h0 + L - NOWN];
    Xw[L] = qL >= 0 ? a.X[qL] : 0.0; Sw[L] = qL >= 0 ? ca.S[qL] : 0.0; Zw[L] = 0.0;
  }
  double rho_o = ca.sc->rho2[0], sig_o = ca.sc->sigma2[0];
  __syncthreads();

  int phase = 0;
  // ---- z of the halo cells of iteration m (the neighbours published it with tag m)
  auto halo_z = [&](int m) {
    const unsigned long long ztag = a.epoch | (unsigned long long)(unsigned)m | 0x80000000ULL;
    const long long zbase = 4LL * a.nslots + (long long)(m & 1) * a.ncell;
    unsigned off[14]; bool need[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      const int hh = t + (k % POP_PERSIST_MAXH) * NT;
      need[k] = k < POP_PERSIST_MAXH && hh < nhalo;
      off[k] = (unsigned)((need[k] ? zbase + a.halo_q[h0 + hh] : zbase) * (long long)sizeof(PWord));
    }
    pword4 got[14];
    unsigned long long want[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) want[k] = ztag;
    wait_pwords14(a.W, off, need, want, got, &dead, a.wait_ticks);
#pragma unroll
    for (int k = 0; k < POP_PERSIST_MAXH; ++k) { const int hh = t + k * NT; if (hh < nhalo) Zw[NOWN + hh] = pword_value(got[k]); }
    __syncthreads();
  };
  // ---- the chunk partials of NF fields out, everybody's in, their ordered totals back (rule of fused_total2 / fused_total)
  auto exchange = [&](double (&v)[2][CP], int nf, double (&tot)[2]) {
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int u = 0; u < CP; ++u) sh[f][u][t] = v[f][u];
    __syncthreads();
    const unsigned long long ptag = a.epoch | (unsigned long long)(unsigned)phase;
    const long long pbase = (long long)(phase & 1) * 2 * a.nslots;               // [field][slot] inside the buffer of the phase
    const int wave = t >> 6, lane = t & 63;                                      // one wave per (field, chunk) tree: see k_pcg_persist
    for (int e = wave; e < 2 * CP; e += 4) {
      const int f = e / CP, u = e % CP;
      const double x = tree_tail64((sh[f][u][lane] + sh[f][u][lane + 128]) + (sh[f][u][lane + 64] + sh[f][u][lane + 192]));
      const int slot = w * CP + u;
      if (lane == 0 && slot < a.nslots && f < nf) st_pword(a.W + pbase + (long long)f * a.nslots + slot, x, ptag);
    }
    const int per_b = (a.nchunk - t + NT - 1) / NT;
    unsigned off[14]; bool need[14]; int kb[POP_CGP_MAXP];
#pragma unroll
    for (int k = 0; k < POP_CGP_MAXP; ++k) {
      const int b = per_b > 0 ? k / per_b : a.nblocks, c = per_b > 0 ? t + (k % per_b) * NT : 0;
      const bool nd = b < a.nblocks;
      kb[k] = nd ? b : -1;
      const long long sl = nd ? (long long)b * a.nchunk + c : 0;
      need[k] = nd; need[POP_CGP_MAXP + k] = nd && nf > 1;
      off[k] = (unsigned)((pbase + sl) * (long long)sizeof(PWord));
      off[POP_CGP_MAXP + k] = (unsigned)((pbase + (nf > 1 ? a.nslots : 0) + sl) * (long long)sizeof(PWord));
    }
    pword4 got[14];
    unsigned long long want[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) want[k] = ptag;
    wait_pwords14(a.W, off, need, want, got, &dead, a.wait_ticks);
    const int nbk = a.nblocks, par = phase & 1;
    for (int b = 0; b < nbk; ++b) {
      double x0 = 0.0, x1 = 0.0;
#pragma unroll
      for (int k = 0; k < POP_CGP_MAXP; ++k)
        if (kb[k] == b) { x0 = x0 + pword_value(got[k]); x1 = x1 + pword_value(got[POP_CGP_MAXP + k]); }
      shb[0][b][t] = x0; shb[1][b][t] = x1;
    }
    __syncthreads();
    for (int e = wave; e < 2 * nbk; e += 4) {
      const int f = e & 1, b = e >> 1;
      const double y = tree_tail64((shb[f][b][lane] + shb[f][b][lane + 128]) + (shb[f][b][lane + 64] + shb[f][b][lane + 192]));
      if (lane == 0) sh_tot[par][f][b] = y;
    }
    __syncthreads();
    tot[0] = 0.0; tot[1] = 0.0;
    for (int b = 0; b < nbk; ++b) { tot[0] = tot[0] + sh_tot[par][0][b]; tot[1] = tot[1] + sh_tot[par][1][b]; }
    ++phase;
  };
  auto apply = [&](const double *W, int u) -> double {
    double ax = cw[u] * W[u * NT + t];
#pragma unroll
    for (int n = 0; n < 8; ++n) ax = ax + wv[u][n] * W[nb[u][n]];
    return ax;
  };

  double rr = 0.0;
  int m = 0, nchecks = 0, converged = 0;
  double v[2][CP], tot[2];
  while (m < a.max_iter) {
    ++m;
    // ---- k_fcg_a: z = r A0R (own cells; the halo cells' from their owners), az = A z, partials (r, z), (az, z)
    PWord *const Zn = a.W + 4LL * a.nslots + (long long)(m & 1) * a.ncell;
    const unsigned long long ztag = a.epoch | (unsigned long long)(unsigned)m | 0x80000000ULL;
    double z[CP];
#pragma unroll
    for (int u = 0; u < CP; ++u) {
      z[u] = 0.0;
      if (inner[u]) {
        z[u] = r[u] * a0r[u];
        Zw[u * NT + t] = z[u];
        st_pword(Zn + q[u], z[u], ztag);
      }
    }
    halo_z(m);                                             // (ends with a barrier: the own z of every thread is in the window too)
    double az[CP];
#pragma unroll
    for (int u = 0; u < CP; ++u) {
      az[u] = 0.0; v[0][u] = 0.0; v[1][u] = 0.0;
      if (inner[u]) {
        az[u] = apply(Zw, u);
        v[0][u] = (r[u] * z[u]) * mk[u]; v[1][u] = (az[u] * z[u]) * mk[u];
      }
    }
    exchange(v, 2, tot);
    // ---- k_fcg_b: the scalar recurrences, then s, q, x, r; s and x at the halo cells with the owner's arithmetic
    const double rho = tot[0], delta = tot[1];
    const double bt = rho / rho_o;
    const double sigma = delta - (bt * bt) * sig_o;
    const double al = rho / sigma;
    rho_o = rho; sig_o = sigma;
#pragma unroll
    for (int u = 0; u < CP; ++u)
      if (inner[u]) {
        const double s = z[u] + bt * Sw[u * NT + t];
        qv[u] = az[u] + bt * qv[u];
        Sw[u * NT + t] = s;
        Xw[u * NT + t] = Xw[u * NT + t] + al * s;
        r[u] = r[u] - al * qv[u];
      }
    for (int L = NOWN + t; L < nwin - 1; L += NT) {        // halo cells: sg = z + beta sg; x += alpha sg
      const double sg = Zw[L] + bt * Sw[L];
      Sw[L] = sg;
      Xw[L] = Xw[L] + al * sg;
    }
    __syncthreads();
    if (m % a.freq == 0) {
      // ---- convergence check: r = b - A x, (r, r)
#pragma unroll
      for (int u = 0; u < CP; ++u) {
        r[u] = inner[u] ? a.Bv[q[u]] - apply(Xw, u) : 0.0;
        v[0][u] = inner[u] ? (r[u] * r[u]) * mk[u] : 0.0; v[1][u] = 0.0;
      }
      exchange(v, 1, tot);
      rr = tot[0];
      ++nchecks;
      if (rr < a.criterion) { converged = 1; break; }
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < CP; ++u) if (inner[u]) a.X[q[u]] = Xw[u * NT + t];
  if (w == 0 && t == 0) { a.out[0] = converged ? (double)m : (double)a.max_iter; a.out[1] = rr; a.out[3] = (double)nchecks; }
  if (t == 0 && dead) a.out[2] = 1.0;
  (void)sq_;
}


// ---- P-CSI (POP_SolversMod.F90:1510-1835, diagonal preconditioner) of a small 2-D system as one resident launch (round 4) ----------------
// The iterations of solver_pcsi_fused after its start-up step.  P-CSI has no inner product: per iteration dx = omega_k r' + (gamma omega_k
// - 1) dx, x += dx, r = b - A x, r' = r / diag -- so what crosses workgroups per iteration is r' of the cells in somebody's halo, and a
// workgroup waits for its NEIGHBOURS only (tagged words, two buffers by the parity of the iteration: a workgroup can be one iteration
// ahead of a neighbour, never two, because it needs that neighbour's r' of the iteration before).  dx and x at the halo cells are advanced
// by the reader with the owner's arithmetic, as k_pcsi_step does for its eight neighbours.  Only the convergence checks (every `freq`
// iterations from `start` on) are grid-wide: the chunk partials of (r, r) by the tree of wg_reduce_store, their total by the rule of
// k_rr_total.  Bitwise the fused form (tests/test_gpu_parity.py::test_persistent_pcg_is_bitwise_the_fused_pcg).
struct PcsiPersistArgs {
  PersistArgs p;                         // X (out: the solution array), Bv, C, weights, mask, plan, W, epoch, limits, out
  const double *Xin, *Rin, *Qin;         // x, r' and dx after the start-up step
  const double *A0R, *omega;             // 1 / diag; omega_k (1-based)
  double csy;
  int start;                             // first iteration count at which a check is made (convergenceCheckStart)
};

template <int CP>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_pcsi_persist(PcsiPersistArgs pa) {
  const PersistArgs &a = pa.p;
  extern __shared__ double lds[];                          // Xw | Qw | Rw, nwin doubles each
  __shared__ double sh[CP][POP_RED_THREADS];
  __shared__ double shb[POP_PERSIST_MAXP][POP_RED_THREADS];
  __shared__ double sh_tot[2][POP_PERSIST_MAXP];
  __shared__ int dead;
  constexpr int NT = POP_RED_THREADS, NOWN = CP * NT;
  const int t = threadIdx.x, w = blockIdx.x;
  const int h0 = a.halo_off[w], nhalo = a.halo_off[w + 1] - h0, nwin = NOWN + nhalo + 1;
  double *Xw = lds, *Qw = lds + nwin, *Rw = lds + 2 * nwin;
  if (t == 0) dead = 0;
  int q[CP]; bool inner[CP];
  double cw[CP], mk[CP], wv[CP][8], bv[CP], a0r[CP];
  unsigned short nb[CP][8];
#pragma unroll
  for (int u = 0; u < CP; ++u) {
    const long long L = (long long)w * NOWN + u * NT + t;
    q[u] = a.own_q[L];
    inner[u] = q[u] >= 0;
    const long long qc = inner[u] ? q[u] : 0;
    const int nxb = a.nxb;
    cw[u] = a.C[qc]; mk[u] = (double)a.mMask8[qc]; bv[u] = a.Bv[qc]; a0r[u] = pa.A0R[qc];
    const long long qs = (qc - nxb - 1 >= 0) ? qc : (long long)nxb + 1;
    wv[u][0] = a.WNo[qs]; wv[u][1] = a.WNo[qs - nxb]; wv[u][2] = a.WEa[qs]; wv[u][3] = a.WEa[qs - 1];
    wv[u][4] = a.WNE[qs]; wv[u][5] = a.WNE[qs - nxb]; wv[u][6] = a.WNE[qs - 1]; wv[u][7] = a.WNE[qs - 1 - nxb];
#pragma unroll
    for (int n = 0; n < 8; ++n) nb[u][n] = a.nbr[L * 8 + n];
  }
  for (int L = t; L < nwin; L += NT) {
    int qL = -1;
    if (L < NOWN) qL = a.own_q[(long long)w * NOWN + L];
    else if (L < NOWN + nhalo) qL = a.halo_q[h0 + L - NOWN];
    Xw[L] = qL >= 0 ? pa.Xin[qL] : 0.0; Qw[L] = qL >= 0 ? pa.Qin[qL] : 0.0; Rw[L] = qL >= 0 ? pa.Rin[qL] : 0.0;
  }
  __syncthreads();

  int phase = 0;
  // r' of the halo cells as published in iteration m (tag m)
  auto halo_r = [&](int m) {
    const unsigned long long ztag = a.epoch | (unsigned long long)(unsigned)m | 0x80000000ULL;
    const long long zbase = 4LL * a.nslots + (long long)(m & 1) * a.ncell;
    unsigned off[14]; bool need[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      const int hh = t + (k % POP_PERSIST_MAXH) * NT;
      need[k] = k < POP_PERSIST_MAXH && hh < nhalo;
      off[k] = (unsigned)((need[k] ? zbase + a.halo_q[h0 + hh] : zbase) * (long long)sizeof(PWord));
    }
    pword4 got[14];
    unsigned long long want[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) want[k] = ztag;
    wait_pwords14(a.W, off, need, want, got, &dead, a.wait_ticks);
#pragma unroll
    for (int k = 0; k < POP_PERSIST_MAXH; ++k) { const int hh = t + k * NT; if (hh < nhalo) Rw[NOWN + hh] = pword_value(got[k]); }
  };
  // the grid-wide total of the chunk partials of v (the checks): as k_pcg_persist's exchange without halo words
  auto exchange = [&](double (&v)[CP]) -> double {
#pragma unroll
    for (int u = 0; u < CP; ++u) sh[u][t] = v[u];
    __syncthreads();
    const unsigned long long ptag = a.epoch | (unsigned long long)(unsigned)phase;
    const long long pbase = (long long)(phase & 1) * 2 * a.nslots;
    const int wave = t >> 6, lane = t & 63;
    for (int u = wave; u < CP; u += 4) {
      const double x = tree_tail64((sh[u][lane] + sh[u][lane + 128]) + (sh[u][lane + 64] + sh[u][lane + 192]));
      const int slot = w * CP + u;
      if (lane == 0 && slot < a.nslots) st_pword(a.W + pbase + slot, x, ptag);
    }
    const int per_b = (a.nchunk - t + NT - 1) / NT;
    unsigned off[14]; bool need[14]; int kb[POP_PERSIST_MAXP];
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      const int kk = k < POP_PERSIST_MAXP ? k : 0;
      const int b = per_b > 0 ? kk / per_b : a.nblocks, c = per_b > 0 ? t + (kk % per_b) * NT : 0;
      need[k] = k < POP_PERSIST_MAXP && b < a.nblocks;
      if (k < POP_PERSIST_MAXP) kb[k] = need[k] ? b : -1;
      off[k] = (unsigned)((pbase + (need[k] ? (long long)b * a.nchunk + c : 0)) * (long long)sizeof(PWord));
    }
    pword4 got[14];
    unsigned long long want[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) want[k] = ptag;
    wait_pwords14(a.W, off, need, want, got, &dead, a.wait_ticks);
    const int nbk = a.nblocks, par = phase & 1;
    for (int b = 0; b < nbk; ++b) {
      double x = 0.0;
#pragma unroll
      for (int k = 0; k < POP_PERSIST_MAXP; ++k)
        if (kb[k] == b) x = x + pword_value(got[k]);
      shb[b][t] = x;
    }
    __syncthreads();
    for (int b = wave; b < nbk; b += 4) {
      const double y = tree_tail64((shb[b][lane] + shb[b][lane + 128]) + (shb[b][lane + 64] + shb[b][lane + 192]));
      if (lane == 0) sh_tot[par][b] = y;
    }
    __syncthreads();
    double total = 0.0;
    for (int b = 0; b < nbk; ++b) total = total + sh_tot[par][b];
    ++phase;
    return total;
  };
  auto apply = [&](const double *W, int u) -> double {
    double ax = cw[u] * W[u * NT + t];
#pragma unroll
    for (int n = 0; n < 8; ++n) ax = ax + wv[u][n] * W[nb[u][n]];
    return ax;
  };

  double rr = 0.0;
  int m = 0, nchecks = 0, converged = 0;
  double v[CP];
  while (m < a.max_iter) {
    ++m;
    // dx = omega r' + (gamma omega - 1) dx; x += dx -- at every window cell (own and halo: the owner's arithmetic)
    const double om = pa.omega[m], cq = pa.csy * om - 1.0;
    for (int L = t; L < nwin - 1; L += NT) {
      const double dx = om * Rw[L] + cq * Qw[L];
      Qw[L] = dx;
      Xw[L] = Xw[L] + dx;
    }
    __syncthreads();
    // r = b - A x, r' = r / diag at the own cells; r' published for the neighbours' halos
    PWord *const Rn = a.W + 4LL * a.nslots + (long long)(m & 1) * a.ncell;
    const unsigned long long ztag = a.epoch | (unsigned long long)(unsigned)m | 0x80000000ULL;
    const bool check = (m % a.freq == 0) && m >= pa.start;
#pragma unroll
    for (int u = 0; u < CP; ++u) {
      v[u] = 0.0;
      if (inner[u]) {
        const double r = bv[u] - apply(Xw, u);
        const double rp = r * a0r[u];
        Rw[u * NT + t] = rp;
        st_pword(Rn + q[u], rp, ztag);
        v[u] = (r * r) * mk[u];
      }
    }
    if (check) {
      rr = exchange(v);
      ++nchecks;
      if (rr < a.criterion) { converged = 1; break; }
    }
    halo_r(m);
    __syncthreads();
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < CP; ++u) if (inner[u]) a.X[q[u]] = Xw[u * NT + t];
  if (w == 0 && t == 0) { a.out[0] = converged ? (double)m : (double)a.max_iter; a.out[1] = rr; a.out[3] = (double)nchecks; }
  if (t == 0 && dead) a.out[2] = 1.0;
}

}  // namespace pop
