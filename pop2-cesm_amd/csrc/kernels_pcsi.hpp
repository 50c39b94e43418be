// kernels_pcsi.hpp -- P-CSI (Preconditioned Classical Stiefel Iteration), diagonal preconditioner:
// POP_SolversMod.F90:1510-1835.  The iteration needs no inner product: per step
//     r' = r * (1/diag);  halo(r');  dx = omega_k r' + (gamma omega_k - 1) dx;  x += dx;  r = b - A x
// and (r,r) only every convergenceCheckFreq steps from convergenceCheckStart on.
//
// Fused form (all blocks on this GPU): ONE launch per iteration.  A thread forms the updated x of its
// cell and of its 8 neighbours (each neighbour's update is recomputed with exactly the operations its
// own thread uses, ghosts read at their source cell through srcmap = the value a halo update would
// deliver), applies the 9-point operator and writes r, dx, x into the other half of a ping-pong pair,
// so no thread reads what another writes in the same launch.  The omega_k sequence depends only on the
// Lanczos eigenvalue bounds: it is tabulated once on the device and indexed by base + j, where j is
// baked into the launch and `base` is a device word the host bumps per interval -- so one hipGraph
// serves every interval of every solve.
// Round 4: on large grids TWO iterations per launch where no check follows (k_pcsi_step_x2, below); on small grids the whole
// iteration loop as one resident launch (k_pcsi_persist, kernels_pcg_persist.hpp).
#pragma once
#include "kernels_barotropic.hpp"

namespace pop {

struct PcsiArgs {
  const double *Xi, *Ri, *Qi;      // state in
  double *Xo, *Ro, *Qo;            // state out
  const double *Bv, *C, *A0R;      // A0R = 1/diag (k_pcsi_a0r)
  const double *omega;             // 1-based table of omega_k; entry 0 = 1/gamma (the start-up step, :1664)
  const int *base;                 // iteration number of the interval's first step, minus 1
  const int *srcmap;
  double *partial;
  const SolverScalars *sc;         // stop flag of the look-ahead scheme (kernels_barotropic.hpp)
  double csy;
  int j;                           // step inside the interval (1-based); j = 0: start-up step
  int remote_ghosts;               // multi-rank: also advance dx, x at ghosts owned by other ranks
  int nchunk;                      // stride of the partial slots per block (the launch may be compacted: DevGrid::red_act)
  int raw_r;                       // EVP preconditioner (r3): Ro receives the residual itself; k_evp_apply_wave2 turns it into r' for the next step
  const int *jfold;                // k_pcsi_step_x2<., true>: per block, the first array row (0-based) beyond a tripole fold; nyb where the block does not touch it
};

// unfused building blocks (multi-rank path and cross-check): whole-array operations as the reference has them
__global__ void k_pcsi_precond(DevGrid g, double *__restrict__ R, const double *__restrict__ C, long long n) {   // :1705-1712
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const double a0r = (C[q] != 0.0) ? 1.0 / C[q] : 0.0;
  R[q] = R[q] * a0r;
}
template <bool FIRST>
__global__ void k_pcsi_update(const double *__restrict__ R, double *__restrict__ Q, double *__restrict__ X, long long n,
                              const double *__restrict__ omega, const int *__restrict__ base, int j, double csy) {   // :1664, :1742-1745
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  double dx;
  if (FIRST) dx = omega[0] * R[q];
  else { const double om = omega[*base + j]; dx = om * R[q] + (csy * om - 1.0) * Q[q]; }
  Q[q] = dx;
  X[q] = X[q] + dx;
}

// 1/diag (0 where the diagonal vanishes), formed once per solve as the reference does (:1592-1605)
__global__ void k_pcsi_a0r(const double *__restrict__ C, double *__restrict__ A0R, long long n) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  A0R[q] = (C[q] != 0.0) ? 1.0 / C[q] : 0.0;
}
// r = b - A x of the fused residual kernel, scaled in place to r' = r * (1/diag)
__global__ void k_pcsi_scale(DevGrid g, double *__restrict__ R, const double *__restrict__ A0R) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (p2 >= g.n2) return;
  const int i = p2 % g.nxb, j = p2 / g.nxb;
  if (!interior(g, b, i, j)) return;
  const long long q = (long long)b * g.n2 + p2;
  R[q] = R[q] * A0R[q];
}

// fused step.  The residual arrays hold the PRECONDITIONED residual r' = r * (1/diag): the product the
// reference forms at the top of the next iteration (:1705-1712) is formed here by the thread that just
// computed r, so a neighbour's update needs three loads (r', dx, x) and no division.
template <bool FIRST, bool WITH_RR>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_pcsi_step(DevGrid g, PcsiArgs a) {
  if (a.sc->stop) return;   // an earlier check has converged: this launch belongs to the look-ahead interval
  // land elimination: x, dx and r' stay exactly 0 where no ocean cell is near, in both halves of the ping-pong buffers
  if (WITH_RR ? red_land_out<1>(g, a.partial, a.nchunk, a.remote_ghosts != 0) : red_land(g, a.remote_ghosts != 0)) return;
  const int p2 = red_cell(g), b = blockIdx.y, nxb = g.nxb;
  double v[1] = {0.0};
  if (p2 < g.n2) {
    const int i = p2 % nxb, j = p2 / nxb;
    const long long q = (long long)b * g.n2 + p2;
    if (interior(g, b, i, j)) {
      const double om = FIRST ? a.omega[0] : a.omega[*a.base + a.j];
      const double cq = FIRST ? 0.0 : a.csy * om - 1.0;
      const bool rim = (i + 1 == g.ib || i + 1 == blk_ie(g, b) || j + 1 == g.jb || j + 1 == blk_je(g, b));
      const int off[8] = {nxb, -nxb, 1, -1, nxb + 1, -nxb + 1, nxb - 1, -nxb - 1};
      // gather first: every load is independent
      double rp[9], qo[9], xo[9];
      long long m[9];
      m[0] = q;
#pragma unroll
      for (int t = 0; t < 8; ++t) { long long mm = q + off[t]; if (rim) mm = a.srcmap[mm]; m[t + 1] = mm; }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const bool ok = m[t] >= 0;
        const long long mm = ok ? m[t] : q;
        rp[t] = a.Ri[mm]; xo[t] = a.Xi[mm]; qo[t] = FIRST ? 0.0 : a.Qi[mm];
        if (!ok) { rp[t] = 0.0; xo[t] = 0.0; qo[t] = 0.0; }
      }
      const double w[9] = {a.C[q], g.WNo[q], g.WNo[q - nxb], g.WEa[q], g.WEa[q - 1], g.WNE[q], g.WNE[q - nxb], g.WNE[q - 1], g.WNE[q - 1 - nxb]};
      const double bq = a.Bv[q], a0r = a.A0R[q];
      double xn[9], dx0 = 0.0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const double dx = FIRST ? om * rp[t] : om * rp[t] + cq * qo[t];
        if (t == 0) dx0 = dx;
        xn[t] = (m[t] >= 0) ? xo[t] + dx : 0.0;
      }
      // btropOperator :2414-2426, same order as btrop_op
      const double ax = w[0] * xn[0] + w[1] * xn[1] + w[2] * xn[2] + w[3] * xn[3] + w[4] * xn[4] + w[5] * xn[5] + w[6] * xn[6] + w[7] * xn[7] + w[8] * xn[8];
      const double r = bq - ax;
      a.Qo[q] = dx0; a.Xo[q] = xn[0]; a.Ro[q] = a.raw_r ? r : r * a0r;
      if (WITH_RR) v[0] = (r * r) * (double)g.mMask8[q];
    } else if (a.remote_ghosts && a.srcmap[q] == q) {
      // ghost owned by another rank (multi-rank fused form): its r' arrived by the halo exchange; dx and x
      // are advanced here with the owner's arithmetic, so they never need to be exchanged
      const double om = FIRST ? a.omega[0] : a.omega[*a.base + a.j];
      const double rp = a.Ri[q];
      const double dx = FIRST ? om * rp : om * rp + (a.csy * om - 1.0) * a.Qi[q];
      a.Qo[q] = dx; a.Xo[q] = a.Xi[q] + dx;
    }
  }
  if (WITH_RR) wg_reduce_store<1>(v, a.partial, b * a.nchunk + red_chunk(g));
}

// fused step with two horizontally adjacent cells per thread (large grids with an even row pitch; not the first
// step): the pair shares its three stencil rows of (r', dx, x), read as (q-1), (q, q+1) in one 16-byte load, (q+2).
// Same operations per cell in the same order and the same reduction tree as k_pcsi_step: bitwise equal.
// RAWR: the EVP form (PcsiArgs::raw_r, host: every launch of a model with that preconditioner) -- 1/diag is not read
template <bool WITH_RR, bool RAWR = false>
__global__ void __launch_bounds__(POP_RED_THREADS / 2)
k_pcsi_step2(DevGrid g, PcsiArgs a) {
  if (a.sc->stop) return;   // an earlier check has converged: this launch belongs to the look-ahead interval
  // land elimination: x, dx and r' stay exactly 0 where no ocean cell is near, in both halves of the ping-pong buffers
  if (WITH_RR ? red_land_out<1>(g, a.partial, a.nchunk, a.remote_ghosts != 0) : red_land(g, a.remote_ghosts != 0)) return;
  __shared__ double sh[POP_RED_THREADS];
  const int b = blockIdx.y, t = threadIdx.x, nxb = g.nxb;
  const long long p0 = (long long)red_chunk(g) * POP_RED_THREADS + 2 * t;
  const bool live0 = p0 < g.n2, live1 = p0 + 1 < g.n2;
  const int pp = live0 ? (int)p0 : 0;
  const int i = pp % nxb, j = pp / nxb;
  const long long q = (long long)b * g.n2 + pp;
  const bool fast = live1 && i + 1 > g.ib && i + 2 < blk_ie(g, b) && j + 1 > g.jb && j + 1 < blk_je(g, b);
  const double om = a.omega[*a.base + a.j];
  const double cq = a.csy * om - 1.0;
  double v0 = 0.0, v1 = 0.0;
  if (fast) {
    double xn[3][4], dxc[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const long long o = q + (long long)(r - 1) * nxb;
      const double2 rc = *reinterpret_cast<const double2 *>(a.Ri + o), qc = *reinterpret_cast<const double2 *>(a.Qi + o),
                    xc = *reinterpret_cast<const double2 *>(a.Xi + o);
      const double d0 = om * a.Ri[o - 1] + cq * a.Qi[o - 1], d1 = om * rc.x + cq * qc.x, d2 = om * rc.y + cq * qc.y,
                   d3 = om * a.Ri[o + 2] + cq * a.Qi[o + 2];
      xn[r][0] = a.Xi[o - 1] + d0; xn[r][1] = xc.x + d1; xn[r][2] = xc.y + d2; xn[r][3] = a.Xi[o + 2] + d3;
      if (r == 1) { dxc[0] = d1; dxc[1] = d2; }
    }
    const double2 cc = *reinterpret_cast<const double2 *>(a.C + q);
    // the off-centre weights from their two U-point terms, as k_fpcg_b2 forms them (two fields instead of three; WNE = xne + yne,
    // WEa = xne + xse - yne - yse, WNo = yne + ynw - xne - xnw: host_setup.cpp, the same additions in the same order)
    const double2 x0 = *reinterpret_cast<const double2 *>(g.XW + q), xm = *reinterpret_cast<const double2 *>(g.XW + q - nxb);
    const double2 y0 = *reinterpret_cast<const double2 *>(g.YW + q), ym = *reinterpret_cast<const double2 *>(g.YW + q - nxb);
    const double x0w = g.XW[q - 1], xmw = g.XW[q - 1 - nxb], y0w = g.YW[q - 1], ymw = g.YW[q - 1 - nxb];
    double2 no0, nom, ea0, ne0, nem;
    const double ne0w = x0w + y0w, nemw = xmw + ymw;
    ne0.x = x0.x + y0.x; ne0.y = x0.y + y0.y; nem.x = xm.x + ym.x; nem.y = xm.y + ym.y;
    const double eaw = x0w + xmw - y0w - ymw;
    ea0.x = x0.x + xm.x - y0.x - ym.x; ea0.y = x0.y + xm.y - y0.y - ym.y;
    no0.x = y0.x + y0w - x0.x - x0w; no0.y = y0.y + y0.x - x0.y - x0.x;
    nom.x = ym.x + ymw - xm.x - xmw; nom.y = ym.y + ym.x - xm.y - xm.x;
    const double2 bq = *reinterpret_cast<const double2 *>(a.Bv + q);
    double2 a0r = make_double2(0.0, 0.0);
    if (!RAWR) a0r = *reinterpret_cast<const double2 *>(a.A0R + q);
    const double axA = cc.x * xn[1][1] + no0.x * xn[2][1] + nom.x * xn[0][1] + ea0.x * xn[1][2] + eaw * xn[1][0] +
                       ne0.x * xn[2][2] + nem.x * xn[0][2] + ne0w * xn[2][0] + nemw * xn[0][0];
    const double axB = cc.y * xn[1][2] + no0.y * xn[2][2] + nom.y * xn[0][2] + ea0.y * xn[1][3] + ea0.x * xn[1][1] +
                       ne0.y * xn[2][3] + nem.y * xn[0][3] + ne0.x * xn[2][1] + nem.x * xn[0][1];
    const double rA = bq.x - axA, rB = bq.y - axB;
    *reinterpret_cast<double2 *>(a.Qo + q) = make_double2(dxc[0], dxc[1]);
    *reinterpret_cast<double2 *>(a.Xo + q) = make_double2(xn[1][1], xn[1][2]);
    *reinterpret_cast<double2 *>(a.Ro + q) = (RAWR || a.raw_r) ? make_double2(rA, rB) : make_double2(rA * a0r.x, rB * a0r.y);
    if (WITH_RR) { v0 = (rA * rA) * (double)g.mMask8[q]; v1 = (rB * rB) * (double)g.mMask8[q + 1]; }
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
      long long m[9];
      m[0] = qn;
#pragma unroll
      for (int n = 1; n < 9; ++n) m[n] = (long long)a.srcmap[qn + off[n - 1]];
      const int self = a.srcmap[qq];
      const double w[9] = {a.C[qn], g.WNo[qn], g.WNo[qn - nxb], g.WEa[qn], g.WEa[qn - 1], g.WNE[qn], g.WNE[qn - nxb], g.WNE[qn - 1], g.WNE[qn - 1 - nxb]};
      const double bvv = a.Bv[qn], a0r = a.A0R[qn], mk = (double)g.mMask8[qn];
      const double ri_own = a.Ri[qq], qi_own = a.Qi[qq], xi_own = a.Xi[qq];       // a ghost cell that advances itself (remote_ghosts)
      double xn[9], dx0 = 0.0;
#pragma unroll
      for (int n = 0; n < 9; ++n) {
        const long long mm = (m[n] >= 0) ? m[n] : qn;
        const double dx = om * a.Ri[mm] + cq * a.Qi[mm];
        const double x = a.Xi[mm] + dx;
        if (n == 0) dx0 = dx;
        xn[n] = (m[n] >= 0) ? x : 0.0;
      }
      if (in_e) {
        const double ax = w[0] * xn[0] + w[1] * xn[1] + w[2] * xn[2] + w[3] * xn[3] + w[4] * xn[4] + w[5] * xn[5] + w[6] * xn[6] + w[7] * xn[7] + w[8] * xn[8];
        const double r = bvv - ax;
        a.Qo[qq] = dx0; a.Xo[qq] = xn[0]; a.Ro[qq] = a.raw_r ? r : r * a0r;
        if (WITH_RR) { const double vv = (r * r) * mk; if (e == 0) v0 = vv; else v1 = vv; }
      } else if (live && a.remote_ghosts && self == (int)qq) {
        const double dx = om * ri_own + cq * qi_own;
        a.Qo[qq] = dx; a.Xo[qq] = xi_own + dx;
      }
    }
  }
  if (WITH_RR) {   // the tree of wg_reduce_store<1> over the 256 cells of the chunk
    sh[2 * t] = v0; sh[2 * t + 1] = v1;
    __syncthreads();
    for (int s = POP_RED_THREADS / 2; s > 0; s >>= 1) {
      if (t < s) sh[t] = sh[t] + sh[t + s];
      __syncthreads();
    }
    if (t == 0) a.partial[(long long)b * a.nchunk + red_chunk(g)] = sh[0];
  }
}


// ---- two P-CSI iterations in ONE pass over the state (round 4; large grids) ---------------------------------------------------------------
// P-CSI has no inner product, so nothing but the stencil couples iteration k + 1 to iteration k: a workgroup that holds x, dx, r' on a
// tile and TWO rings of cells around it can form both iterations of the tile itself -- x1 = x0 + dx1 on tile + 2 rings (pointwise),
// r'1 = (b - A x1) / diag and then dx2, x2 on tile + 1 ring, r'2 on the tile -- reading the state once and writing it once per two
// iterations instead of twice (13 -> ~8 words per point and iteration with 64 x 8 tiles).  Every cell value is formed by the operations
// k_pcsi_step2 uses for it, whoever forms it: bitwise two k_pcsi_step2 launches (tests/test_gpu_parity.py::
// test_two_step_pcsi_is_bitwise_the_one_step_pcsi).  The cells of the rings are addressed by their ARRAY position (the block has two
// ghost rings: NGHOST = 2) and read at their source cell (srcmap: own index, the cyclic image, or -1 = fill: x, dx, r' are 0 there in
// every iteration).  Beyond a tripole fold (FOLD) a ghost cell's stencil neighbours are its array neighbours in mirrored order.
// Launch: 64 x 8 threads per tile of the physical domain (the tile lists of the 3-D LDS kernels, DevGrid::lds_act8); no (r, r): the
// iterations before a check go through k_pcsi_step2.
struct Pcsi2Tile {
  static constexpr int R = 8, W = 64 + 4, H = R + 4, N = W * H;      // tile + two rings
  double x1[N], dx1[N], x2[N];
};
// RAW: the pair before a convergence check also leaves the residual r2 itself in `raw` (a scratch field); k_pcsi_rr_chunks then forms the
// chunk partials of (r, r) exactly as k_pcsi_step2<true> does, so the check sees the same number
template <bool RAW, bool FOLD>
__global__ void __launch_bounds__(512)
k_pcsi_step_x2(DevGrid g, PcsiArgs a, double *__restrict__ raw) {
  if (a.sc->stop) return;   // an earlier check has converged: this launch belongs to the look-ahead interval
  using T = Pcsi2Tile;
  __shared__ T t;
  const int nxb = g.nxb, nyb = g.nyb, b = blockIdx.y;
  const int tiles_i = (nxb - 2 * NGHOST + 63) / 64, tiles_j = (nyb - 2 * NGHOST + T::R - 1) / T::R;
  int ti, tj;
  bool listed = false;
  if (!lds_tile_active<8>(g, b, tiles_i, tiles_j, ti, tj, listed)) return;
  const int i0 = NGHOST + ti * 64, j0 = NGHOST + tj * T::R;
  if (!listed && land_tile(g, b, i0, 64, j0, T::R)) return;       // x, dx, r' stay exactly 0 where no ocean cell is near
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 64 + tx;
  const double om1 = a.omega[*a.base + a.j], cq1 = a.csy * om1 - 1.0;
  const double om2 = a.omega[*a.base + a.j + 1], cq2 = a.csy * om2 - 1.0;
  const long long boff = (long long)b * g.n2;
  // array position -> (source cell or -1, does the cell exist and belong to the physical domain)
  auto src_of = [&](int i, int j) -> long long {
    if (i < 0 || i >= nxb || j < 0 || j >= nyb) return -1;
    const int m = a.srcmap[boff + (long long)j * nxb + i];
    return m;
  };
  // own position: a physical cell, or (tiles overhang the physical domain) a ghost position / a position outside the array, which is
  // treated like a cell of the rings: formed at its source cell for the benefit of its physical neighbours, never stored
  const int i = i0 + tx, j = j0 + ty;
  const int lc = (ty + 2) * T::W + tx + 2;
  const bool own = i + 1 <= blk_ie(g, b) && j + 1 <= blk_je(g, b);            // physical (i0, j0 >= ib - 1, jb - 1 already)
  const long long om_ = src_of(i, j);
  const long long qsafe = boff + (long long)NGHOST * nxb + NGHOST;            // a cell that exists and has all its neighbours in the array
  const long long q = (om_ >= 0) ? om_ : qsafe;
  // ring duty: thread tid < NR1 forms a cell of the first ring (both iterations' stencil values there), tid < NR1 + NR2 one of the second
  constexpr int NR1 = (64 + 2) * (T::R + 2) - 64 * T::R, NR2 = T::N - (64 + 2) * (T::R + 2);
  int hl = -1, hi = 0, hj = 0; bool ring1 = false;
  if (tid < NR1) {
    ring1 = true;
    int li, lj;
    const int W1 = 66;
    if (tid < W1) { lj = 0; li = tid; }
    else if (tid < 2 * W1) { lj = T::R + 1; li = tid - W1; }
    else if (tid < 2 * W1 + T::R) { lj = 1 + (tid - 2 * W1); li = 0; }
    else { lj = 1 + (tid - 2 * W1 - T::R); li = W1 - 1; }
    hi = i0 - 1 + li; hj = j0 - 1 + lj; hl = (lj + 1) * T::W + li + 1;
  } else if (tid < NR1 + NR2) {
    const int e = tid - NR1;
    int li, lj;
    if (e < T::W) { lj = 0; li = e; }
    else if (e < 2 * T::W) { lj = T::H - 1; li = e - T::W; }
    else if (e < 2 * T::W + T::R + 2) { lj = 1 + (e - 2 * T::W); li = 0; }
    else { lj = 1 + (e - 2 * T::W - (T::R + 2)); li = T::W - 1; }
    hi = i0 - 2 + li; hj = j0 - 2 + lj; hl = lj * T::W + li;
  }
  const long long hm = (hl >= 0) ? src_of(hi, hj) : -1;
  const long long hq = (hm >= 0) ? hm : qsafe;
  // ---- iteration k: dx1, x1 at the own position and at the ring cell (0 where there is no source cell: fill)
  double x1o, dx1o;
  {
    const double r0 = a.Ri[q], q0 = a.Qi[q], x0 = a.Xi[q];
    const double hr0 = a.Ri[hq], hq0 = a.Qi[hq], hx0 = a.Xi[hq];
    dx1o = om1 * r0 + cq1 * q0; x1o = x0 + dx1o;
    if (om_ < 0) { dx1o = 0.0; x1o = 0.0; }
    double hdx = om1 * hr0 + cq1 * hq0, hx1 = hx0 + hdx;
    if (hm < 0) { hdx = 0.0; hx1 = 0.0; }
    t.x1[lc] = x1o; t.dx1[lc] = dx1o;
    if (hl >= 0) { t.x1[hl] = hx1; t.dx1[hl] = hdx; }
  }
  // weights of the own position and of the first-ring cell (at their source cells)
  // (the off-centre weights from their two U-point terms -- two fields instead of three; the additions of host_setup.cpp in their order, as in
  // k_fpcg_b2 and k_pcsi_step2)
  const double cc = a.C[q], bq = a.Bv[q], a0r = a.A0R[q];
  const double x00 = g.XW[q], x0m = g.XW[q - nxb], xm0 = g.XW[q - 1], xmm = g.XW[q - 1 - nxb];
  const double y00 = g.YW[q], y0m = g.YW[q - nxb], ym0 = g.YW[q - 1], ymm = g.YW[q - 1 - nxb];
  const long long hq1 = (ring1 && hm >= 0) ? hm : qsafe;
  const double hcc = a.C[hq1], hbq = a.Bv[hq1], ha0r = a.A0R[hq1];
  const double hx00 = g.XW[hq1], hx0m = g.XW[hq1 - nxb], hxm0 = g.XW[hq1 - 1], hxmm = g.XW[hq1 - 1 - nxb];
  const double hy00 = g.YW[hq1], hy0m = g.YW[hq1 - nxb], hym0 = g.YW[hq1 - 1], hymm = g.YW[hq1 - 1 - nxb];
  const double wne = x00 + y00, wse = x0m + y0m, wnw = xm0 + ym0, wsw = xmm + ymm;
  const double we = x00 + x0m - y00 - y0m, ww = xm0 + xmm - ym0 - ymm;
  const double wn = y00 + ym0 - x00 - xm0, ws = y0m + ymm - x0m - xmm;
  const double hwne = hx00 + hy00, hwse = hx0m + hy0m, hwnw = hxm0 + hym0, hwsw = hxmm + hymm;
  const double hwe = hx00 + hx0m - hy00 - hy0m, hww = hxm0 + hxmm - hym0 - hymm;
  const double hwn = hy00 + hym0 - hx00 - hxm0, hws = hy0m + hymm - hx0m - hxmm;
  __syncthreads();
  // A ghost cell G beyond a tripole fold is the mirror image of its source cell S: the array neighbour G + (di, dj) holds the value of
  // S - (di, dj).  The operator at S -- its weights, its order of additions -- is formed at G by walking the tile the other way round.
  const int jf = FOLD ? a.jfold[b] : 0;
  const int odW = (FOLD && j >= jf) ? -T::W : T::W, od1 = (FOLD && j >= jf) ? -1 : 1;
  const int hdW = (FOLD && hj >= jf) ? -T::W : T::W, hd1 = (FOLD && hj >= jf) ? -1 : 1;
  auto stencil = [&](const double *X, int l, int dW, int d1, double c0, double n, double s_, double e, double w_, double ne, double se, double nw, double sw) {
    return c0 * X[l] + n * X[l + dW] + s_ * X[l - dW] + e * X[l + d1] + w_ * X[l - d1] +
           ne * X[l + dW + d1] + se * X[l - dW + d1] + nw * X[l + dW - d1] + sw * X[l - dW - d1];
  };
  // ---- r'1, then dx2, x2 at the own cell and at the first-ring cell
  double x2o, dx2o;
  {
    const double r1 = bq - stencil(t.x1, lc, odW, od1, cc, wn, ws, we, ww, wne, wse, wnw, wsw);
    const double rp = r1 * a0r;
    dx2o = om2 * rp + cq2 * dx1o; x2o = x1o + dx2o;
    t.x2[lc] = (om_ >= 0) ? x2o : 0.0;
    if (ring1) {
      double hx2 = 0.0;
      if (hm >= 0) {
        const double hr1 = hbq - stencil(t.x1, hl, hdW, hd1, hcc, hwn, hws, hwe, hww, hwne, hwse, hwnw, hwsw);
        const double hrp = hr1 * ha0r;
        const double hdx2 = om2 * hrp + cq2 * t.dx1[hl];
        hx2 = t.x1[hl] + hdx2;
      }
      t.x2[hl] = hx2;
    }
  }
  __syncthreads();
  // ---- r'2 at the own cell; the state after two iterations
  if (own) {
    const double r2 = bq - stencil(t.x2, lc, T::W, 1, cc, wn, ws, we, ww, wne, wse, wnw, wsw);
    a.Qo[q] = dx2o; a.Xo[q] = x2o; a.Ro[q] = r2 * a0r;
    if (RAW) raw[q] = r2;
  }
}
// chunk partials of (r, r) from the residual k_pcsi_step_x2<true> left: the cells, the factors and the tree of k_pcsi_step2<true>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_pcsi_rr_chunks(DevGrid g, PcsiArgs a, const double *__restrict__ raw) {
  if (a.sc->stop) return;
  if (red_land_out<1>(g, a.partial, a.nchunk, false)) return;
  const int p2 = red_cell(g), b = blockIdx.y;
  double v[1] = {0.0};
  if (p2 < g.n2) {
    const int i = p2 % g.nxb, j = p2 / g.nxb;
    const long long q = (long long)b * g.n2 + p2;
    if (interior(g, b, i, j)) { const double r = raw[q]; v[0] = (r * r) * (double)g.mMask8[q]; }
  }
  wg_reduce_store<1>(v, a.partial, b * a.nchunk + red_chunk(g));
}

}  // namespace pop
