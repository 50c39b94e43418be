// host_pcsi.cpp -- solver preprocessing for P-CSI (host logic, no HIP): the Lanczos estimate of the
// extreme eigenvalues of the diagonally preconditioned barotropic operator.
//
// Restates POP_SolversPrep for solverChoice = 'PCSI' (POP_SolversMod.F90:181-320, called once from
// initial.F90:353), PcsiLanczos (:2699-2990) and ratqr (:3122-3222, EISPACK RATQR).  The centre weight
// is the one init_barotropic leaves behind (barotropic.F90:231-252): centerWgtIndep -
// TAREA/(alpha*2*dtp*dtp*grav) on ocean points.  Init-time work on all blocks of the decomposition, like
// the grid set-up, so every rank obtains the same two numbers without communication.
#include <algorithm>
#include "pop_internal.hpp"

namespace pop {

namespace {

// smallest eigenvalue of a positive definite symmetric tridiagonal matrix (1-based d, e; e[1] arbitrary)
int ratqr(int n, double eps1, const std::vector<double> &d, const std::vector<double> &e, double &mineig) {
  std::vector<double> bd(n + 2), w(n + 2);
  double f, ep, delta = 0.0, err = 0.0, p, q = 0.0, qp = 0.0, r, s = 0.0, tot;
  for (int i = 1; i <= n; ++i) w[i] = d[i];
  tot = w[1];
  for (int i = 1; i <= n; ++i) {
    p = q;
    bd[i] = e[i] * e[i];
    q = 0.0;
    if (i != n) q = std::fabs(e[i + 1]);
    tot = std::fmin(w[i] - p - q, tot);
  }
  bd[1] = 0.0;
  if (tot < 0.0) tot = 0.0;
  else for (int i = 1; i <= n; ++i) w[i] = w[i] - tot;
  for (;;) {
    tot = tot + s;
    delta = w[n] - s;
    if (delta <= eps1) break;
    f = bd[n] / delta;
    qp = delta + f;
    p = 1.0;
    for (int ii = 1; ii <= n - 1; ++ii) {
      const int i = n - ii;
      q = w[i] - s - f;
      r = q / qp;
      p = p * r + 1.0;
      ep = f * r;
      w[i + 1] = qp + ep;
      delta = q - ep;
      if (delta <= eps1) break;
      f = bd[i] / q;
      qp = delta + f;
      bd[i + 1] = qp * ep;
    }
    if (delta <= eps1) break;
    w[1] = qp;
    s = qp / p;
    if (tot + s <= tot) return 1;          // irregular end of iteration
  }
  w[1] = tot;
  err = err + std::fabs(delta);
  (void)err;
  mineig = w[1];
  return 0;
}

}  // namespace

// barotropic.F90:231-252: centerWgtIndep - TAREA/(alpha*2*dtp*dtp*grav) on ocean points, all blocks
std::vector<double> host_center_init(HostModel &h) {
  const size_t A2 = h.n2 * h.nblocks_tot;
  const std::vector<double> &WC0 = h.f2["centerWgtIndep"], &TAREA = h.f2["TAREA"];
  const std::vector<int> &KMT = h.i2["KMT"];
  const double alpha = 1.0 / 3.0;          // time_management.F90:437
  std::vector<double> C(A2);
  for (size_t p = 0; p < A2; ++p) {
    const double dc = (KMT[p] >= 1) ? TAREA[p] / (alpha * 2.0 * h.dtp * h.dtp * GRAV) : 0.0;
    C[p] = WC0[p] - dc;
  }
  return C;
}

int host_pcsi_prep(HostModel &h, const std::vector<double> &C) {
  const pop_config &c = h.c;
  const size_t n2 = h.n2, A2 = n2 * h.nblocks_tot;
  const int nxb = h.nxb, nyb = h.nyb, NB = h.nblocks_tot;
  const std::vector<double> &WNE = h.f2["btropWgtNE"], &WEa = h.f2["btropWgtEast"], &WNo = h.f2["btropWgtNorth"];
  const std::vector<double> &mMask = h.f2["mMask"];
  const bool evp = use_evp(c);             // :2789, :2843, :2884 preconditioner() instead of the diagonal
  std::vector<double> A0R(A2), R(A2, 1.0), S(A2), Q(A2, 0.0), Q1(A2, 0.0), P(A2), WORK(A2), WORK1(A2);
  for (size_t p = 0; p < A2; ++p) A0R[p] = (C[p] != 0.0) ? 1.0 / C[p] : 0.0;
  auto precond = [&](std::vector<double> &out, const std::vector<double> &in) {
    if (evp) host_evp_apply(h, out.data(), in.data());
    else for (size_t p = 0; p < A2; ++p) out[p] = in[p] * A0R[p];
  };
  auto op = [&](std::vector<double> &AX, const std::vector<double> &X) {   // btropOperator :2414-2426 (all but the outer ring)
    std::fill(AX.begin(), AX.end(), 0.0);
    for (int b = 0; b < NB; ++b) for (int j = 1; j <= nyb - 2; ++j) for (int i = 1; i <= nxb - 2; ++i) {
      const size_t q = b * n2 + (size_t)j * nxb + i;
      AX[q] = C[q] * X[q] + WNo[q] * X[q + nxb] + WNo[q - nxb] * X[q - nxb] + WEa[q] * X[q + 1] + WEa[q - 1] * X[q - 1] +
              WNE[q] * X[q + nxb + 1] + WNE[q - nxb] * X[q - nxb + 1] + WNE[q - 1] * X[q + nxb - 1] + WNE[q - 1 - nxb] * X[q - nxb - 1];
    }
  };
  const int maxstep = c.maxlanczosstep > 0 ? c.maxlanczosstep : 20;       // maxlanczosstep :626
  const double crit = c.lanczos_convergence_criterion > 0.0 ? c.lanczos_convergence_criterion : 0.1;    // LanczosconvergenceCriterion :616
  precond(S, R);
  for (size_t p = 0; p < A2; ++p) WORK[p] = S[p] * R[p];
  double csc = -host_global_sum(h, WORK.data(), mMask.data()), csa, csb = 0.0, u = 0.0, v = 0.0, mineig = 1.0;
  if (!(csc > 0.0)) { h.err = "PcsiLanczos: start vector has zero norm (singular operator)"; return 1; }
  for (size_t p = 0; p < A2; ++p) Q[p] = (1 / std::sqrt(csc)) * R[p];
  host_halo_r8_loc(h, Q.data(), 1, 0.0, 0, 0);
  std::vector<double> vcsa(maxstep + 2), vcsb(maxstep + 2), mcsa(maxstep + 2), mcsb(maxstep + 2);
  h.pcsi_lanczos_steps = 0;
  for (int m = 1; m <= maxstep; ++m) {
    h.pcsi_lanczos_steps = m;
    precond(P, Q);
    host_halo_r8_loc(h, P.data(), 1, 0.0, 0, 0);
    op(WORK1, P);
    for (size_t p = 0; p < A2; ++p) { R[p] = WORK1[p] - csb * Q1[p]; WORK[p] = P[p] * R[p]; }
    csa = -host_global_sum(h, WORK.data(), mMask.data());
    for (size_t p = 0; p < A2; ++p) R[p] = R[p] - csa * Q[p];
    host_halo_r8_loc(h, R.data(), 1, 0.0, 0, 0);
    precond(S, R);
    for (size_t p = 0; p < A2; ++p) WORK[p] = S[p] * R[p];
    csc = -host_global_sum(h, WORK.data(), mMask.data());
    csb = std::sqrt(csc);
    vcsa[m] = csa; vcsb[m] = csb;
    if (m == 1) u = vcsa[1] + vcsb[1];                                   // Gershgorin bound on the largest eigenvalue
    else u = std::fmax(u, vcsa[m] + vcsb[m] + vcsb[m - 1]);
    if (csb == 0.0) { h.err = "PcsiLanczos: breakdown (beta == 0)"; return 1; }
    for (size_t p = 0; p < A2; ++p) { Q1[p] = Q[p]; Q[p] = (1 / csb) * R[p]; }
    if (m % 10 == 0 || m == maxstep) {
      for (int i = 1; i <= m - 1; ++i) { mcsa[i] = vcsa[i]; mcsb[i + 1] = vcsb[i]; }
      mcsa[m] = vcsa[m]; mcsb[1] = 0.0;
      if (ratqr(m, 1.0e-8, mcsa, mcsb, v)) { h.err = "PcsiLanczos: error estimating the smallest eigenvalue"; return 1; }
      if (std::fabs(1 - v / mineig) < crit) break;
      mineig = v;
    }
  }
  h.pcsi_max_eig = u; h.pcsi_min_eig = v;
  return 0;
}

}  // namespace pop
