// host_evp.cpp -- preprocessing of the EVP (error vector propagation) block preconditioner (host logic, no HIP).
//
// Restates POP_SolversPrep's EVP branch (POP_SolversMod.F90:252-290), EvpBlockPartition (:2992-3040), EvpPre
// (:2434-2506), ExplicitBlockEvpPre (:2508-2616), inverse (:3042-3120) and, for the Lanczos pass of P-CSI, the
// host form of preconditioner / ExplicitEvp (:2268-2369, :2618-2696).  Every block is cut into sub-blocks of at
// most 8x8 cells; on a sub-block without land the reduced operator (centre + the four corner weights; the N, S,
// E, W weights are small and dropped) with a zero rim is solved exactly by marching from guessed values on the
// west column / south row and correcting the guess with the inverse of the influence matrix; sub-blocks with land
// fall back to the diagonal.  Like the rest of the init-time work it runs on all blocks of the decomposition.
#include "pop_internal.hpp"

namespace pop {

namespace {

// 1-based start indices of the pieces of a block row / column of m = n_block-2 cells (:3016-3038)
std::vector<int> partition(int m, int mm) {
  const int mb = (m - 3) / mm + 1;
  std::vector<int> mdi(mb + 2, 0);
  mdi[1] = 2;
  if (mb == 1) { mdi[2] = m; return mdi; }
  for (int i = 1; i <= mb - 2; ++i) mdi[i + 1] = 2 + i * mm;
  mdi[mb] = (mdi[mb - 1] + m) / 2;          // the last two pieces share what is left
  mdi[mb + 1] = m;
  return mdi;
}

// inverse of an n x n matrix by LU without pivoting (:3042-3120); column-major, 0-based here
void lu_inverse(std::vector<double> &a, std::vector<double> &c, int n) {
  auto at = [n](std::vector<double> &M, int i, int j) -> double & { return M[(size_t)j * n + i]; };
  std::vector<double> L((size_t)n * n, 0.0), U((size_t)n * n, 0.0), b(n, 0.0), d(n), x(n);
  for (int k = 0; k < n - 1; ++k)
    for (int i = k + 1; i < n; ++i) {
      const double coeff = at(a, i, k) / at(a, k, k);
      at(L, i, k) = coeff;
      for (int j = k + 1; j < n; ++j) at(a, i, j) = at(a, i, j) - coeff * at(a, k, j);
    }
  for (int i = 0; i < n; ++i) at(L, i, i) = 1.0;
  for (int j = 0; j < n; ++j) for (int i = 0; i <= j; ++i) at(U, i, j) = at(a, i, j);
  for (int k = 0; k < n; ++k) {
    b[k] = 1.0;
    d[0] = b[0];
    for (int i = 1; i < n; ++i) {
      d[i] = b[i];
      for (int j = 0; j < i; ++j) d[i] = d[i] - at(L, i, j) * d[j];
    }
    x[n - 1] = d[n - 1] / at(U, n - 1, n - 1);
    for (int i = n - 2; i >= 0; --i) {
      x[i] = d[i];
      for (int j = n - 1; j > i; --j) x[i] = x[i] - at(U, i, j) * x[j];
      x[i] = x[i] / at(U, i, i);
    }
    for (int i = 0; i < n; ++i) at(c, i, k) = x[i];
    b[k] = 0.0;
  }
}

// sub-block arrays: (a, c) 1-based with rim, leading dimension EVP_LD
inline int sb(int a, int c) { return (a - 1) + EVP_LD * (c - 1); }
inline int rv(int k, int j) { return (k - 1) + EVP_LE * (j - 1); }

// one marching sweep over rows 2..jmax, columns 2..imax of the sub-block: y(i+1,j+1) from its SW neighbours
template <bool WITH_F>
void march(double *y, const double *cc, const double *ne, const double *scale, const double *f, int imax, int jmax) {
  for (int j = 2; j <= jmax; ++j)
    for (int i = 2; i <= imax; ++i) {
      if (WITH_F)
        y[sb(i + 1, j + 1)] = (f[sb(i, j)] - cc[sb(i, j)] * y[sb(i, j)] - ne[sb(i, j - 1)] * y[sb(i + 1, j - 1)] -
                               ne[sb(i - 1, j)] * y[sb(i - 1, j + 1)] - ne[sb(i - 1, j - 1)] * y[sb(i - 1, j - 1)]) * scale[sb(i, j)];
      else
        y[sb(i + 1, j + 1)] = (-cc[sb(i, j)] * y[sb(i, j)] - ne[sb(i, j - 1)] * y[sb(i + 1, j - 1)] -
                               ne[sb(i - 1, j)] * y[sb(i - 1, j + 1)] - ne[sb(i - 1, j - 1)] * y[sb(i - 1, j - 1)]) / scale[sb(i, j)];
    }
}

// influence matrix of a sub-block and its inverse (:2508-2616); returns the largest |rinv*rin - I|
double influence(const double *cc, const double *ne, double *rinv, int n, int m) {
  const int nm = n + m - 5;
  std::vector<double> y(EVP_LD * EVP_LD, 0.0), rin(EVP_LE * EVP_LE, 0.0);
  auto record = [&](int row) {
    for (int i = 1; i <= n - 2; ++i) rin[rv(row, i)] = -y[sb(i + 2, m)];        // north rim
    for (int j = 1; j <= m - 3; ++j) rin[rv(row, n - 2 + j)] = -y[sb(n, m - j)];  // east rim, downwards
  };
  for (int ii = 1; ii <= m - 2; ++ii) {       // unit values on the west column, top to bottom
    y[sb(2, m - ii)] = 1.0;
    march<false>(y.data(), cc, ne, ne, nullptr, n - 1, m - 1);
    record(ii);
    y[sb(2, m - ii)] = 0.0;
  }
  for (int ii = 1; ii <= n - 3; ++ii) {       // unit values on the south row
    y[sb(ii + 2, 2)] = 1.0;
    march<false>(y.data(), cc, ne, ne, nullptr, n - 1, m - 1);
    record(m - 2 + ii);
    y[sb(ii + 2, 2)] = 0.0;
  }
  std::vector<double> W((size_t)nm * nm), RI((size_t)nm * nm, 0.0);
  for (int j = 0; j < nm; ++j) for (int i = 0; i < nm; ++i) W[(size_t)j * nm + i] = rin[rv(i + 1, j + 1)];
  lu_inverse(W, RI, nm);
  for (int j = 0; j < nm; ++j) for (int i = 0; i < nm; ++i) rinv[rv(i + 1, j + 1)] = RI[(size_t)j * nm + i];
  double worst = 0.0;
  for (int j = 1; j <= nm; ++j) for (int i = 1; i <= nm; ++i) {
    double w = 0.0;
    for (int k = 1; k <= nm; ++k) w = w + rinv[rv(i, k)] * rin[rv(k, j)];
    if (i == j) w = w - 1.0;
    worst = std::fmax(worst, std::fabs(w));
  }
  return worst;
}

}  // namespace

int host_evp_prep(HostModel &h, const std::vector<double> &C) {
  EvpHost &E = h.evp;
  const int nxb = h.nxb, nyb = h.nyb, NB = h.nblocks_tot;
  const size_t n2 = h.n2;
  const std::vector<double> &WNE = h.f2["btropWgtNE"];
  E.xidx = partition(nxb - 2, EVP_BS); E.yidx = partition(nyb - 2, EVP_BS);
  E.xnb = (int)E.xidx.size() - 2; E.ynb = (int)E.yidx.size() - 2;
  const size_t nsb = (size_t)E.xnb * E.ynb, tot = nsb * NB;
  E.cc.assign(tot * EVP_LD * EVP_LD, 0.0); E.ne.assign(tot * EVP_LD * EVP_LD, 0.0);
  E.icc.assign(tot * EVP_LD * EVP_LD, 0.0); E.ine.assign(tot * EVP_LD * EVP_LD, 0.0);
  E.rinv.assign(tot * EVP_LE * EVP_LE, 0.0); E.land.assign(tot, 0);
  for (int b = 0; b < NB; ++b) {
    const BlockInfo &B = h.all_blocks[b];
    for (int j = 1; j <= E.ynb; ++j) for (int i = 1; i <= E.xnb; ++i) {
      // 1-based block indices of the sub-block with its rim
      const int is = E.xidx[i], ie = E.xidx[i + 1] + 1, js = E.yidx[j], je = E.yidx[j + 1] + 1;
      const int ln = ie - is + 1, lm = je - js + 1;
      const size_t s = (size_t)b * nsb + (size_t)(j - 1) * E.xnb + (i - 1);
      double *cc = &E.cc[s * EVP_LD * EVP_LD], *ne = &E.ne[s * EVP_LD * EVP_LD];
      bool dry = false;
      for (int c = 1; c <= lm; ++c) for (int a = 1; a <= ln; ++a) {
        const size_t q = b * n2 + (size_t)(js + c - 2) * nxb + (is + a - 2);
        cc[sb(a, c)] = C[q]; ne[sb(a, c)] = WNE[q];
        if (a >= 2 && a <= ln - 1 && c >= 2 && c <= lm - 1 && WNE[q] == 0.0) dry = true;
      }
      // :2483-2488: the reference compares indices of the (2:nx-1) slice with the block's ib..je
      if (is + 1 < B.ib || ie - 2 > B.ie || js + 1 < B.jb || je - 2 > B.je) dry = true;
      E.land[s] = dry;
      if (!dry && influence(cc, ne, &E.rinv[s * EVP_LE * EVP_LE], ln, lm) > 1.0e-8) {
        h.err = "POP_SolversPrep: EVP influence matrix not invertible to 1e-8 (check the EVP sub-block size)";
        return 1;
      }
    }
  }
  for (size_t p = 0; p < E.cc.size(); ++p) {
    if (E.cc[p] != 0.0) E.icc[p] = 1.0 / E.cc[p];
    if (E.ne[p] != 0.0) E.ine[p] = 1.0 / E.ne[p];
  }
  return 0;
}

// preconditioner (:2268-2369) on every block: PX = M^-1 X on the physical cells, 0 on the ghost cells
void host_evp_apply(const HostModel &h, double *PX, const double *X) {
  const EvpHost &E = h.evp;
  const int nxb = h.nxb, NB = h.nblocks_tot;
  const size_t n2 = h.n2, nsb = (size_t)E.xnb * E.ynb;
  std::fill(PX, PX + n2 * NB, 0.0);
  std::vector<double> f(EVP_LD * EVP_LD), y(EVP_LD * EVP_LD);
  for (int b = 0; b < NB; ++b)
    for (int j = 1; j <= E.ynb; ++j) for (int i = 1; i <= E.xnb; ++i) {
      const int is = E.xidx[i], ie = E.xidx[i + 1] + 1, js = E.yidx[j], je = E.yidx[j + 1] + 1;
      const int n = ie - is + 1, m = je - js + 1, nm = n + m - 5;
      const size_t s = (size_t)b * nsb + (size_t)(j - 1) * E.xnb + (i - 1);
      auto cell = [&](int a, int c) { return b * n2 + (size_t)(js + c - 2) * nxb + (is + a - 2); };
      if (E.land[s]) {
        const double *icc = &E.icc[s * EVP_LD * EVP_LD];
        for (int c = 2; c <= m - 1; ++c) for (int a = 2; a <= n - 1; ++a) PX[cell(a, c)] = X[cell(a, c)] * icc[sb(a, c)];
        continue;
      }
      const double *cc = &E.cc[s * EVP_LD * EVP_LD], *ne = &E.ne[s * EVP_LD * EVP_LD], *ine = &E.ine[s * EVP_LD * EVP_LD];
      const double *rinv = &E.rinv[s * EVP_LE * EVP_LE];
      std::fill(f.begin(), f.end(), 0.0); std::fill(y.begin(), y.end(), 0.0);
      for (int c = 2; c <= m - 1; ++c) for (int a = 2; a <= n - 1; ++a) f[sb(a, c)] = X[cell(a, c)];
      march<true>(y.data(), cc, ne, ine, f.data(), n - 1, m - 1);
      auto r = [&](int k) { return k <= n - 2 ? y[sb(k + 2, m)] : y[sb(n, m - (k - (n - 2)))]; };   // what reached the N and E rim
      for (int jj = 1; jj <= m - 2; ++jj) for (int k = 1; k <= nm; ++k) y[sb(2, m - jj)] = y[sb(2, m - jj)] + rinv[rv(k, jj)] * r(k);
      for (int ii = 1; ii <= n - 3; ++ii) for (int k = 1; k <= nm; ++k) y[sb(ii + 2, 2)] = y[sb(ii + 2, 2)] + rinv[rv(k, m - 2 + ii)] * r(k);
      march<true>(y.data(), cc, ne, ine, f.data(), n - 2, m - 2);
      for (int c = 2; c <= m - 1; ++c) for (int a = 2; a <= n - 1; ++a) PX[cell(a, c)] = y[sb(a, c)];
    }
}

}  // namespace pop
