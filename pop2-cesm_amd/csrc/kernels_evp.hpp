// kernels_evp.hpp -- EVP block preconditioner on the device (POP_SolversMod.F90:2268-2369 preconditioner,
// :2618-2696 ExplicitEvp; preprocessing in host_evp.cpp).
//
// One thread owns one sub-block (at most 8x8 cells + rim): the solve is two sequential marching sweeps with a
// small dense correction in between, in exactly the reference's operation order, so the only parallelism is
// across sub-blocks (n2/64 of them per block).  The marching array lives in LDS as y[cell][thread] (bank-conflict
// free, 50 KB per 64-thread workgroup); the per-sub-block coefficients are stored [cell][sub-block] so that a
// wave reads them coalesced.  Ghost cells of PX are never written here: the buffer starts zeroed and the halo
// update that follows every application (:1352, :1719, :2023, :2130) fills them, as in the reference where PX is
// zeroed first.
#pragma once

namespace pop {

#define POP_EVP_THREADS 64

struct EvpDev {
  long long S;                    // local sub-blocks (stride of the coefficient arrays)
  const int4 *meta;               // x: cell index of (1,1) of the sub-block with rim; y: n | m << 8; z: land
  const double *cc, *ne, *icc, *ine;   // [EVP_LD*EVP_LD][S]
  const double *rinv;             // [EVP_LE*EVP_LE][S]
};

__global__ void __launch_bounds__(POP_EVP_THREADS)
k_evp_apply(EvpDev e, int nxb, const double *__restrict__ X, double *__restrict__ PX) {
  __shared__ double ys[EVP_LD * EVP_LD * POP_EVP_THREADS];
  const int t = threadIdx.x;
  const long long s = (long long)blockIdx.x * POP_EVP_THREADS + t;
  if (s >= e.S) return;
  const int4 mt = e.meta[s];
  const int n = mt.y & 255, m = mt.y >> 8;
  auto cell = [&](int a, int c) { return (long long)mt.x + (long long)(c - 1) * nxb + (a - 1); };
  auto co = [&](const double *A, int a, int c) { return A[(long long)((a - 1) + EVP_LD * (c - 1)) * e.S + s]; };
  if (mt.z) {   // sub-block with land: diagonal scaling (:2344-2348)
    for (int c = 2; c <= m - 1; ++c) for (int a = 2; a <= n - 1; ++a) PX[cell(a, c)] = X[cell(a, c)] * co(e.icc, a, c);
    return;
  }
  auto y = [&](int a, int c) -> double & { return ys[((a - 1) + EVP_LD * (c - 1)) * POP_EVP_THREADS + t]; };
  for (int c = 1; c <= m; ++c) for (int a = 1; a <= n; ++a) y(a, c) = 0.0;
  auto sweep = [&](int imax, int jmax) {
    for (int j = 2; j <= jmax; ++j)
      for (int i = 2; i <= imax; ++i)
        y(i + 1, j + 1) = (X[cell(i, j)] - co(e.cc, i, j) * y(i, j) - co(e.ne, i, j - 1) * y(i + 1, j - 1) -
                           co(e.ne, i - 1, j) * y(i - 1, j + 1) - co(e.ne, i - 1, j - 1) * y(i - 1, j - 1)) * co(e.ine, i, j);
  };
  sweep(n - 1, m - 1);
  // what reached the north / east rim, then the corrected values on the west column and the south row (:2667-2680)
  const int nm = n + m - 5;
  auto r = [&](int k) { return k <= n - 2 ? y(k + 2, m) : y(n, m - (k - (n - 2))); };
  auto rinv = [&](int k, int j) { return e.rinv[(long long)((k - 1) + EVP_LE * (j - 1)) * e.S + s]; };
  for (int jj = 1; jj <= m - 2; ++jj) {
    double acc = y(2, m - jj);
    for (int k = 1; k <= nm; ++k) acc = acc + rinv(k, jj) * r(k);
    y(2, m - jj) = acc;
  }
  for (int ii = 1; ii <= n - 3; ++ii) {
    double acc = y(ii + 2, 2);
    for (int k = 1; k <= nm; ++k) acc = acc + rinv(k, m - 2 + ii) * r(k);
    y(ii + 2, 2) = acc;
  }
  sweep(n - 2, m - 2);
  for (int c = 2; c <= m - 1; ++c) for (int a = 2; a <= n - 1; ++a) PX[cell(a, c)] = y(a, c);
}

}  // namespace pop
