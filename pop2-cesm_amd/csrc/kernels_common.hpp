// kernels_common.hpp -- shared device helpers: MWJF equation of state, column indexing.
#pragma once
#include "device_types.hpp"

namespace pop {

#define POP_COL_THREADS 64   // one wavefront per workgroup for column-march kernels
#define POP_STENCIL_MAX_THREADS 512   // stencil column kernels: up to 64 x 8 rows per workgroup

// Column-kernel prologue: one thread per (i,j) of local block b; returns false for threads
// outside the physical domain ib..ie, jb..je (or outside the block).
struct Col {
  int i, j, b, p2;        // 0-based i,j; p2 = j*nxb+i
  long long q2;           // b*n2 + p2
  long long base3;        // b*n3 + p2   (+ (k-1)*n2 for level k)
};
// XCD-aware tile order: the hardware deals workgroups round-robin over the 8 XCDs (blocks b and b+8
// share one L2, MI355X_MICROARCH.md), so consecutive blockIdx.x are mapped to tiles that are nT/8
// apart: every XCD then owns one contiguous band of the (i,j) plane and the j+-1 rows re-read by
// neighbouring workgroups hit that XCD's own L2 instead of being fetched once per XCD.
// gridDim.x must be a multiple of 8 (col_grid_x); surplus tiles idle.
__host__ __device__ inline int col_grid_x(int n2, int threads) { const int nt = (n2 + threads - 1) / threads; return 8 * ((nt + 7) / 8); }
__host__ inline int col_grid_x2d(int nxb, int nyb, int tx, int ty) { const int nt = ((nxb + tx - 1) / tx) * ((nyb + ty - 1) / ty); return 8 * ((nt + 7) / 8); }
__device__ __forceinline__ bool col_setup(const DevGrid &g, Col &c, bool interior_only) {
  const int tile = g.xcd_remap ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  c.b = blockIdx.y;
  if (blockDim.y > 1) {
    // 2-D workgroup (stencil kernels): blockDim.y consecutive rows of one 64-wide i strip, one wave per
    // row, so the j+-1 rows a wave reads are the rows its sibling waves on the same CU are reading
    const int tiles_i = (g.nxb + blockDim.x - 1) / blockDim.x;
    c.i = (tile % tiles_i) * blockDim.x + threadIdx.x;
    c.j = (tile / tiles_i) * blockDim.y + threadIdx.y;
    if (c.i >= g.nxb || c.j >= g.nyb) return false;
    c.p2 = c.j * g.nxb + c.i;
  } else {
    c.p2 = tile * blockDim.x + threadIdx.x;
    if (c.p2 >= g.n2) return false;
    c.i = c.p2 % g.nxb;
    c.j = c.p2 / g.nxb;
  }
  if (interior_only && (c.i + 1 < g.ib || c.i + 1 > g.ie || c.j + 1 < g.jb || c.j + 1 > g.je)) return false;
  c.q2 = (long long)c.b * g.n2 + c.p2;
  c.base3 = (long long)c.b * g.n3 + c.p2;
  return true;
}

// ---- McDougall, Wright, Jackett & Feistel (2003) equation of state as used by the
// reference (state_mod.F90:418-498, state_range_opt='enforce' :383-389, ranges :1054-1057).
// Coefficients are the published MWJF values scaled to g/cm^3 (numerator * 0.001).
struct MwjfP {   // pressure-dependent coefficients, wave-uniform per level
  double n0, n2, ns1t0, d0, d1, d3;
};
__device__ __forceinline__ MwjfP mwjf_level(double pbar) {
  const double p = 10.0 * pbar;
  MwjfP c;
  c.n0 = (9.99843699e+2 * 0.001) + p * ((1.04004591e-2 * 0.001) + p * (-3.24041825e-8 * 0.001));
  c.n2 = (-5.45928211e-2 * 0.001) + p * ((1.03970529e-7 * 0.001) + p * (-1.23869360e-11 * 0.001));
  c.ns1t0 = (2.96938239e+0 * 0.001) + p * (5.18761880e-6 * 0.001);
  c.d0 = 1.0e+0 + p * 5.30848875e-6;
  c.d1 = 7.28606739e-3 + (p * p * p) * -1.27934137e-17;
  c.d3 = 3.68390573e-7 + (p * p) * -3.03175128e-16;
  return c;
}
template <bool DERIV>
__device__ __forceinline__ double mwjf_rho(const MwjfP &c, double TK, double SK, double *drdt, double *drds) {
  const double n1 = 7.35212840e+0 * 0.001, n3 = 3.98476704e-4 * 0.001;
  const double ns1t1 = -7.23268813e-3 * 0.001, ns2t0 = 2.12382341e-3 * 0.001;
  const double d2 = -4.60835542e-5, d4 = 1.80809186e-10, ds1t0 = 2.14691708e-3, ds1t1 = -9.27062484e-6;
  const double ds1t3 = -1.78343643e-10, dsqt0 = 4.76534122e-6, dsqt2 = 1.63410736e-9;
  double TQ = fmin(TK, 999.0); TQ = fmax(TQ, -2.0);
  double SQ = fmin(SK, 0.999); SQ = fmax(SQ, 0.0);
  SQ = 1000.0 * SQ;
  const double SQR = sqrt(SQ);
  const double W1 = c.n0 + TQ * (n1 + TQ * (c.n2 + n3 * TQ)) + SQ * (c.ns1t0 + ns1t1 * TQ + ns2t0 * SQ);
  const double W2 = c.d0 + TQ * (c.d1 + TQ * (d2 + TQ * (c.d3 + d4 * TQ))) +
                    SQ * (ds1t0 + TQ * (ds1t1 + TQ * TQ * ds1t3) + SQR * (dsqt0 + TQ * TQ * dsqt2));
  const double DEN = 1.0 / W2;
  if (DERIV) {
    double W3 = n1 + TQ * (2.0 * c.n2 + 3.0 * n3 * TQ) + ns1t1 * SQ;
    double W4 = c.d1 + SQ * ds1t1 + TQ * (2.0 * (d2 + SQ * SQR * dsqt2) + TQ * (3.0 * (c.d3 + SQ * ds1t3) + TQ * 4.0 * d4));
    *drdt = (W3 - W1 * DEN * W4) * DEN;
    W3 = c.ns1t0 + ns1t1 * TQ + 2.0 * ns2t0 * SQ;
    W4 = ds1t0 + TQ * (ds1t1 + TQ * TQ * ds1t3) + 1.5 * SQR * (dsqt0 + TQ * TQ * dsqt2);
    *drds = (W3 - W1 * DEN * W4) * DEN * 1000.0;
  }
  return W1 * DEN;
}

}  // namespace pop
