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
  double csy;
  int j;                           // step inside the interval (1-based); j = 0: start-up step
  int remote_ghosts;               // multi-rank: also advance dx, x at ghosts owned by other ranks
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
  if (!interior(g, i, j)) return;
  const long long q = (long long)b * g.n2 + p2;
  R[q] = R[q] * A0R[q];
}

// fused step.  The residual arrays hold the PRECONDITIONED residual r' = r * (1/diag): the product the
// reference forms at the top of the next iteration (:1705-1712) is formed here by the thread that just
// computed r, so a neighbour's update needs three loads (r', dx, x) and no division.
template <bool FIRST, bool WITH_RR>
__global__ void __launch_bounds__(POP_RED_THREADS)
k_pcsi_step(DevGrid g, PcsiArgs a) {
  const int p2 = red_cell(g), b = blockIdx.y, nxb = g.nxb;
  double v[1] = {0.0};
  if (p2 < g.n2) {
    const int i = p2 % nxb, j = p2 / nxb;
    const long long q = (long long)b * g.n2 + p2;
    if (interior(g, i, j)) {
      const double om = FIRST ? a.omega[0] : a.omega[*a.base + a.j];
      const double cq = FIRST ? 0.0 : a.csy * om - 1.0;
      const bool rim = (i + 1 == g.ib || i + 1 == g.ie || j + 1 == g.jb || j + 1 == g.je);
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
      a.Qo[q] = dx0; a.Xo[q] = xn[0]; a.Ro[q] = r * a0r;
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
  if (WITH_RR) wg_reduce_store<1>(v, a.partial, b * gridDim.x + red_chunk(g));
}

}  // namespace pop
