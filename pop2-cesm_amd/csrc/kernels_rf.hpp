// kernels_rf.hpp -- Robert-Asselin-Williams time filter, step_RF (source/step_mod.F90:919-1350) for
// sfc_layer_varthick, fully coupled normalisation, lrf_conserveVT = .false.
// Streaming kernels; the column sum of the filter term runs in the reference's k order inside one thread.
#pragma once
#include "kernels_common.hpp"

namespace pop {

struct RfParams { double rn, rc; int nonzero_new; double dz1, grav; };

// F(new) += rn*W, F(cur) += rc*W with W = F(old) + F(new) - 2 F(cur)   (:976-1028)
__global__ void k_rf_filter(long long n, const double *__restrict__ FO, double *__restrict__ FC, double *__restrict__ FN, RfParams p) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const double w = FO[q] + FN[q] - 2.0 * FC[q];
  if (p.nonzero_new) FN[q] = FN[q] + p.rn * w;
  FC[q] = FC[q] + p.rc * w;
}

// tracer n, vertical interior k = 2..km (:1031-1049) and WORKN = TAREA * sum_k dz(k) MASK(k) S(k) (:1052-1064);
// blockIdx.z = tracer
struct RfTracerArgs { const double *TO[2]; double *TC[2], *TN[2], *WORKN[2]; };
__global__ void __launch_bounds__(POP_COL_THREADS)
k_rf_tracer_interior(DevGrid g, RfParams p, RfTracerArgs a) {
  Col c;
  if (!col_setup(g, c, false)) return;
  const int n = blockIdx.z, km = g.km, kmt = g.KMT[c.q2];
  const long long n2 = g.n2;
  const double *__restrict__ TO = a.TO[n];
  double *__restrict__ TC = a.TC[n];
  double *__restrict__ TN = a.TN[n];
  double w = 0.0;
  for (int k = 2; k <= km; ++k) {
    const long long q = c.base3 + (long long)(k - 1) * n2;
    const double S = TO[q] + TN[q] - 2.0 * TC[q];
    if (p.nonzero_new) TN[q] = TN[q] + p.rn * S;
    TC[q] = TC[q] + p.rc * S;
    const double mk = (kmt >= k) ? 1.0 : 0.0;
    w = w + g.dz[k] * mk * S;
  }
  a.WORKN[n][c.q2] = g.TAREA[c.q2] * w;
}

// surface level, thickness weighted (:1073-1093); blockIdx.y = tracer; WORKN = TAREA*MASK(1)*S
struct RfSurfArgs { const double *TO[2]; double *TC[2], *TN[2], *WORKN[2]; const double *PO, *PC, *PN; };
__global__ void k_rf_surface(DevGrid g, RfParams p, RfSurfArgs a) {
  const long long q2 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long a2 = (long long)g.n2 * g.nblocks;
  if (q2 >= a2) return;
  const int n = blockIdx.y;
  const long long b = q2 / g.n2, q = b * g.n3 + (q2 - b * g.n2);
  const double to = a.TO[n][q], tc = a.TC[n][q], tn = a.TN[n][q];
  const double S = (p.dz1 + a.PO[q2] / p.grav) * to + (p.dz1 + a.PN[q2] / p.grav) * tn - 2.0 * (p.dz1 + a.PC[q2] / p.grav) * tc;
  if (p.nonzero_new) a.TN[n][q] = (p.dz1 + a.PN[q2] / p.grav) * tn + p.rn * S;
  a.TC[n][q] = (p.dz1 + a.PC[q2] / p.grav) * tc + p.rc * S;
  const double mk = (g.KMT[q2] >= 1) ? 1.0 : 0.0;
  a.WORKN[n][q2] = g.TAREA[q2] * mk * S;
}

// PSURF filter (:1101-1112); WB = W*TAREA for the conservation sum (:1117)
__global__ void k_rf_psurf(DevGrid g, RfParams p, const double *__restrict__ PO, double *__restrict__ PC, double *__restrict__ PN, double *__restrict__ WB) {
  const long long q2 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q2 >= (long long)g.n2 * g.nblocks) return;
  const double w = PO[q2] + PN[q2] - 2.0 * PC[q2];
  if (p.nonzero_new) PN[q2] = PN[q2] + p.rn * w;
  PC[q2] = PC[q2] + p.rc * w;
  WB[q2] = w * g.TAREA[q2];
}

// conservation term of PSURF, surface tracer = (tracer*thickness)/thickness (:1121-1145), WB = TAREA*(dz1 + P(cur)/g) (:1160)
__global__ void k_rf_psurf_adjust(DevGrid g, RfParams p, double rf_sump, double *__restrict__ PC, double *__restrict__ PN,
                                  double *__restrict__ T0C, double *__restrict__ T0N, double *__restrict__ T1C, double *__restrict__ T1N,
                                  double *__restrict__ WB) {
  const long long q2 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q2 >= (long long)g.n2 * g.nblocks) return;
  const long long b = q2 / g.n2, q = b * g.n3 + (q2 - b * g.n2);
  const double w2 = (g.KMT[q2] >= 1) ? rf_sump : 0.0;
  double pn = PN[q2], pc = PC[q2];
  if (p.nonzero_new) { pn = pn - p.rn * w2; PN[q2] = pn; }
  pc = pc - p.rc * w2; PC[q2] = pc;
  if (p.nonzero_new) { T0N[q] = T0N[q] / (p.dz1 + pn / p.grav); T1N[q] = T1N[q] / (p.dz1 + pn / p.grav); }
  T0C[q] = T0C[q] / (p.dz1 + pc / p.grav); T1C[q] = T1C[q] / (p.dz1 + pc / p.grav);
  WB[q2] = g.TAREA[q2] * (p.dz1 + pc / p.grav);
}

// conservation adjustment of one tracer at every level (:1192-1206): 3-D parallel
__global__ void k_rf_conserve(DevGrid g, RfParams p, double fnew, double fcur, double *__restrict__ TC, double *__restrict__ TN) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y + 1, b = blockIdx.z;
  if (p2 >= g.n2) return;
  const long long q2 = (long long)b * g.n2 + p2, q = (long long)b * g.n3 + (long long)(k - 1) * g.n2 + p2;
  const double oo = (g.KMT[q2] >= k && g.RCALCT[q2] > 0.0) ? 1.0 : 0.0;
  if (p.nonzero_new) TN[q] = TN[q] - fnew * oo;
  TC[q] = TC[q] - fcur * oo;
}

}  // namespace pop
