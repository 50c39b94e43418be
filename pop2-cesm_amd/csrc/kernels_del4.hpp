// kernels_del4.hpp -- biharmonic horizontal mixing (source/hmix_del4.F90).
//
// del4 = Laplacian applied twice.  The first application (with the optional spatial scaling AMF /
// AHF and the land mask) is one 3-D-parallel kernel that writes the intermediate field on the ring
// ib-1..ie+1, jb-1..je+1 (hdiffu_del4 :730-790, hdifft_del4 :1021-1043); the second application
// has exactly the del2 form, so the tracer / momentum right-hand-side kernels consume the
// intermediate field through their ordinary 5-point path with the del4 coefficient set
// (init_del4u :262-370, init_del4t :563-577, built on the host).
#pragma once

namespace pop {

// Each thread handles POP_DEL4_KC consecutive levels of its column: the 2-D operator weights (9 arrays for
// tracers, 13 for momentum) are loaded once per chunk instead of once per level -- at tx0.1v3 they were 3/4 of
// the kernel's traffic (55 of 72 GB) in the one-level-per-thread form.
#define POP_DEL4_KC 8
__global__ void k_del4_d2t(DevGrid g, const double *__restrict__ AHF, const double *__restrict__ T0, const double *__restrict__ T1,
                           double *__restrict__ D0, double *__restrict__ D1, int tile) {
  int p2;
  const int k0 = blockIdx.y * POP_DEL4_KC + 1, b = blockIdx.z;
  if (!patch_cell(g, tile, b, p2)) return;
  const int i = p2 % g.nxb, j = p2 / g.nxb, nxb = g.nxb;
  if (i + 1 < g.ib - 1 || i + 1 > g.ie + 1 || j + 1 < g.jb - 1 || j + 1 > g.je + 1) return;
  const long long q = (long long)b * g.n2 + p2;
  const int kmt = g.KMT[q], kmtn = g.KMTN[q], kmts = g.KMTS[q], kmte = g.KMTE[q], kmtw = g.KMTW[q];
  const double dtn = g.DTN[q], dts = g.DTS[q], dte = g.DTE[q], dtw = g.DTW[q];
  const double ahf = AHF[q];
  const int k1 = min(k0 + POP_DEL4_KC - 1, g.km);
  for (int k = k0; k <= k1; ++k) {
    const long long o = (long long)b * g.n3 + (long long)(k - 1) * g.n2 + p2;
    const double CN = (k <= kmtn && k <= kmt) ? dtn : 0.0, CS = (k <= kmts && k <= kmt) ? dts : 0.0;
    const double CE = (k <= kmte && k <= kmt) ? dte : 0.0, CW = (k <= kmtw && k <= kmt) ? dtw : 0.0;
    const double CC = -(CN + CS + CE + CW);
    D0[o] = ahf * (CC * T0[o] + CN * T0[o + nxb] + CS * T0[o - nxb] + CE * T0[o + 1] + CW * T0[o - 1]);
    D1[o] = ahf * (CC * T1[o] + CN * T1[o + nxb] + CS * T1[o - nxb] + CE * T1[o + 1] + CW * T1[o - 1]);
  }
}

__global__ void k_del4_d2u(DevGrid g, const double *__restrict__ AMF, const double *__restrict__ U, const double *__restrict__ V,
                           double *__restrict__ DU, double *__restrict__ DV, int tile) {
  int p2;
  const int k0 = blockIdx.y * POP_DEL4_KC + 1, b = blockIdx.z;
  if (!patch_cell(g, tile, b, p2)) return;
  const int i = p2 % g.nxb, j = p2 / g.nxb, nxb = g.nxb;
  if (i + 1 < g.ib - 1 || i + 1 > g.ie + 1 || j + 1 < g.jb - 1 || j + 1 > g.je + 1) return;
  const long long q = (long long)b * g.n2 + p2;
  const int kmu = g.KMU[q];
  const double cc = g.DUC[q] + g.DUM[q];
  const double dun = g.DUN[q], dus = g.DUS[q], due = g.DUE[q], duw = g.DUW[q];
  const double dmc = g.DMC[q], dmn = g.DMN[q], dms = g.DMS[q], dme = g.DME[q], dmw = g.DMW[q];
  const double amf = AMF[q];
  const int k1 = min(k0 + POP_DEL4_KC - 1, g.km);
  for (int k = k0; k <= k1; ++k) {
    const long long o = (long long)b * g.n3 + (long long)(k - 1) * g.n2 + p2;
    double du = 0.0, dv = 0.0;
    if (k <= kmu) {
      const double u0 = U[o], un = U[o + nxb], us = U[o - nxb], ue = U[o + 1], uw = U[o - 1];
      const double v0 = V[o], vn = V[o + nxb], vs = V[o - nxb], ve = V[o + 1], vw = V[o - 1];
      du = (cc * u0 + dun * un + dus * us + due * ue + duw * uw) + (dmc * v0 + dmn * vn + dms * vs + dme * ve + dmw * vw);
      dv = (cc * v0 + dun * vn + dus * vs + due * ve + duw * vw) - (dmc * u0 + dmn * un + dms * us + dme * ue + dmw * uw);
      du = amf * du; dv = amf * dv;
    }
    DU[o] = du; DV[o] = dv;
  }
}

inline int del4_create(HostModel &, const DevGrid &, MixDev &, std::vector<void *> &, std::string &) { return 0; }

inline int mix_hdifft_del4(const HostModel &h, const DevGrid &g, const StepParams &, const MixDev &m, const double *T0, const double *T1,
                           double *D0, double *D1, double *, double *, hipStream_t st, std::string &err) {
  const int tile = patch_rows(g, h.tun.del4_tile);
  hipLaunchKernelGGL(k_del4_d2t, dim3(patch_grid_x(g, tile), (g.km + POP_DEL4_KC - 1) / POP_DEL4_KC, g.nblocks), dim3(tile ? 64 * tile : 256), 0, st, g, m.D4AHF, T0, T1, D0, D1, tile);
  if (hipGetLastError() != hipSuccess) { err = "del4 tracer kernel launch failed"; return 1; }
  return 0;
}
inline int mix_hdiffu_del4(const HostModel &h, const DevGrid &g, const StepParams &, const MixDev &m, const double *U, const double *V,
                           double *DU, double *DV, double *, double *, hipStream_t st, std::string &err) {
  const int tile = patch_rows(g, h.tun.del4_tile);
  hipLaunchKernelGGL(k_del4_d2u, dim3(patch_grid_x(g, tile), (g.km + POP_DEL4_KC - 1) / POP_DEL4_KC, g.nblocks), dim3(tile ? 64 * tile : 256), 0, st, g, m.D4AMF, U, V, DU, DV, tile);
  if (hipGetLastError() != hipSuccess) { err = "del4 momentum kernel launch failed"; return 1; }
  return 0;
}

}  // namespace pop
