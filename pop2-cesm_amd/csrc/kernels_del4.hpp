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
// PBC: partial bottom cells (hmix_del4.F90:964-984, 683-697): neighbour weights scaled by min(thickness) / thickness
template <bool PBC = false>
__global__ void k_del4_d2t(DevGrid g, const double *__restrict__ AHF, const double *__restrict__ T0, const double *__restrict__ T1,
                           double *__restrict__ D0, double *__restrict__ D1, int tile) {
  int p2;
  const int k0 = blockIdx.y * POP_DEL4_KC + 1, b = blockIdx.z;
  if (!patch_cell(g, tile, b, p2)) return;
  const int i = p2 % g.nxb, j = p2 / g.nxb, nxb = g.nxb;
  if (i + 1 < g.ib - 1 || i + 1 > blk_ie(g, b) + 1 || j + 1 < g.jb - 1 || j + 1 > blk_je(g, b) + 1) return;
  const long long q = (long long)b * g.n2 + p2;
  const int kmt = g.KMT[q], kmtn = g.KMTN[q], kmts = g.KMTS[q], kmte = g.KMTE[q], kmtw = g.KMTW[q];
  const double dtn = g.DTN[q], dts = g.DTS[q], dte = g.DTE[q], dtw = g.DTW[q];
  const double ahf = AHF[q];
  double dzbc = 0, dzbcn = 0, dzbcs = 0, dzbce = 0, dzbcw = 0;
  if (PBC) { dzbc = g.DZBC[q]; dzbcn = g.DZBC[q + nxb]; dzbcs = g.DZBC[q - nxb]; dzbce = g.DZBC[q + 1]; dzbcw = g.DZBC[q - 1]; }
  const int k1 = min(k0 + POP_DEL4_KC - 1, g.km);
  for (int k = k0; k <= k1; ++k) {
    const long long o = (long long)b * g.n3 + (long long)(k - 1) * g.n2 + p2;
    double CN = dtn, CS = dts, CE = dte, CW = dtw;
    if (PBC) {
      const double dzt = pbc_dz(g, k, kmt, dzbc);
      CN = dtn * fmin(dzt, pbc_dz(g, k, kmtn, dzbcn)) / dzt; CS = dts * fmin(dzt, pbc_dz(g, k, kmts, dzbcs)) / dzt;
      CE = dte * fmin(dzt, pbc_dz(g, k, kmte, dzbce)) / dzt; CW = dtw * fmin(dzt, pbc_dz(g, k, kmtw, dzbcw)) / dzt;
    }
    if (!(k <= kmtn && k <= kmt)) CN = 0.0;
    if (!(k <= kmts && k <= kmt)) CS = 0.0;
    if (!(k <= kmte && k <= kmt)) CE = 0.0;
    if (!(k <= kmtw && k <= kmt)) CW = 0.0;
    const double CC = -(CN + CS + CE + CW);
    D0[o] = ahf * (CC * T0[o] + CN * T0[o + nxb] + CS * T0[o - nxb] + CE * T0[o + 1] + CW * T0[o - 1]);
    D1[o] = ahf * (CC * T1[o] + CN * T1[o + nxb] + CS * T1[o - nxb] + CE * T1[o + 1] + CW * T1[o - 1]);
  }
}

template <bool PBC = false>
__global__ void k_del4_d2u(DevGrid g, const double *__restrict__ AMF, const double *__restrict__ U, const double *__restrict__ V,
                           double *__restrict__ DU, double *__restrict__ DV, int tile) {
  int p2;
  const int k0 = blockIdx.y * POP_DEL4_KC + 1, b = blockIdx.z;
  if (!patch_cell(g, tile, b, p2)) return;
  const int i = p2 % g.nxb, j = p2 / g.nxb, nxb = g.nxb;
  if (i + 1 < g.ib - 1 || i + 1 > blk_ie(g, b) + 1 || j + 1 < g.jb - 1 || j + 1 > blk_je(g, b) + 1) return;
  const long long q = (long long)b * g.n2 + p2;
  const int kmu = g.KMU[q];
  const double cc = g.DUC[q] + g.DUM[q];
  const double dun = g.DUN[q], dus = g.DUS[q], due = g.DUE[q], duw = g.DUW[q];
  const double dmc = g.DMC[q], dmn = g.DMN[q], dms = g.DMS[q], dme = g.DME[q], dmw = g.DMW[q];
  const double amf = AMF[q];
  int kmun = 0, kmus = 0, kmue = 0, kmuw = 0; double dzub = 0, dzubn = 0, dzubs = 0, dzube = 0, dzubw = 0;
  if (PBC) {
    kmun = g.KMU[q + nxb]; kmus = g.KMU[q - nxb]; kmue = g.KMU[q + 1]; kmuw = g.KMU[q - 1];
    dzub = g.DZUB[q]; dzubn = g.DZUB[q + nxb]; dzubs = g.DZUB[q - nxb]; dzube = g.DZUB[q + 1]; dzubw = g.DZUB[q - 1];
  }
  const int k1 = min(k0 + POP_DEL4_KC - 1, g.km);
  for (int k = k0; k <= k1; ++k) {
    const long long o = (long long)b * g.n3 + (long long)(k - 1) * g.n2 + p2;
    double du = 0.0, dv = 0.0;
    if (k <= kmu) {
      const double u0 = U[o], un = U[o + nxb], us = U[o - nxb], ue = U[o + 1], uw = U[o - 1];
      const double v0 = V[o], vn = V[o + nxb], vs = V[o - nxb], ve = V[o + 1], vw = V[o - 1];
      double cn = dun, cs = dus, ce = due, cw = duw;
      if (PBC) {
        const double dzu = pbc_dz(g, k, kmu, dzub);
        cn = dun * fmin(pbc_dz(g, k, kmun, dzubn), dzu) / dzu; cs = dus * fmin(pbc_dz(g, k, kmus, dzubs), dzu) / dzu;
        ce = due * fmin(pbc_dz(g, k, kmue, dzube), dzu) / dzu; cw = duw * fmin(pbc_dz(g, k, kmuw, dzubw), dzu) / dzu;
      }
      du = (cc * u0 + cn * un + cs * us + ce * ue + cw * uw) + (dmc * v0 + dmn * vn + dms * vs + dme * ve + dmw * vw);
      dv = (cc * v0 + cn * vn + cs * vs + ce * ve + cw * vw) - (dmc * u0 + dmn * un + dms * us + dme * ue + dmw * uw);
      du = amf * du; dv = amf * dv;
    }
    DU[o] = du; DV[o] = dv;
  }
}

inline int del4_create(HostModel &, const DevGrid &, MixDev &, std::vector<void *> &, std::string &) { return 0; }

inline int mix_hdifft_del4(const HostModel &h, const DevGrid &g, const StepParams &, const MixDev &m, const double *T0, const double *T1,
                           double *D0, double *D1, double *, double *, hipStream_t st, std::string &err) {
  const int tile = patch_rows(g, h.tun.del4_tile);
  const dim3 G(patch_grid_x(g, tile), (g.km + POP_DEL4_KC - 1) / POP_DEL4_KC, g.nblocks), B(tile ? 64 * tile : 256);
  if (g.pbc) hipLaunchKernelGGL(k_del4_d2t<true>, G, B, 0, st, g, m.D4AHF, T0, T1, D0, D1, tile);
  else hipLaunchKernelGGL(k_del4_d2t<false>, G, B, 0, st, g, m.D4AHF, T0, T1, D0, D1, tile);
  if (hipGetLastError() != hipSuccess) { err = "del4 tracer kernel launch failed"; return 1; }
  return 0;
}
inline int mix_hdiffu_del4(const HostModel &h, const DevGrid &g, const StepParams &, const MixDev &m, const double *U, const double *V,
                           double *DU, double *DV, double *, double *, hipStream_t st, std::string &err) {
  const int tile = patch_rows(g, h.tun.del4_tile);
  const dim3 G(patch_grid_x(g, tile), (g.km + POP_DEL4_KC - 1) / POP_DEL4_KC, g.nblocks), B(tile ? 64 * tile : 256);
  if (g.pbc) hipLaunchKernelGGL(k_del4_d2u<true>, G, B, 0, st, g, m.D4AMF, U, V, DU, DV, tile);
  else hipLaunchKernelGGL(k_del4_d2u<false>, G, B, 0, st, g, m.D4AMF, U, V, DU, DV, tile);
  if (hipGetLastError() != hipSuccess) { err = "del4 momentum kernel launch failed"; return 1; }
  return 0;
}

}  // namespace pop
