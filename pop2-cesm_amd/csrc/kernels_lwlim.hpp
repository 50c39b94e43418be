// kernels_lwlim.hpp -- tracer advection with one-dimensional flux limiters, tadvect = 3 ('lw_lim': second-order
// forward-in-time Lax-Wendroff fluxes limited direction by direction; source/advection.F90:2684-3280, with
// comp_flux_vel_ghost :1014-1120 and the luse_lw_lim branches of comp_flux_vel :2017-2125 and advt :1667-1708).
//
// The reference works level by level inside its block loop and carries the flux through the top face (AUX) and the flux
// velocities of level k+1 (FLUX_VEL_prev) from one level to the next; the outermost ghost ring, where a block cannot
// form flux velocities itself, is patched from a pre-pass with halo updates.  Here the same numbers come from whole-array
// launches:
//   k_lw_flux   column march: UTE, VTN and the vertical velocity WTKB of every level as 3-D fields (UTW(i) = UTE(i-1),
//               VTS(j) = VTN(j-1): the same expressions), then ONE halo update of the three fields gives the ring its
//               owners' values -- what the pre-pass patches in;
//   k_lw_z      vertical direction, parallel in (i,j,k): the flux through the top face of level k is the flux through the
//               bottom face of level k-1 (AUX(k) = AUXB(k-1)), so both are evaluated from the level's own neighbourhood;
//               -> XOUT, XSTAR on the whole array;
//   k_lw_x      zonal direction on every row: east-face values of cells i and i-1 from XSTAR(i-2 .. i+2) -> XOUT, XSTAR2;
//   k_lw_y      meridional direction + divergence term from XSTAR2(j-2 .. j+2) -> L(T), which the tracer right-hand side
//               reads instead of forming centred advection.
// Every used value is formed by the operations of the reference in their order (parity with the oracle's level-by-level
// restatement: tests/test_gpu_parity.py).  Correctness first: seven launches and nine 3-D work fields per step.
// PBC: the partial_bottom_cells branches (:629-640, :667-678, :2040-2062, :2110, :2757-2766, :2912-2944, :3065, :3140): face
// velocities weighted with the U-cell thicknesses, vertical fluxes and limiter coefficients with the T-cell thicknesses (a
// template flag, so the full-cell instantiation keeps its instructions).
#pragma once
#include "kernels_common.hpp"

namespace pop {

struct LwDev {
  double *UTE, *VTN, *WTKB;          // (nxb,nyb,km,block)
  double *XOUT[2], *XSTAR[2], *XSTAR2[2];
  const double *HTE, *HTN, *DXT, *DYT;
};

// flux velocities and vertical velocity of every level (comp_flux_vel :2068-2113); columns with i >= 1, j >= 1
template <bool PBC>
__global__ void __launch_bounds__(POP_COL_THREADS)
k_lw_flux(DevGrid g, LwDev w, const double *__restrict__ U, const double *__restrict__ V, const double *__restrict__ DH) {
  Col c;
  if (!col_setup(g, c, false)) return;
  const int nxb = g.nxb, km = g.km;
  const long long n2 = g.n2;
  const bool ok = c.i >= 1 && c.j >= 1;
  const int kmt = g.KMT[c.q2];
  double dyu00 = 0, dyu0m = 0, dyum0 = 0, dyumm = 0, dxu00 = 0, dxu0m = 0, dxum0 = 0, dxumm = 0;
  if (ok) {
    dyu00 = g.DYU[c.q2]; dyu0m = g.DYU[c.q2 - nxb]; dyum0 = g.DYU[c.q2 - 1]; dyumm = g.DYU[c.q2 - 1 - nxb];
    dxu00 = g.DXU[c.q2]; dxu0m = g.DXU[c.q2 - nxb]; dxum0 = g.DXU[c.q2 - 1]; dxumm = g.DXU[c.q2 - 1 - nxb];
  }
  const double tarear = g.TAREA_R[c.q2];
  int kmu00 = 0, kmu0m = 0, kmum0 = 0, kmumm = 0;
  double zb00 = 0, zb0m = 0, zbm0 = 0, zbmm = 0;
  if (PBC && ok) {
    kmu00 = g.KMU[c.q2]; kmu0m = g.KMU[c.q2 - nxb]; kmum0 = g.KMU[c.q2 - 1]; kmumm = g.KMU[c.q2 - 1 - nxb];
    zb00 = g.DZUB[c.q2]; zb0m = g.DZUB[c.q2 - nxb]; zbm0 = g.DZUB[c.q2 - 1]; zbmm = g.DZUB[c.q2 - 1 - nxb];
  }
  double wtk = DH[c.q2];
  for (int k = 1; k <= km; ++k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    double UTE = 0.0, VTN = 0.0, wtkb = 0.0;
    if (ok) {
      const double u00 = U[o], u0m = U[o - nxb], um0 = U[o - 1], umm = U[o - 1 - nxb];
      const double v00 = V[o], v0m = V[o - nxb], vm0 = V[o - 1], vmm = V[o - 1 - nxb];
      double UTW, VTS;
      if constexpr (PBC) {
        const double z00 = pbc_dz(g, k, kmu00, zb00), z0m = pbc_dz(g, k, kmu0m, zb0m), zm0 = pbc_dz(g, k, kmum0, zbm0), zmm = pbc_dz(g, k, kmumm, zbmm);
        UTE = 0.5 * (u00 * dyu00 * z00 + u0m * dyu0m * z0m);
        UTW = 0.5 * (um0 * dyum0 * zm0 + umm * dyumm * zmm);
        VTN = 0.5 * (v00 * dxu00 * z00 + vm0 * dxum0 * zm0);
        VTS = 0.5 * (v0m * dxu0m * z0m + vmm * dxumm * zmm);
      } else {
        UTE = 0.5 * (u00 * dyu00 + u0m * dyu0m);
        UTW = 0.5 * (um0 * dyum0 + umm * dyumm);
        VTN = 0.5 * (v00 * dxu00 + vm0 * dxum0);
        VTS = 0.5 * (v0m * dxu0m + vmm * dxumm);
      }
      if (k < km) { const double FC = (VTN - VTS + UTE - UTW) * tarear; wtkb = (k < kmt) ? (PBC ? wtk + FC : wtk + g.dz[k] * FC) : 0.0; }
    }
    w.UTE[o] = UTE; w.VTN[o] = VTN; w.WTKB[o] = wtkb;
    wtk = wtkb;
  }
}

// flux through the bottom face of level k of one column (lw_lim :3086-3126); x(m) = tracer at level m, wt / wb / wbp1 =
// vertical velocity at the top of levels k, k+1, k+2 (wt already 0 at k = 1: varthick surface layer)
// PBC: dzb = thickness of the column's bottom cell (level kmt)
template <bool PBC>
__device__ __forceinline__ double lw_auxb(const DevGrid &g, int k, int kmt, double dzb, double adv_dt, double adv_dt_r, double wt, double wb, double wbp1,
                                          double xkm1, double xk, double xkp1, double xkp2) {
  const int km = g.km;
  if (!(k + 1 <= kmt)) return 0.0;
  if constexpr (PBC) {   // :2912-2944 with :3086-3126
    const double zm = pbc_dz(g, k - 1, kmt, dzb), z0 = pbc_dz(g, k, kmt, dzb), z1 = pbc_dz(g, k + 1, kmt, dzb), z2 = pbc_dz(g, k + 2, kmt, dzb);
    const double dTR = xkp1 - xk;
    if (wb > 0.0) {
      const double LW = (z1 - adv_dt * wb) / (z0 + z1);
      double MU = 0.0;
      if (wbp1 > 0.0) MU = (z1 * adv_dt_r - wbp1) / wb;
      else if (wbp1 < 0.0) MU = -wbp1 / wb * (z1 + adv_dt * wbp1) / (z1 + z2);
      double r = wb * xkp1;
      if (k + 2 <= kmt) {
        const double dTRp1 = xkp2 - xkp1;
        if (dTR > 0.0 && dTRp1 > 0.0) r = wb * (xkp1 - fmin(LW * dTR, MU * dTRp1));
        else if (dTR < 0.0 && dTRp1 < 0.0) r = wb * (xkp1 - fmax(LW * dTR, MU * dTRp1));
      }
      return r;
    }
    if (wb < 0.0) {
      const double LW = (z0 + adv_dt * wb) / (z0 + z1);
      double MU = 0.0;
      if (wt < 0.0) MU = -(z0 * adv_dt_r + wt) / wb;
      else if (wt > 0.0) MU = -wt / wb * (z0 - adv_dt * wt) / (zm + z0);
      double r = wb * xk;
      if (k > 1) {
        const double dTRm1 = xk - xkm1;
        if (dTR > 0.0 && dTRm1 > 0.0) r = wb * (xk + fmin(LW * dTR, MU * dTRm1));
        else if (dTR < 0.0 && dTRm1 < 0.0) r = wb * (xk + fmax(LW * dTR, MU * dTRm1));
      }
      return r;
    }
    return 0.0;
  }
  const double pz_k = (k < km) ? 1.0 / (g.dz[k] + g.dz[k + 1]) : 0.5 / g.dz[km];
  const double dTR = xkp1 - xk;
  if (wb > 0.0) {
    const double work2 = g.dz[k + 1] * pz_k, work3 = adv_dt * pz_k;
    const double LW = work2 - work3 * wb;
    double MU = 0.0;
    if (wbp1 > 0.0) MU = (g.dz[k + 1] * adv_dt_r - wbp1) / wb;
    else if (wbp1 < 0.0) {
      const double pz_kp1 = (k + 1 < km) ? 1.0 / (g.dz[k + 1] + g.dz[k + 2]) : 0.5 / g.dz[km];
      MU = -wbp1 / wb * (g.dz[k + 1] + adv_dt * wbp1) * pz_kp1;
    }
    double r = wb * xkp1;
    if (k + 2 <= kmt) {
      const double dTRp1 = xkp2 - xkp1;
      if (dTR > 0.0 && dTRp1 > 0.0) r = wb * (xkp1 - fmin(LW * dTR, MU * dTRp1));
      else if (dTR < 0.0 && dTRp1 < 0.0) r = wb * (xkp1 - fmax(LW * dTR, MU * dTRp1));
    }
    return r;
  }
  if (wb < 0.0) {
    const double work1 = g.dz[k] * pz_k, work3 = adv_dt * pz_k;
    const double LW = work1 + work3 * wb;
    double MU = 0.0;
    if (wt < 0.0) MU = -(g.dz[k] * adv_dt_r + wt) / wb;
    else if (wt > 0.0) {
      const double pz_km1 = 1.0 / (g.dz[k - 1] + g.dz[k]);          // wt > 0 only for k > 1
      MU = -wt / wb * (g.dz[k] - adv_dt * wt) * pz_km1;
    }
    double r = wb * xk;
    if (k > 1) {
      const double dTRm1 = xk - xkm1;
      if (dTR > 0.0 && dTRm1 > 0.0) r = wb * (xk + fmin(LW * dTR, MU * dTRm1));
      else if (dTR < 0.0 && dTRm1 < 0.0) r = wb * (xk + fmax(LW * dTR, MU * dTRm1));
    }
    return r;
  }
  return 0.0;
}

// vertical direction (:3078-3136); whole array, one thread per (i,j,k), blockIdx.z = block * 2 + tracer
template <bool PBC>
__global__ void __launch_bounds__(256)
k_lw_z(DevGrid g, LwDev w, const double *__restrict__ X0, const double *__restrict__ X1, double adv_dt) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y + 1, b = blockIdx.z >> 1, n = blockIdx.z & 1;
  if (p2 >= g.n2) return;
  const int km = g.km;
  const long long n2 = g.n2, base3 = (long long)b * g.n3 + p2, o = base3 + (long long)(k - 1) * n2;
  const double *__restrict__ X = n ? X1 : X0;
  const int kmt = g.KMT[(long long)b * n2 + p2];
  const double adv_dt_r = 1.0 / adv_dt;
  auto xat = [&](int m) { return (m >= 1 && m <= km) ? X[base3 + (long long)(m - 1) * n2] : 0.0; };
  auto wat = [&](int m) { return (m >= 1 && m <= km) ? w.WTKB[base3 + (long long)(m - 1) * n2] : 0.0; };   // bottom of level m
  const double xkm2 = xat(k - 2), xkm1 = xat(k - 1), xk = xat(k), xkp1 = xat(k + 1), xkp2 = xat(k + 2);
  const double wb_km2 = wat(k - 2), wb_km1 = wat(k - 1), wb_k = wat(k), wb_kp1 = wat(k + 1);
  // top of level m: 0 at the surface (varthick), else the bottom of level m - 1
  const double wt_k = (k == 1) ? 0.0 : wb_km1, wt_km1 = (k - 1 <= 1) ? 0.0 : wb_km2;
  const double dzb = PBC ? g.DZBC[(long long)b * n2 + p2] : 0.0;
  const double aux = (k == 1) ? wt_k * xk : lw_auxb<PBC>(g, k - 1, kmt, dzb, adv_dt, adv_dt_r, wt_km1, wb_km1, wb_k, xkm2, xkm1, xk, xkp1);
  const double auxb = lw_auxb<PBC>(g, k, kmt, dzb, adv_dt, adv_dt_r, wt_k, wb_k, wb_kp1, xkm1, xk, xkp1, xkp2);
  const double xout = PBC ? (aux - auxb - (wt_k - wb_k) * xk) / pbc_dz(g, k, kmt, dzb) : (aux - auxb - (wt_k - wb_k) * xk) * g.dzr[k];
  w.XOUT[n][o] = xout;
  w.XSTAR[n][o] = xk - adv_dt * xout;
}

// value on the east face of cell (i, j) (:2981-3015 coefficients, :3138-3176 face value); q = cell index in the 2-D block
// array, o = in the level's 3-D slab
// thickness of T cell q at level k
__device__ __forceinline__ double lw_dzt(const DevGrid &g, int k, long long q) { return pbc_dz(g, k, g.KMT[q], g.DZBC[q]); }
template <bool PBC>
__device__ __forceinline__ double lw_face_x(const DevGrid &g, const LwDev &w, int k, long long q, long long o, const double *__restrict__ XS, double adv_dt) {
  auto udt = [&](long long d) {   // UTE_to_UVEL_E: 1 / HTE, with partial bottom cells 1 / HTE / min(DZT(i), DZT(i+1)) (:629-640)
    if constexpr (PBC) return adv_dt * w.UTE[o + d] * (1.0 / w.HTE[q + d] / fmin(lw_dzt(g, k, q + d), lw_dzt(g, k, q + d + 1)));
    else return adv_dt * w.UTE[o + d] * (1.0 / w.HTE[q + d]);
  };
  auto kmaske = [&](long long d) { return (k <= g.KMT[q + d] && k <= g.KMT[q + d + 1]) ? 1.0 : 0.0; };
  auto px = [&](long long d) { return 1.0 / (w.DXT[q + d] + w.DXT[q + d + 1]); };
  const double U0 = udt(0);
  const double CE = PBC ? w.UTE[o] * (g.TAREA_R[q] / lw_dzt(g, k, q)) : w.UTE[o] * g.TAREA_R[q];
  const double dTR = (XS[o + 1] - XS[o]) * kmaske(0);
  if (U0 > 0.0) {
    const double Um = udt(-1);
    const double LW = (w.DXT[q] - U0) * px(0);
    double MU = (Um > 0.0) ? (w.DXT[q] - Um) / U0 : 0.0;
    if (Um < 0.0) MU = -Um / U0 * ((w.DXT[q] + Um) * px(-1));        // LW_x(i-1) of a westward face
    if (CE > 0.0) {
      const double dTRm1 = (XS[o] - XS[o - 1]) * kmaske(-1);
      if (dTR > 0.0 && dTRm1 > 0.0) return XS[o] + fmin(LW * dTR, MU * dTRm1);
      if (dTR < 0.0 && dTRm1 < 0.0) return XS[o] + fmax(LW * dTR, MU * dTRm1);
      return XS[o];
    }
    return XS[o] + LW * dTR;                                          // CE and U0 have the same sign; kept for exactness
  }
  if (U0 < 0.0) {
    const double Up = udt(1);
    const double LW = (w.DXT[q + 1] + U0) * px(0);
    double MU = (Up < 0.0) ? -(w.DXT[q + 1] + Up) / U0 : 0.0;
    if (Up > 0.0) MU = -Up / U0 * ((w.DXT[q + 1] - Up) * px(1));      // LW_x(i+1) of an eastward face
    if (CE < 0.0) {
      const double dTRp1 = (XS[o + 2] - XS[o + 1]) * kmaske(1);
      if (dTR > 0.0 && dTRp1 > 0.0) return XS[o + 1] - fmin(LW * dTR, MU * dTRp1);
      if (dTR < 0.0 && dTRp1 < 0.0) return XS[o + 1] - fmax(LW * dTR, MU * dTRp1);
      return XS[o + 1];
    }
    return XS[o] + LW * dTR;
  }
  return XS[o] + (w.DXT[q] * px(0)) * dTR;
}
// zonal direction (:3138-3196): columns ib..ie of EVERY row (the meridional pass reads two rows beyond the physical domain)
template <bool PBC>
__global__ void __launch_bounds__(256)
k_lw_x(DevGrid g, LwDev w, const double *__restrict__ X0, const double *__restrict__ X1, double adv_dt) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y + 1, b = blockIdx.z >> 1, n = blockIdx.z & 1;
  if (p2 >= g.n2) return;
  const int i = p2 % g.nxb;
  if (i + 1 < g.ib || i + 1 > blk_ie(g, b)) return;
  const long long n2 = g.n2, q = (long long)b * n2 + p2, o = (long long)b * g.n3 + (long long)(k - 1) * n2 + p2;
  const double *__restrict__ X = n ? X1 : X0;
  const double *__restrict__ XS = w.XSTAR[n];
  const double te = lw_face_x<PBC>(g, w, k, q, o, XS, adv_dt), tew = lw_face_x<PBC>(g, w, k, q - 1, o - 1, XS, adv_dt);
  const double tar = PBC ? g.TAREA_R[q] / lw_dzt(g, k, q) : g.TAREA_R[q];   // :2761
  const double CE = w.UTE[o] * tar, CW = -w.UTE[o - 1] * tar;
  const double work1 = CE * te + CW * tew - (CE + CW) * X[o];
  w.XOUT[n][o] = w.XOUT[n][o] + work1;
  w.XSTAR2[n][o] = XS[o] - adv_dt * work1;
}

// value on the north face of cell (i, j) (:3019-3060, :3204-3240)
template <bool PBC>
__device__ __forceinline__ double lw_face_y(const DevGrid &g, const LwDev &w, int k, long long q, long long o, const double *__restrict__ XS, double adv_dt) {
  const long long nx = g.nxb;
  auto vdt = [&](long long d) {   // VTN_to_VVEL_N (:667-678)
    if constexpr (PBC) return adv_dt * w.VTN[o + d * nx] * (1.0 / w.HTN[q + d * nx] / fmin(lw_dzt(g, k, q + d * nx), lw_dzt(g, k, q + (d + 1) * nx)));
    else return adv_dt * w.VTN[o + d * nx] * (1.0 / w.HTN[q + d * nx]);
  };
  auto kmaskn = [&](long long d) { return (k <= g.KMT[q + d * nx] && k <= g.KMT[q + (d + 1) * nx]) ? 1.0 : 0.0; };
  auto py = [&](long long d) { return 1.0 / (w.DYT[q + d * nx] + w.DYT[q + (d + 1) * nx]); };
  const double V0 = vdt(0);
  const double CN = PBC ? w.VTN[o] * (g.TAREA_R[q] / lw_dzt(g, k, q)) : w.VTN[o] * g.TAREA_R[q];
  const double dTR = (XS[o + nx] - XS[o]) * kmaskn(0);
  if (V0 > 0.0) {
    const double Vm = vdt(-1);
    const double LW = (w.DYT[q] - V0) * py(0);
    double MU = (Vm > 0.0) ? (w.DYT[q] - Vm) / V0 : 0.0;
    if (Vm < 0.0) MU = -Vm / V0 * ((w.DYT[q] + Vm) * py(-1));
    if (CN > 0.0) {
      const double dTRm1 = (XS[o] - XS[o - nx]) * kmaskn(-1);
      if (dTR > 0.0 && dTRm1 > 0.0) return XS[o] + fmin(LW * dTR, MU * dTRm1);
      if (dTR < 0.0 && dTRm1 < 0.0) return XS[o] + fmax(LW * dTR, MU * dTRm1);
      return XS[o];
    }
    return XS[o] + LW * dTR;
  }
  if (V0 < 0.0) {
    const double Vp = vdt(1);
    const double LW = (w.DYT[q + nx] + V0) * py(0);
    double MU = (Vp < 0.0) ? -(w.DYT[q + nx] + Vp) / V0 : 0.0;
    if (Vp > 0.0) MU = -Vp / V0 * ((w.DYT[q + nx] - Vp) * py(1));
    if (CN < 0.0) {
      const double dTRp1 = (XS[o + 2 * nx] - XS[o + nx]) * kmaskn(1);
      if (dTR > 0.0 && dTRp1 > 0.0) return XS[o + nx] - fmin(LW * dTR, MU * dTRp1);
      if (dTR < 0.0 && dTRp1 < 0.0) return XS[o + nx] - fmax(LW * dTR, MU * dTRp1);
      return XS[o + nx];
    }
    return XS[o] + LW * dTR;
  }
  return XS[o] + (w.DYT[q] * py(0)) * dTR;
}
// meridional direction + divergence term (:3204-3262) on the physical cells -> L(T) in XOUT
template <bool PBC>
__global__ void __launch_bounds__(256)
k_lw_y(DevGrid g, LwDev w, const double *__restrict__ X0, const double *__restrict__ X1, double adv_dt) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y + 1, b = blockIdx.z >> 1, n = blockIdx.z & 1;
  if (p2 >= g.n2) return;
  const int i = p2 % g.nxb, j = p2 / g.nxb, nxb = g.nxb;
  if (i + 1 < g.ib || i + 1 > blk_ie(g, b) || j + 1 < g.jb || j + 1 > blk_je(g, b)) return;
  const long long n2 = g.n2, q = (long long)b * n2 + p2, o = (long long)b * g.n3 + (long long)(k - 1) * n2 + p2;
  const double *__restrict__ X = n ? X1 : X0;
  const double *__restrict__ XS = w.XSTAR2[n];
  const double tn = lw_face_y<PBC>(g, w, k, q, o, XS, adv_dt), tns = lw_face_y<PBC>(g, w, k, q - nxb, o - nxb, XS, adv_dt);
  const double dztk = PBC ? lw_dzt(g, k, q) : 0.0;
  const double tar = PBC ? g.TAREA_R[q] / dztk : g.TAREA_R[q];
  const double FVN = w.VTN[o] * tar, FVS = -w.VTN[o - nxb] * tar, FUE = w.UTE[o] * tar, FUW = -w.UTE[o - 1] * tar;
  const double wt = (k == 1) ? 0.0 : w.WTKB[o - n2], wb = w.WTKB[o];
  const double DIV = (PBC ? (wt - wb) / dztk : (wt - wb) * g.dzr[k]) + FUE + FUW + FVN + FVS;   // :3065-3069
  w.XOUT[n][o] = w.XOUT[n][o] + FVN * tn + FVS * tns - (FVN + FVS - DIV) * X[o];
}

}  // namespace pop
