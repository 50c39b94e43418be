// kernels_momentum_lds.hpp -- momentum right-hand side with the 3x3 stencils staged through LDS.
//
// Same arithmetic, evaluation order and results as k_momentum_rhs (kernels_baroclinic.hpp; clinic,
// baroclinic.F90:1724-1890); what changes is where the neighbours come from.  A workgroup owns a
// 64 x R tile of columns and marches k.  Per level every thread loads only ITS cell of the stencil
// fields (U, V at curtime, the time-averaged density, U, V at mixtime) plus -- for the first
// (64+2)(R+2) - 64R threads -- one halo cell, writes them into an LDS tile and reads its 8
// neighbours from there: 14 instead of 43 global loads per thread and level.  The tile is double
// buffered (one barrier per level) and the next level's cells are loaded before the current level
// is computed, so their latency hides behind the arithmetic.
#pragma once
#include "kernels_baroclinic.hpp"

namespace pop {

template <int R>
struct MomTile {
  static constexpr int W = POP_COL_THREADS + 2, H = R + 2, N = W * H, NHALO = N - POP_COL_THREADS * R;
  double u[2][N], v[2][N], f[2][N], um[2][N], vm[2][N];
  static_assert(NHALO <= POP_COL_THREADS * R, "every halo cell needs a thread");
};

// PF: how many levels ahead the cells of the stencil fields are loaded (1: the next level while this one is computed;
// 2: two levels in flight, 26 more registers)
template <int R, int PF = 1>
__global__ void __launch_bounds__(POP_COL_THREADS * R)
k_momentum_rhs_lds(DevGrid g, StepParams sp, MomentumRhsArgs a, int tj_first, int tj_count) {
  using T = MomTile<R>;
  __shared__ T t;
  const int nxb = g.nxb, nyb = g.nyb, km = g.km;
  const long long n2 = g.n2;
  // tiles start at the first physical column/row (0-based NGHOST); this launch covers the tile rows tj_first .. +tj_count
  const int tiles_i = (nxb - 2 * NGHOST + POP_COL_THREADS - 1) / POP_COL_THREADS;
  const int b = blockIdx.y;
  int ti, tj;
  bool listed = false;
  if (tj_first == 0 && tj_count == (nyb - 2 * NGHOST + R - 1) / R) { if (!lds_tile_active<R>(g, b, tiles_i, tj_count, ti, tj, listed)) return; }
  else if (!lds_tile(g.lds_order, tiles_i, tj_count, ti, tj)) return;
  tj += tj_first;
  const int i0 = NGHOST + ti * POP_COL_THREADS, j0 = NGHOST + tj * R;
  if (!listed && land_tile(g, b, i0, POP_COL_THREADS, j0, R)) return;   // no ocean column in the tile: U, V stay 0 there
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * POP_COL_THREADS + tx;
  const int i = i0 + tx, j = j0 + ty;
  const bool inb = i < nxb && j < nyb;                       // cell exists (tiles may overhang the array)
  const bool act = inb && i + 1 <= g.ie && j + 1 <= g.je;    // physical column (i0, j0 are already >= ib, jb)
  const int p2 = inb ? j * nxb + i : 0;
  const long long q2 = (long long)b * n2 + p2, base3 = (long long)b * g.n3 + p2;
  const int lc = (ty + 1) * T::W + tx + 1;                   // this cell inside the tile
  // halo duty: thread tid < NHALO owns halo cell number tid
  int hl = -1; long long hbase = 0; bool hin = false;
  if (tid < T::NHALO) {
    int li, lj;
    if (tid < T::W) { lj = 0; li = tid; }
    else if (tid < 2 * T::W) { lj = T::H - 1; li = tid - T::W; }
    else if (tid < 2 * T::W + R) { lj = 1 + (tid - 2 * T::W); li = 0; }
    else { lj = 1 + (tid - 2 * T::W - R); li = T::W - 1; }
    hl = lj * T::W + li;
    const int hi = i0 - 1 + li, hj = j0 - 1 + lj;
    hin = hi >= 0 && hi < nxb && hj >= 0 && hj < nyb;
    hbase = (long long)b * g.n3 + (hin ? hj * nxb + hi : 0);
  }
  // per-column constants (as k_momentum_rhs)
  const int kmu = act ? g.KMU[q2] : 0;
  double dyu[3][3], dxu[3][3];
#pragma unroll
  for (int dj = -1; dj <= 1; ++dj)
#pragma unroll
    for (int di = -1; di <= 1; ++di) {
      dyu[dj + 1][di + 1] = act ? g.DYU[q2 + dj * nxb + di] : 0.0;
      dxu[dj + 1][di + 1] = act ? g.DXU[q2 + dj * nxb + di] : 0.0;
    }
  double uar = 0, fcor = 0, kxu = 0, kyu = 0, dxur = 0, dyur = 0, hur = 0, cc_h = 0, dun = 0, dus = 0, due = 0, duw = 0;
  double dmc = 0, dmn = 0, dms = 0, dme = 0, dmw = 0, smfx = 0, smfy = 0, wuk = 0;
  if (act) {
    uar = g.UAREA_R[q2]; fcor = g.FCOR[q2]; kxu = g.KXU[q2]; kyu = g.KYU[q2];
    dxur = g.DXUR[q2]; dyur = g.DYUR[q2]; hur = g.HUR[q2];
    cc_h = g.DUC[q2] + g.DUM[q2];
    dun = g.DUN[q2]; dus = g.DUS[q2]; due = g.DUE[q2]; duw = g.DUW[q2];
    dmc = g.DMC[q2]; dmn = g.DMN[q2]; dms = g.DMS[q2]; dme = g.DME[q2]; dmw = g.DMW[q2];
    smfx = (kmu >= 1) ? g.SMF1[q2] : 0.0; smfy = (kmu >= 1) ? g.SMF2[q2] : 0.0;
    wuk = a.DHU[q2];
  }
  const double amf_next = (act && a.D2N[0]) ? a.AMF[q2] : 0.0;
  double vuf = smfx, vvf = smfy;
  double rhokmx = 0.0, rhokmy = 0.0, sumx = 0.0, sumy = 0.0, zx = 0.0, zy = 0.0;
  double uc_km1 = 0.0, vc_km1 = 0.0;
  // cell values of one level: stencil fields + the column-only fields
  struct Lev { double u, v, f, um, vm, uo, vo, vvc; };
  struct Hal { double u, v, f, um, vm; };
  auto rho_f = [&](long long o, int k) {
    if (sp.pavg) return 0.25 * (a.RHONEW[o] + 2.0 * a.RHOCUR[o] + a.RHOOLD[o]) * g.bouss[k];
    return a.RHOCUR[o] * g.bouss[k];
  };
  auto load_cell = [&](int k) {
    Lev L{0, 0, 0, 0, 0, 0, 0, 0};
    if (inb) {
      const long long o = base3 + (long long)(k - 1) * n2;
      L.u = a.UCUR[o]; L.v = a.VCUR[o]; L.f = rho_f(o, k); L.um = a.UMIX[o]; L.vm = a.VMIX[o];
      L.uo = a.UOLD[o]; L.vo = a.VOLD[o]; L.vvc = a.VVC[o];
    }
    return L;
  };
  auto load_halo = [&](int k) {
    Hal Hh{0, 0, 0, 0, 0};
    if (hin) {
      const long long o = hbase + (long long)(k - 1) * n2;
      Hh.u = a.UCUR[o]; Hh.v = a.VCUR[o]; Hh.f = rho_f(o, k); Hh.um = a.UMIX[o]; Hh.vm = a.VMIX[o];
    }
    return Hh;
  };
  Lev cur = load_cell(1);
  Hal hal = load_halo(1);
  Lev nxt = cur; Hal nhal = hal;
  if (PF == 2) { nxt = load_cell(km > 1 ? 2 : 1); nhal = load_halo(km > 1 ? 2 : 1); }
  double *__restrict__ const UNp = a.UNEW;
  double *__restrict__ const VNp = a.VNEW;
  for (int k = 1; k <= km; ++k) {
    const int buf = k & 1;
    t.u[buf][lc] = cur.u; t.v[buf][lc] = cur.v; t.f[buf][lc] = cur.f; t.um[buf][lc] = cur.um; t.vm[buf][lc] = cur.vm;
    if (hl >= 0) { t.u[buf][hl] = hal.u; t.v[buf][hl] = hal.v; t.f[buf][hl] = hal.f; t.um[buf][hl] = hal.um; t.vm[buf][hl] = hal.vm; }
    const int kp1 = (k < km) ? k + 1 : km;
    Lev nx2 = nxt; Hal nh2 = nhal;
    if (PF == 1) { nxt = load_cell(kp1); nhal = load_halo(kp1); }          // in flight while this level is computed
    else { const int kp2 = (k + 2 <= km) ? k + 2 : km; nx2 = load_cell(kp2); nh2 = load_halo(kp2); }
    __syncthreads();
    if (act) {
      const long long o = base3 + (long long)(k - 1) * n2;
      const double uc_k = cur.u, vc_k = cur.v, uo_k = cur.uo, vo_k = cur.vo;
      const double uc_kp1 = nxt.u, vc_kp1 = nxt.v, uo_kp1 = nxt.uo, vo_kp1 = nxt.vo;
      double u[3][3], v[3][3], ud[3][3], vd[3][3];
#pragma unroll
      for (int dj = -1; dj <= 1; ++dj)
#pragma unroll
        for (int di = -1; di <= 1; ++di) {
          const double uu = t.u[buf][lc + dj * T::W + di], vv = t.v[buf][lc + dj * T::W + di];
          u[dj + 1][di + 1] = uu; v[dj + 1][di + 1] = vv;
          ud[dj + 1][di + 1] = uu * dyu[dj + 1][di + 1];
          vd[dj + 1][di + 1] = vv * dxu[dj + 1][di + 1];
        }
#define UD(di, dj) ud[(dj) + 1][(di) + 1]
#define VD(di, dj) vd[(dj) + 1][(di) + 1]
#define UU(di, dj) u[(dj) + 1][(di) + 1]
#define VV(di, dj) v[(dj) + 1][(di) + 1]
      const double UUW = 0.25 * (UD(0, 0) + UD(-1, 0)) + 0.125 * (UD(0, -1) + UD(-1, -1) + UD(0, 1) + UD(-1, 1));
      const double UUE = 0.25 * (UD(1, 0) + UD(0, 0)) + 0.125 * (UD(1, -1) + UD(0, -1) + UD(1, 1) + UD(0, 1));
      const double VUS = 0.25 * (VD(0, 0) + VD(0, -1)) + 0.125 * (VD(-1, 0) + VD(-1, -1) + VD(1, 0) + VD(1, -1));
      const double VUN = 0.25 * (VD(0, 1) + VD(0, 0)) + 0.125 * (VD(-1, 1) + VD(-1, 0) + VD(1, 1) + VD(1, 0));
      const double wukb = wuk + g.c2dz[k] * 0.5 * (VUN - VUS + UUE - UUW) * uar;
      const double cc = VUN - VUS + UUE - UUW;
      double LU = 0.5 * (cc * UU(0, 0) + VUN * UU(0, 1) - VUS * UU(0, -1) + UUE * UU(1, 0) - UUW * UU(-1, 0)) * uar;
      double LV = 0.5 * (cc * VV(0, 0) + VUN * VV(0, 1) - VUS * VV(0, -1) + UUE * VV(1, 0) - UUW * VV(-1, 0)) * uar;
      if (k == 1) { LU = LU + g.dzr[k] * wuk * uc_k; LV = LV + g.dzr[k] * wuk * vc_k; }
      else { LU = LU + g.dz2r[k] * wuk * (uc_km1 + uc_k); LV = LV + g.dz2r[k] * wuk * (vc_km1 + vc_k); }
      if (k < km) { LU = LU - g.dz2r[k] * wukb * (uc_k + uc_kp1); LV = LV - g.dz2r[k] * wukb * (vc_k + vc_kp1); }
      if (k <= kmu) {
        LU = LU + uc_k * vc_k * kyu - vc_k * vc_k * kxu;
        LV = LV + uc_k * vc_k * kxu - uc_k * uc_k * kyu;
      } else { LU = 0.0; LV = 0.0; }
      double FX = -LU, FY = -LV;
      if (sp.impcor && sp.leapfrogts) {
        FX = FX + fcor * (sp.gamma * vc_k + (1.0 - sp.gamma) * vo_k);
        FY = FY - fcor * (sp.gamma * uc_k + (1.0 - sp.gamma) * uo_k);
      } else if (!sp.impcor && sp.leapfrogts) {
        FX = FX + fcor * vc_k; FY = FY - fcor * uc_k;
      } else {
        FX = FX + fcor * vo_k; FY = FY - fcor * uo_k;
      }
      {
        const double f00 = t.f[buf][lc], f10 = t.f[buf][lc + 1], f01 = t.f[buf][lc + T::W], f11 = t.f[buf][lc + T::W + 1];
        double rhokx = 0.0, rhoky = 0.0;
        if (k <= kmu) {
          rhokx = dxur * 0.5 * (f11 - f00 - f01 + f10);
          rhoky = dyur * 0.5 * (f11 - f00 + f01 - f10);
        }
        if (k == 1) { rhokmx = rhokx; rhokmy = rhoky; sumx = 0.0; sumy = 0.0; }
        const double factor = g.dzw[k - 1] * sp.grav * 0.5;
        sumx = sumx + factor * (rhokx + rhokmx);
        sumy = sumy + factor * (rhoky + rhokmy);
        rhokmx = rhokx; rhokmy = rhoky;
        FX = FX - sumx; FY = FY - sumy;
      }
      {
        const double um0 = t.um[buf][lc], umn = t.um[buf][lc + T::W], ums = t.um[buf][lc - T::W], ume = t.um[buf][lc + 1], umw = t.um[buf][lc - 1];
        const double vm0 = t.vm[buf][lc], vmn = t.vm[buf][lc + T::W], vms = t.vm[buf][lc - T::W], vme = t.vm[buf][lc + 1], vmw = t.vm[buf][lc - 1];
        double hdu = sp.am * ((cc_h * um0 + dun * umn + dus * ums + due * ume + duw * umw) +
                              (dmc * vm0 + dmn * vmn + dms * vms + dme * vme + dmw * vmw));
        double hdv = sp.am * ((cc_h * vm0 + dun * vmn + dus * vms + due * vme + duw * vmw) -
                              (dmc * um0 + dmn * umn + dms * ums + dme * ume + dmw * umw));
        if (k > kmu) { hdu = 0.0; hdv = 0.0; }
        FX = FX + hdu; FY = FY + hdv;
      }
      if (a.D2N[0]) {   // hdiffu_del4's first Laplacian of the current velocity, for the next step (same expression as k_del4_d2u)
        double du = 0.0, dv = 0.0;
        if (k <= kmu) {
          const double u0 = t.u[buf][lc], un = t.u[buf][lc + T::W], us = t.u[buf][lc - T::W], ue = t.u[buf][lc + 1], uw = t.u[buf][lc - 1];
          const double v0 = t.v[buf][lc], vn = t.v[buf][lc + T::W], vs = t.v[buf][lc - T::W], ve = t.v[buf][lc + 1], vw = t.v[buf][lc - 1];
          du = (cc_h * u0 + dun * un + dus * us + due * ue + duw * uw) + (dmc * v0 + dmn * vn + dms * vs + dme * ve + dmw * vw);
          dv = (cc_h * v0 + dun * vn + dus * vs + due * ve + duw * vw) - (dmc * u0 + dmn * un + dms * us + dme * ue + dmw * uw);
          du = amf_next * du; dv = amf_next * dv;
        }
        a.D2N[0][o] = du; a.D2N[1][o] = dv;
      }
      {
        const double vvc = cur.vvc;
        double vufb = vvc * (uo_k - uo_kp1) * g.dzwr[k];
        double vvfb = vvc * (vo_k - vo_kp1) * g.dzwr[k];
        if (k == kmu) {
          const double vmag = sp.bottom_drag * sqrt(uo_k * uo_k + vo_k * vo_k);
          vufb = vmag * uo_k; vvfb = vmag * vo_k;
        }
        const double vdu = (k <= kmu) ? (vuf - vufb) * g.dzr[k] : 0.0;
        const double vdv = (k <= kmu) ? (vvf - vvfb) * g.dzr[k] : 0.0;
        vuf = vufb; vvf = vvfb;
        FX = FX + vdu; FY = FY + vdv;
      }
      if (k > kmu) { FX = 0.0; FY = 0.0; }
      if (sp.impcor) {
        const double W1 = sp.c2dtu * sp.beta * fcor;
        const double W2 = sp.c2dtu / (1.0 + W1 * W1);
        UNp[o] = (FX + W1 * FY) * W2;
        VNp[o] = (FY - W1 * FX) * W2;
      } else { UNp[o] = sp.c2dtu * FX; VNp[o] = sp.c2dtu * FY; }
      zx = zx + FX * g.dz[k]; zy = zy + FY * g.dz[k];
      wuk = wukb;
      uc_km1 = uc_k; vc_km1 = vc_k;
#undef UD
#undef VD
#undef UU
#undef VV
    }
    cur = nxt; hal = nhal;
    if (PF == 2) { nxt = nx2; nhal = nh2; }
  }
  if (act) { a.ZX[q2] = zx * hur; a.ZY[q2] = zy * hur; }
}

// tj_first / tj_count: window of tile rows (default: all) -- the rim / interior split around a halo exchange
template <int R>
inline void launch_momentum_lds(const DevGrid &g, const StepParams &sp, const MomentumRhsArgs &a, hipStream_t st, int tj_first = 0, int tj_count = -1) {
  const int tiles_i = (g.nxb - 2 * NGHOST + POP_COL_THREADS - 1) / POP_COL_THREADS;
  const int tiles_j = (g.nyb - 2 * NGHOST + R - 1) / R;
  if (tj_count < 0) tj_count = tiles_j - tj_first;
  if (tj_count <= 0) return;
  const bool whole = tj_first == 0 && tj_count == tiles_j;
  const dim3 G(whole ? lds_launch_x<R>(g, tiles_i, tiles_j) : lds_grid_x(g.lds_order, tiles_i, tj_count), g.nblocks), B(POP_COL_THREADS, R);
  hipLaunchKernelGGL((k_momentum_rhs_lds<R, 1>), G, B, 0, st, g, sp, a, tj_first, tj_count);
}

}  // namespace pop
