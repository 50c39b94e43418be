// kernels_momentum_lds.hpp -- momentum right-hand side with the 3x3 stencils staged through LDS.
//
// Same arithmetic, evaluation order and results as k_momentum_rhs (kernels_baroclinic.hpp; clinic,
// baroclinic.F90:1724-1890); what changes is where the neighbours come from and how the level loop is
// scheduled.  A workgroup owns a 64 x R tile of columns and marches k.  Per level every thread loads only
// ITS cell of the stencil fields plus -- for the first (64+2)(R+2) - 64R threads -- one halo cell, writes
// them into a double-buffered LDS tile (one barrier per level) and reads its neighbours from there.
//
// Round 3 (what the ISA of the round-2 kernel showed):
//  * The transports U*DYU(*DZU), V*DXU(*DZU) of advu are staged as PRODUCTS formed by the cell's owner: the nine
//    neighbour values of DYU / DXU per column (18 register doubles, 18 multiplies per level) are gone, and with
//    partial bottom cells the owner multiplies by its own thickness -- the same product of the same operands.
//  * The level body is branch-free: loads are issued by every lane at a clamped (always valid) address, stores go
//    to the field or -- for lanes outside the physical domain -- to a dump area.  With loads and stores inside
//    divergent `if` blocks the compiler had to wait for `vmcnt(0)` before the arithmetic of a level, i.e. for the
//    whole prefetch of the next level: no latency was hidden.  Straight-line code gets counted waits.
//  * Vertical constants (dz(k), dzr(k), ...) come through the constant address space (DevGrid::CArr): scalar
//    loads instead of three uniform VECTOR loads per level, each with its own vmcnt(0).
#pragma once
#include "kernels_baroclinic.hpp"

namespace pop {

template <int R, bool PBC>
struct MomTile {
  static constexpr int W = POP_COL_THREADS + 2, H = R + 2, N = W * H, NHALO = N - POP_COL_THREADS * R;
  double u[2][N], v[2][N], ud[2][N], vd[2][N], f[2][N], um[2][N], vm[2][N];
  double dzu[PBC ? 2 : 1][PBC ? N : 1];      // partial bottom cells: thickness of the U cell at the level
  static_assert(NHALO <= POP_COL_THREADS * R, "every halo cell needs a thread");
};

// PBC: partial bottom cells (advection.F90:1245-1300, 1352, 1381-1467; hmix_del2.F90:852-886 / hmix_del4.F90:683-812;
// vertical_mix.F90:946-995; baroclinic.F90:1037-1039) with DZU formed from KMU / DZUB (pbc_dz)
template <int R, bool PBC = false>
__global__ void __launch_bounds__(POP_COL_THREADS * R)
k_momentum_rhs_lds(DevGrid g, StepParams sp, MomentumRhsArgs a, int tj_first, int tj_count) {
#ifdef POP_PROBE_MOM_CONTRACT
  // MEASUREMENT PROBE, never part of libpop_amd.so (profiles/r04_probe_mom_contract.txt): the same kernel with multiply-adds fused, to
  // see what the instruction count of a level is worth.  Results are no longer those of the reference's operation order.
#pragma clang fp contract(fast)
#endif
  using T = MomTile<R, PBC>;
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
  const bool act = inb && i + 1 <= blk_ie(g, b) && j + 1 <= blk_je(g, b);    // physical column (i0, j0 are already >= ib, jb)
  // every lane addresses a cell that exists: lanes beyond the array read cell 0 of the block (their values are never used: the
  // stencil of a physical column stays inside the block) and store to the dump area
  const int p2 = inb ? j * nxb + i : 0;
  const long long q2 = (long long)b * n2 + p2, base3 = (long long)b * g.n3 + p2;
  const int lc = (ty + 1) * T::W + tx + 1;                   // this cell inside the tile
  // halo duty: thread tid < NHALO owns halo cell number tid; the others rewrite their own cell (same value, no branch)
  int hl = lc; long long hq2 = q2, hbase = base3;
  if (tid < T::NHALO) {
    int li, lj;
    if (tid < T::W) { lj = 0; li = tid; }
    else if (tid < 2 * T::W) { lj = T::H - 1; li = tid - T::W; }
    else if (tid < 2 * T::W + R) { lj = 1 + (tid - 2 * T::W); li = 0; }
    else { lj = 1 + (tid - 2 * T::W - R); li = T::W - 1; }
    hl = lj * T::W + li;
    const int hi = i0 - 1 + li, hj = j0 - 1 + lj;
    const bool hin = hi >= 0 && hi < nxb && hj >= 0 && hj < nyb;
    const int hp2 = hin ? hj * nxb + hi : 0;
    hq2 = (long long)b * n2 + hp2; hbase = (long long)b * g.n3 + hp2;
  }
  // per-column constants
  const int kmu_own = g.KMU[q2];
  const int kmu = act ? kmu_own : 0;
  const double dyu_o = g.DYU[q2], dxu_o = g.DXU[q2], dyu_h = g.DYU[hq2], dxu_h = g.DXU[hq2];
  const double dzub_o = PBC ? g.DZUB[q2] : 0.0, dzub_h = PBC ? g.DZUB[hq2] : 0.0;
  const int kmu_h = PBC ? g.KMU[hq2] : 0;
  const double uar = g.UAREA_R[q2], fcor = g.FCOR[q2], kxu = g.KXU[q2], kyu = g.KYU[q2];
  const double dxur = g.DXUR[q2], dyur = g.DYUR[q2], hur = g.HUR[q2];
  const double cc_h = g.DUC[q2] + g.DUM[q2];
  const double dun = g.DUN[q2], dus = g.DUS[q2], due = g.DUE[q2], duw = g.DUW[q2];
  const double dmc = g.DMC[q2], dmn = g.DMN[q2], dms = g.DMS[q2], dme = g.DME[q2], dmw = g.DMW[q2];
  const double smf1 = g.SMF1[q2], smf2 = g.SMF2[q2];
  const double smfx = (kmu >= 1) ? smf1 : 0.0, smfy = (kmu >= 1) ? smf2 : 0.0;
  double wuk = a.DHU[q2];
  const bool d2n = a.D2N[0] != nullptr;
  const double amf_next = (d2n ? a.AMF : g.UAREA_R)[q2];     // (unused without D2N; the load stays unconditional)
  double vuf = smfx, vvf = smfy;
  double rhokmx = 0.0, rhokmy = 0.0, sumx = 0.0, sumy = 0.0, zx = 0.0, zy = 0.0;
  double uc_km1 = 0.0, vc_km1 = 0.0;
  // outputs: the field for physical columns, the dump area for the other lanes (one store instruction either way)
  double *const dump = g.dump + tid;
  double *__restrict__ const UNp = act ? a.UNEW + base3 : dump;
  double *__restrict__ const VNp = act ? a.VNEW + base3 : dump + 512;
  double *__restrict__ const D0p = (act && d2n) ? a.D2N[0] + base3 : dump + 1024;
  double *__restrict__ const D1p = (act && d2n) ? a.D2N[1] + base3 : dump + 1536;
  const long long ostep = act ? n2 : 0;                      // level stride of the outputs (dump: none)
  // density factor of the pressure gradient: time-averaged with pressure averaging (three fields), else the current density read
  // three times (same address: two of the loads hit L1) so that the load count of a level does not depend on the step type
  const double *__restrict__ const RN = sp.pavg ? a.RHONEW : a.RHOCUR;
  const double *__restrict__ const RO = sp.pavg ? a.RHOOLD : a.RHOCUR;
  const bool pavg = sp.pavg != 0;
  // cell values of one level.  Own: the column's own velocities (current and old time), which level k - 1 also needs (vertical
  // advection and diffusion through its bottom face): they are loaded TWO levels ahead, so that no level waits for a load issued
  // in its own iteration.  Lev / Hal: the other fields of the cell and of the halo cell, one level ahead.
  struct Own { double u, v, uo, vo; };
  struct Lev { double rn, rc, ro, um, vm, vvc; };
  struct Hal { double u, v, rn, rc, ro, um, vm; };
  auto load_own = [&](int k) {
    Own L;
    const long long o = base3 + (long long)(k - 1) * n2;
    L.u = a.UCUR[o]; L.v = a.VCUR[o]; L.uo = a.UOLD[o]; L.vo = a.VOLD[o];
    return L;
  };
  auto load_cell = [&](int k) {
    Lev L;
    const long long o = base3 + (long long)(k - 1) * n2;
    L.rn = RN[o]; L.rc = a.RHOCUR[o]; L.ro = RO[o]; L.um = a.UMIX[o]; L.vm = a.VMIX[o]; L.vvc = a.VVC[o];
    return L;
  };
  auto load_halo = [&](int k) {
    Hal Hh;
    const long long o = hbase + (long long)(k - 1) * n2;
    Hh.u = a.UCUR[o]; Hh.v = a.VCUR[o]; Hh.rn = RN[o]; Hh.rc = a.RHOCUR[o]; Hh.ro = RO[o]; Hh.um = a.UMIX[o]; Hh.vm = a.VMIX[o];
    return Hh;
  };
  auto rho_f = [&](double rn, double rc, double ro, double bk) { return pavg ? 0.25 * (rn + 2.0 * rc + ro) * bk : rc * bk; };
  Own own = load_own(1);
  Lev cur = load_cell(1);
  Hal hal = load_halo(1);
  Own own1 = load_own(km > 1 ? 2 : 1);                       // level k + 1 of the own column
  for (int k = 1; k <= km; ++k) {
    const int buf = k & 1;
    const double bk = g.bouss[k];
    const double dzk = g.dz[k];
    {
      double pu = own.u * dyu_o, pv = own.v * dxu_o, hu = hal.u * dyu_h, hv = hal.v * dxu_h;
      if (PBC) {
        const double zo = pbc_dz(g, k, kmu_own, dzub_o), zh = pbc_dz(g, k, kmu_h, dzub_h);
        pu = pu * zo; pv = pv * zo; hu = hu * zh; hv = hv * zh;
        t.dzu[buf][lc] = zo; t.dzu[buf][hl] = zh;
      }
      // own cell first, halo cell second: a lane without halo duty has hl = lc and hbase = base3, i.e. its "halo" loads ARE the loads of
      // its own cell (same addresses, same level), so it rewrites its own cell with the same values and no select is needed (r4: the
      // 14 selects `hd ? hal.x : own.x` were 28 v_cndmask of the 368 VALU instructions of a level)
      t.u[buf][lc] = own.u; t.v[buf][lc] = own.v; t.ud[buf][lc] = pu; t.vd[buf][lc] = pv;
      t.f[buf][lc] = rho_f(cur.rn, cur.rc, cur.ro, bk); t.um[buf][lc] = cur.um; t.vm[buf][lc] = cur.vm;
      t.u[buf][hl] = hal.u; t.v[buf][hl] = hal.v; t.ud[buf][hl] = hu; t.vd[buf][hl] = hv;
      t.f[buf][hl] = rho_f(hal.rn, hal.rc, hal.ro, bk);
      t.um[buf][hl] = hal.um; t.vm[buf][hl] = hal.vm;
    }
    const int kp1 = (k < km) ? k + 1 : km, kp2 = (k + 2 <= km) ? k + 2 : km;
    const Lev nxt = load_cell(kp1);                          // in flight while this level is computed
    const Hal nhal = load_halo(kp1);
    const Own own2 = load_own(kp2);
    __syncthreads();
    {
      const double uc_k = own.u, vc_k = own.v, uo_k = own.uo, vo_k = own.vo;
      const Own &nx = own1;                                  // level k + 1 (level km again at the bottom)
#define TU(di, dj) t.u[buf][lc + (dj) * T::W + (di)]
#define TV(di, dj) t.v[buf][lc + (dj) * T::W + (di)]
#define UD(di, dj) t.ud[buf][lc + (dj) * T::W + (di)]
#define VD(di, dj) t.vd[buf][lc + (dj) * T::W + (di)]
      const double UUW = 0.25 * (UD(0, 0) + UD(-1, 0)) + 0.125 * (UD(0, -1) + UD(-1, -1) + UD(0, 1) + UD(-1, 1));
      const double UUE = 0.25 * (UD(1, 0) + UD(0, 0)) + 0.125 * (UD(1, -1) + UD(0, -1) + UD(1, 1) + UD(0, 1));
      const double VUS = 0.25 * (VD(0, 0) + VD(0, -1)) + 0.125 * (VD(-1, 0) + VD(-1, -1) + VD(1, 0) + VD(1, -1));
      const double VUN = 0.25 * (VD(0, 1) + VD(0, 0)) + 0.125 * (VD(-1, 1) + VD(-1, 0) + VD(1, 1) + VD(1, 0));
      const double u0 = TU(0, 0), un = TU(0, 1), us = TU(0, -1), ue = TU(1, 0), uw = TU(-1, 0);
      const double v0 = TV(0, 0), vn = TV(0, 1), vs = TV(0, -1), ve = TV(1, 0), vw = TV(-1, 0);
      const double dzu = PBC ? pbc_dz(g, k, kmu_own, dzub_o) : 0.0, dzu_kp1 = PBC ? pbc_dz(g, kp1, kmu_own, dzub_o) : 0.0;
      const double wukb = PBC ? wuk + (VUN - VUS + UUE - UUW) * uar : wuk + g.c2dz[k] * 0.5 * (VUN - VUS + UUE - UUW) * uar;
      const double cc = VUN - VUS + UUE - UUW;
      double LU = 0.5 * (cc * u0 + VUN * un - VUS * us + UUE * ue - UUW * uw) * uar;
      double LV = 0.5 * (cc * v0 + VUN * vn - VUS * vs + UUE * ve - UUW * vw) * uar;
      if (PBC) { LU = LU / dzu; LV = LV / dzu; }
      if (k == 1) { LU = LU + g.dzr[k] * wuk * uc_k; LV = LV + g.dzr[k] * wuk * vc_k; }
      else if (PBC) { LU = LU + 0.5 / dzu * wuk * (uc_km1 + uc_k); LV = LV + 0.5 / dzu * wuk * (vc_km1 + vc_k); }
      else { LU = LU + g.dz2r[k] * wuk * (uc_km1 + uc_k); LV = LV + g.dz2r[k] * wuk * (vc_km1 + vc_k); }
      if (k < km) {
        if (PBC) { LU = LU - 0.5 / dzu * wukb * (uc_k + nx.u); LV = LV - 0.5 / dzu * wukb * (vc_k + nx.v); }
        else { LU = LU - g.dz2r[k] * wukb * (uc_k + nx.u); LV = LV - g.dz2r[k] * wukb * (vc_k + nx.v); }
      }
      const bool wet = k <= kmu;
      LU = wet ? LU + uc_k * vc_k * kyu - vc_k * vc_k * kxu : 0.0;
      LV = wet ? LV + uc_k * vc_k * kxu - uc_k * uc_k * kyu : 0.0;
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
        const double rhokx = wet ? dxur * 0.5 * (f11 - f00 - f01 + f10) : 0.0;
        const double rhoky = wet ? dyur * 0.5 * (f11 - f00 + f01 - f10) : 0.0;
        if (k == 1) { rhokmx = rhokx; rhokmy = rhoky; sumx = 0.0; sumy = 0.0; }
        const double factor = g.dzw[k - 1] * sp.grav * 0.5;
        sumx = sumx + factor * (rhokx + rhokmx);
        sumy = sumy + factor * (rhoky + rhokmy);
        rhokmx = rhokx; rhokmy = rhoky;
        FX = FX - sumx; FY = FY - sumy;
      }
      double cn = dun, cs = dus, ce = due, cw = duw;
      if (PBC) {   // the four neighbour weights of both Laplacians scaled by min(DZU) / DZU; the central one unchanged
        const double zn = t.dzu[buf][lc + T::W], zs = t.dzu[buf][lc - T::W], ze = t.dzu[buf][lc + 1], zw = t.dzu[buf][lc - 1];
        cn = dun * fmin(zn, dzu) / dzu; cs = dus * fmin(zs, dzu) / dzu; ce = due * fmin(ze, dzu) / dzu; cw = duw * fmin(zw, dzu) / dzu;
      }
      {
        const double um0 = t.um[buf][lc], umn = t.um[buf][lc + T::W], ums = t.um[buf][lc - T::W], ume = t.um[buf][lc + 1], umw = t.um[buf][lc - 1];
        const double vm0 = t.vm[buf][lc], vmn = t.vm[buf][lc + T::W], vms = t.vm[buf][lc - T::W], vme = t.vm[buf][lc + 1], vmw = t.vm[buf][lc - 1];
        const double hdu = sp.am * ((cc_h * um0 + cn * umn + cs * ums + ce * ume + cw * umw) +
                                    (dmc * vm0 + dmn * vmn + dms * vms + dme * vme + dmw * vmw));
        const double hdv = sp.am * ((cc_h * vm0 + cn * vmn + cs * vms + ce * vme + cw * vmw) -
                                    (dmc * um0 + dmn * umn + dms * ums + dme * ume + dmw * umw));
        FX = FX + (wet ? hdu : 0.0); FY = FY + (wet ? hdv : 0.0);
      }
      {   // hdiffu_del4's first Laplacian of the current velocity, for the next step (same expression as k_del4_d2u); to the dump
          // area when it is not formed
        double du = (cc_h * u0 + cn * un + cs * us + ce * ue + cw * uw) + (dmc * v0 + dmn * vn + dms * vs + dme * ve + dmw * vw);
        double dv = (cc_h * v0 + cn * vn + cs * vs + ce * ve + cw * vw) - (dmc * u0 + dmn * un + dms * us + dme * ue + dmw * uw);
        du = wet ? amf_next * du : 0.0; dv = wet ? amf_next * dv : 0.0;
        D0p[(long long)(k - 1) * (d2n ? ostep : 0)] = du; D1p[(long long)(k - 1) * (d2n ? ostep : 0)] = dv;
      }
      {
        const double vvc = cur.vvc;
        double vufb = vvc * (uo_k - nx.uo) * g.dzwr[k];
        double vvfb = vvc * (vo_k - nx.vo) * g.dzwr[k];
        if (PBC) {
          const double Wd = (k < km) ? 0.5 * (dzu + dzu_kp1) : 0.5 * dzu_kp1;
          vufb = vvc * (uo_k - nx.uo) / Wd; vvfb = vvc * (vo_k - nx.vo) / Wd;
        }
        if (k == kmu) {
          const double vmag = sp.bottom_drag * sqrt(uo_k * uo_k + vo_k * vo_k);
          vufb = vmag * uo_k; vvfb = vmag * vo_k;
        }
        double vdu, vdv;
        if (PBC) { vdu = wet ? (vuf - vufb) / dzu : 0.0; vdv = wet ? (vvf - vvfb) / dzu : 0.0; }
        else { vdu = wet ? (vuf - vufb) * g.dzr[k] : 0.0; vdv = wet ? (vvf - vvfb) * g.dzr[k] : 0.0; }
        vuf = vufb; vvf = vvfb;
        FX = FX + vdu; FY = FY + vdv;
      }
      if (!wet) { FX = 0.0; FY = 0.0; }
      double xu, xv;
      if (sp.impcor) {
        const double W1 = sp.c2dtu * sp.beta * fcor;
        const double W2 = sp.c2dtu / (1.0 + W1 * W1);
        xu = (FX + W1 * FY) * W2;
        xv = (FY - W1 * FX) * W2;
      } else { xu = sp.c2dtu * FX; xv = sp.c2dtu * FY; }
      UNp[(long long)(k - 1) * ostep] = xu; VNp[(long long)(k - 1) * ostep] = xv;
      if (PBC) { zx = zx + FX * dzu; zy = zy + FY * dzu; }
      else { zx = zx + FX * dzk; zy = zy + FY * dzk; }
      wuk = wukb;
      uc_km1 = uc_k; vc_km1 = vc_k;
#undef TU
#undef TV
#undef UD
#undef VD
    }
    cur = nxt; hal = nhal; own = own1; own1 = own2;
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
  if (g.pbc) hipLaunchKernelGGL((k_momentum_rhs_lds<R, true>), G, B, 0, st, g, sp, a, tj_first, tj_count);
  else hipLaunchKernelGGL((k_momentum_rhs_lds<R, false>), G, B, 0, st, g, sp, a, tj_first, tj_count);
}

}  // namespace pop
