// kernels_tracer_lds.hpp -- tracer right-hand side (centred advection) with the horizontal stencils staged
// through LDS.  Same arithmetic, evaluation order and results as k_tracer_rhs<false,false> (kernels_baroclinic.hpp;
// tracer_update, baroclinic.F90:1981-2300); only the source of the neighbours and the schedule of the level loop
// change.  A workgroup owns a 64 x R tile of columns and marches k; per level every thread loads ITS cell of
// U, V, T, S (curtime) and the two mixing-time tracers plus one halo cell into a double-buffered LDS tile
// and reads the 5-point / corner neighbours from there, with the next level's cells already in flight.
//
// Round 3, as in kernels_momentum_lds.hpp: the face transports are staged as the PRODUCTS U*DYU(*DZU), V*DXU(*DZU)
// formed by the owner of the U cell (eight neighbour constants and eight multiplies per column and level gone;
// with partial bottom cells the owner's thickness goes into the same product), the level body is branch-free (clamped
// loads, dump stores) so that the compiler can count its waits instead of draining the prefetch with vmcnt(0), and the
// vertical constants are scalar loads (DevGrid::CArr).
#pragma once
#include "kernels_baroclinic.hpp"

namespace pop {

template <int R, bool PBC>
struct TrcTile {
  static constexpr int W = POP_COL_THREADS + 2, H = R + 2, N = W * H, NHALO = N - POP_COL_THREADS * R;
  double ud[2][N], vd[2][N], tc[2][2][N], tm[2][2][N];
  double dzt[PBC ? 2 : 1][PBC ? N : 1];      // partial bottom cells: thickness of the T cell at the level
  static_assert(NHALO <= POP_COL_THREADS * R, "every halo cell needs a thread");
};

// FWD: the forward elimination of impvmixt (vertical_mix.F90:1263-1340; pressure-averaging predictor, PSFC = PSURF(cur))
// runs on the right-hand side as it is formed -- level k's value and VDC(k) are in registers anyway -- and E, F are
// stored instead of the right-hand side: the separate solve then only substitutes back (k_impvmixt_back).  Three field
// passes per tracer less than right-hand side + k_impvmixt; the same operations in the same order (bitwise equal, tested).
// PBC: partial bottom cells (advection.F90:2040-2062, 2110, 2223-2294; hmix_del2.F90:1034-1051 / hmix_del4.F90:964-984;
// vertical_mix.F90:790-807, 1279-1287; sw_absorption.F90:880-921).
// HDIN (round 4): the horizontal mixing tendency of both tracers is given (TracerRhsArgs::HDT: Gent-McWilliams, formed by phase_hmix_gm) and read in
// place of the del2 operator on the staged mix-time tracers, which are then neither loaded nor staged -- k_tracer_rhs<DEL4 = true>'s `FT = HDT`
template <int R, bool FWD = false, bool PBC = false, bool HDIN = false>
__global__ void __launch_bounds__(POP_COL_THREADS * R, POP_TRC_WAVES)
k_tracer_rhs_lds(DevGrid g, StepParams sp, TracerRhsArgs a) {
  using T = TrcTile<R, PBC>;
  __shared__ T t;
  const int nxb = g.nxb, nyb = g.nyb, km = g.km;
  const long long n2 = g.n2;
  const int tiles_i = (nxb - 2 * NGHOST + POP_COL_THREADS - 1) / POP_COL_THREADS;
  const int tiles_j = (nyb - 2 * NGHOST + R - 1) / R, b = blockIdx.y;
  int ti, tj;
  bool listed = false;
  if (!lds_tile_active<R>(g, b, tiles_i, tiles_j, ti, tj, listed)) return;
  const int i0 = NGHOST + ti * POP_COL_THREADS, j0 = NGHOST + tj * R;
  if (!listed && land_tile(g, b, i0, POP_COL_THREADS, j0, R)) return;   // no ocean column in the tile
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * POP_COL_THREADS + tx;
  const int i = i0 + tx, j = j0 + ty;
  const bool inb = i < nxb && j < nyb;
  const bool act = inb && i + 1 <= blk_ie(g, b) && j + 1 <= blk_je(g, b);
  // every lane addresses a cell that exists (lanes beyond the array: cell 0, whose values are never used) and stores either to
  // the field or to the dump area
  const int p2 = inb ? j * nxb + i : 0;
  const long long q2 = (long long)b * n2 + p2, base3 = (long long)b * g.n3 + p2;
  const int lc = (ty + 1) * T::W + tx + 1;
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
  const int kmt_own = g.KMT[q2];
  const int kmt = act ? kmt_own : 0;
  const int kmtn = g.KMTN[q2], kmts = g.KMTS[q2], kmte = g.KMTE[q2], kmtw = g.KMTW[q2];
  const double dtn = g.DTN[q2], dts = g.DTS[q2], dte = g.DTE[q2], dtw = g.DTW[q2];
  const double dyu_o = g.DYU[q2], dxu_o = g.DXU[q2], dyu_h = g.DYU[hq2], dxu_h = g.DXU[hq2];
  const double tarear = g.TAREA_R[q2];
  const double pcur = a.PCUR[q2];
  const double psfac = (pcur - a.POLD[q2]);
  double wtk = a.DH[q2];
  double stf[2], tfw[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) { stf[n] = a.STF[n][q2]; tfw[n] = a.TFW[n][q2]; }
  // partial bottom cells: bottom level / thickness of the own and the halo U cell (transports) and T cell (mixing weights)
  const int kmu_o = PBC ? g.KMU[q2] : 0, kmu_h = PBC ? g.KMU[hq2] : 0, kmt_h = PBC ? g.KMT[hq2] : 0;
  const double dzub_o = PBC ? g.DZUB[q2] : 0.0, dzub_h = PBC ? g.DZUB[hq2] : 0.0, dzbc_o = PBC ? g.DZBC[q2] : 0.0, dzbc_h = PBC ? g.DZBC[hq2] : 0.0;
  double sw_q = 0.0, sw_tkm1 = 1.0;
  int sw_chli = 0;
  if (a.sw_on) { sw_q = fmax(a.QSW[q2], 0.0); if (a.sw_type == 2) sw_chli = a.swCHLI[q2]; }
  // KPP's non-local source is +-0 below level KBL (blmix: ghat = 0 there, so the flux difference is 0 - 0), and the sum below starts
  // from +0.0: a level below KBL reads a zero word instead of the field (one cache line for the whole launch)
  const int kbl_own = a.KBL ? a.KBL[q2] : km;
  const int ksrc = a.use_kpp_src ? kbl_own : 0;
  const bool d2n = a.D2N[0] != nullptr;
  const double ahf_next = (d2n ? a.AHF : g.TAREA_R)[q2];     // (unused without D2N; the load stays unconditional)
  const long long vdcbase = ((long long)b * (km + 2)) * n2 + p2;
  const double *const zero = g.zero;
  const double *__restrict__ const SRC0 = a.use_kpp_src ? a.KPP_SRC[0] : zero - base3;
  const double *__restrict__ const SRC1 = a.use_kpp_src ? a.KPP_SRC[1] : zero - base3;
  struct Lev { double u, v, tc[2], tm[2], to[2], vdc[2], src[2]; };
  struct Hal { double u, v, tc[2], tm[2]; };
  auto load_cell = [&](int k) {
    Lev L;
    const long long o = base3 + (long long)(k - 1) * n2;
    L.u = a.UCUR[o]; L.v = a.VCUR[o];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      L.tc[n] = a.TCUR[n][o]; L.tm[n] = (HDIN ? a.HDT[n] : a.TMIX[n])[o]; L.to[n] = a.TOLD[n][o];
      L.vdc[n] = a.VDC[n][vdcbase + (long long)k * n2];
    }
    const bool rd = k <= ksrc;
    L.src[0] = *(rd ? SRC0 + o : zero); L.src[1] = *(rd ? SRC1 + o : zero);
    return L;
  };
  auto load_halo = [&](int k) {
    Hal Hh;
    const long long o = hbase + (long long)(k - 1) * n2;
    Hh.u = a.UCUR[o]; Hh.v = a.VCUR[o];
#pragma unroll
    for (int n = 0; n < 2; ++n) { Hh.tc[n] = a.TCUR[n][o]; Hh.tm[n] = HDIN ? 0.0 : a.TMIX[n][o]; }
    return Hh;
  };
  // outputs: the field for physical columns, the dump area for the other lanes
  double *const dump = g.dump + tid;
  double *__restrict__ const TN0 = act ? a.TNEW[0] + base3 : dump;
  double *__restrict__ const TN1 = act ? a.TNEW[1] + base3 : dump + 512;
  double *__restrict__ const D0p = (act && d2n) ? a.D2N[0] + base3 : dump + 1024;
  double *__restrict__ const D1p = (act && d2n) ? a.D2N[1] + base3 : dump + 1536;
  double *__restrict__ const E0p = (FWD && act) ? a.E[0] + base3 : dump + 2048;
  double *__restrict__ const E1p = (FWD && act) ? a.E[1] + base3 : dump + 2560;
  double *__restrict__ const F0p = (FWD && act) ? a.F[0] + base3 : dump + 3072;
  double *__restrict__ const F1p = (FWD && act) ? a.F[1] + base3 : dump + 3584;
  const long long ostep = act ? n2 : 0;
  // FWD, land columns at k = 1 with pressure averaging: the reference leaves TNEW(1) as it was (a don't-care the elimination reads)
  double tn1_old[2] = {0.0, 0.0};
  if (FWD) { tn1_old[0] = TN0[0]; tn1_old[1] = TN1[0]; }
  Lev cur = load_cell(1);
  Hal hal = load_halo(1);
  double vtf[2] = {0, 0}, tc_km1[2] = {0.0, 0.0};
  // forward-elimination state per tracer (FWD)
  const double hfac1 = g.dz[1] / a.c2dtt;
  const double H1 = hfac1 + pcur / (sp.grav * a.c2dtt);
  double fwA[2] = {0, 0}, fwB[2] = {0, 0}, fwF[2] = {0, 0};
  for (int k = 1; k <= km; ++k) {
    const int buf = k & 1;
    {
      double pu = cur.u * dyu_o, pv = cur.v * dxu_o, hu = hal.u * dyu_h, hv = hal.v * dxu_h;
      if (PBC) {
        const double zo = pbc_dz(g, k, kmu_o, dzub_o), zh = pbc_dz(g, k, kmu_h, dzub_h);
        pu = pu * zo; pv = pv * zo; hu = hu * zh; hv = hv * zh;
        const double to_ = pbc_dz(g, k, kmt_own, dzbc_o), th_ = pbc_dz(g, k, kmt_h, dzbc_h);
        t.dzt[buf][lc] = to_; t.dzt[buf][hl] = th_;
      }
      t.ud[buf][lc] = pu; t.vd[buf][lc] = pv;
      // (a lane without halo duty has hl = lc and hbase = base3: its "halo" loads are the loads of its own cell, the same values -- no select)
      t.ud[buf][hl] = hu; t.vd[buf][hl] = hv;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        t.tc[buf][n][lc] = cur.tc[n]; t.tc[buf][n][hl] = hal.tc[n];
        if (!HDIN) { t.tm[buf][n][lc] = cur.tm[n]; t.tm[buf][n][hl] = hal.tm[n]; }
      }
    }
    const int kp1 = (k < km) ? k + 1 : km;
    const Lev nxt = load_cell(kp1);
    const Hal nhal = load_halo(kp1);
    __syncthreads();
    {
      const long long oo = (long long)(k - 1) * ostep;
      const double ud00 = t.ud[buf][lc], ud0m = t.ud[buf][lc - T::W], udm0 = t.ud[buf][lc - 1], udmm = t.ud[buf][lc - 1 - T::W];
      const double vd00 = t.vd[buf][lc], vd0m = t.vd[buf][lc - T::W], vdm0 = t.vd[buf][lc - 1], vdmm = t.vd[buf][lc - 1 - T::W];
      const double UTE = 0.5 * (ud00 + ud0m);
      const double UTW = 0.5 * (udm0 + udmm);
      const double VTN = 0.5 * (vd00 + vdm0);
      const double VTS = 0.5 * (vd0m + vdmm);
      const double hdiv = VTN - VTS + UTE - UTW;
      const double dzt = PBC ? pbc_dz(g, k, kmt_own, dzbc_o) : 0.0, dzt_kp1 = PBC ? pbc_dz(g, kp1, kmt_own, dzbc_o) : 0.0;
      double wtkb = 0.0;
      if (k < km) { const double FC = hdiv * tarear; wtkb = (k < kmt) ? (PBC ? wtk + FC : wtk + g.dz[k] * FC) : 0.0; }
      double CN = dtn, CS = dts, CE = dte, CW = dtw;
      if (PBC) {
        CN = dtn * fmin(dzt, t.dzt[buf][lc + T::W]) / dzt; CS = dts * fmin(dzt, t.dzt[buf][lc - T::W]) / dzt;
        CE = dte * fmin(dzt, t.dzt[buf][lc + 1]) / dzt; CW = dtw * fmin(dzt, t.dzt[buf][lc - 1]) / dzt;
      }
      if (!(k <= kmtn && k <= kmt)) CN = 0.0;
      if (!(k <= kmts && k <= kmt)) CS = 0.0;
      if (!(k <= kmte && k <= kmt)) CE = 0.0;
      if (!(k <= kmtw && k <= kmt)) CW = 0.0;
      const double CC = -(CN + CS + CE + CW);
      const double dz2rk = g.dz2r[k], dzrk = g.dzr[k], dzwrk = g.dzwr[k];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const double *TM = t.tm[buf][n], *TC = t.tc[buf][n];
        const double tc_k = cur.tc[n], tc_kp1 = nxt.tc[n], to_k = cur.to[n], to_kp1 = nxt.to[n];
        double FT = HDIN ? cur.tm[n] : sp.ah * (CC * TM[lc] + CN * TM[lc + T::W] + CS * TM[lc - T::W] + CE * TM[lc + 1] + CW * TM[lc - 1]);
        // hdifft_del4's first Laplacian of the current tracers, for the next step (same expression as k_del4_d2t); to the dump area
        // when it is not formed
        (n == 0 ? D0p : D1p)[d2n ? oo : 0] = ahf_next * (CC * TC[lc] + CN * TC[lc + T::W] + CS * TC[lc - T::W] + CE * TC[lc + 1] + CW * TC[lc - 1]);
        double L = 0.5 * (hdiv * tc_k + VTN * TC[lc + T::W] - VTS * TC[lc - T::W] + UTE * TC[lc + 1] - UTW * TC[lc - 1]) * tarear;
        if (PBC) {
          L = L / dzt;
          if (k != 1) L = L + 0.5 / dzt * wtk * (tc_km1[n] + tc_k);
          if (k < km) L = L - 0.5 / dzt * wtkb * (tc_k + tc_kp1);
        } else {
          if (k != 1) L = L + dz2rk * wtk * (tc_km1[n] + tc_k);
          if (k < km) L = L - dz2rk * wtkb * (tc_k + tc_kp1);
        }
        FT = FT - L;
        if (k == 1) vtf[n] = (kmt >= 1) ? stf[n] : 0.0;
        double vtfb, vd;
        if (PBC) {
          vtfb = (kmt > k) ? cur.vdc[n] * (to_k - to_kp1) / (0.5 * (dzt + dzt_kp1)) : 0.0;
          vd = (k <= kmt) ? (vtf[n] - vtfb) / dzt : 0.0;
        } else {
          vtfb = (kmt > k) ? cur.vdc[n] * (to_k - to_kp1) * dzwrk : 0.0;
          vd = (k <= kmt) ? (vtf[n] - vtfb) * dzrk : 0.0;
        }
        vtf[n] = vtfb;
        FT = FT + vd;
        if (k == 1) FT = FT + g.dzr[1] * tfw[n];
        double src = 0.0;
        if (a.use_kpp_src) src = src + cur.src[n];
        if (a.sw_on && n == 0) src = src + sw_source(a, sw_q, k, kmt, dzrk, sw_chli, sw_tkm1, PBC ? dzt : 0.0);
        FT = FT + src;
        // the value the right-hand side stores (land columns at k = 1 with pressure averaging keep what TNEW(1) held)
        double rhs;
        bool keep = false;
        if (k == 1 && sp.pavg) { keep = !(kmt > 0); rhs = a.c2dtt * FT - 2.0 * tc_k * psfac / (sp.grav * g.dz[1]); }
        else rhs = (k <= kmt) ? a.c2dtt * FT : 0.0;
        if (!FWD) {
          double *const dst = (n == 0 ? TN0 : TN1) + oo;
          *(keep ? dump + 4096 : dst) = rhs;
        } else {
          if (keep) rhs = tn1_old[n];
          double *const Ep = (n == 0 ? E0p : E1p), *const Fp = (n == 0 ? F0p : F1p);
          if (k == 1) {
            const double A = g.afac_t[1] * cur.vdc[n];
            const double D = H1 + A;
            const double Ek = A / D;
            fwA[n] = A; fwB[n] = H1 * Ek; fwF[n] = hfac1 * rhs / D;
            Ep[oo] = Ek; Fp[oo] = fwF[n];
          } else {
            const double C = fwA[n];
            double hf = g.dz[k] / a.c2dtt;
            double A = g.afac_t[k] * cur.vdc[n];
            if (PBC) { A = sp.aidif * cur.vdc[n] / (0.5 * (dzt + pbc_dz(g, k + 1, kmt_own, dzbc_o))); hf = dzt / a.c2dtt; }
            fwA[n] = A;
            const bool below = k > kmt;
            const double D = (k == kmt) ? hf + fwB[n] : hf + A + fwB[n];
            const double Ek = A / D;
            if (!below) { fwB[n] = (hf + fwB[n]) * Ek; fwF[n] = (hf * rhs + C * fwF[n]) / D; }
            else fwF[n] = 0.0;
            *(below ? dump + 4096 : Ep + oo) = Ek;
            Fp[oo] = fwF[n];
          }
        }
        tc_km1[n] = tc_k;
      }
      wtk = wtkb;
    }
    cur = nxt; hal = nhal;
  }
}

template <int R>
inline void launch_tracer_lds(const DevGrid &g, const StepParams &sp, const TracerRhsArgs &a, hipStream_t st, bool fwd = false) {
  const int tiles_i = (g.nxb - 2 * NGHOST + POP_COL_THREADS - 1) / POP_COL_THREADS;
  const int tiles_j = (g.nyb - 2 * NGHOST + R - 1) / R;
  const dim3 G(lds_launch_x<R>(g, tiles_i, tiles_j), g.nblocks), B(POP_COL_THREADS, R);
  if (g.pbc) {
    if (fwd) hipLaunchKernelGGL((k_tracer_rhs_lds<R, true, true>), G, B, 0, st, g, sp, a);
    else hipLaunchKernelGGL((k_tracer_rhs_lds<R, false, true>), G, B, 0, st, g, sp, a);
  } else if (a.HDT[0]) {   // the mixing tendency given (Gent-McWilliams; not with partial bottom cells: refused at create)
    if (fwd) hipLaunchKernelGGL((k_tracer_rhs_lds<R, true, false, true>), G, B, 0, st, g, sp, a);
    else hipLaunchKernelGGL((k_tracer_rhs_lds<R, false, false, true>), G, B, 0, st, g, sp, a);
  } else if (fwd) hipLaunchKernelGGL((k_tracer_rhs_lds<R, true, false>), G, B, 0, st, g, sp, a);
  else hipLaunchKernelGGL((k_tracer_rhs_lds<R, false, false>), G, B, 0, st, g, sp, a);
}

}  // namespace pop
