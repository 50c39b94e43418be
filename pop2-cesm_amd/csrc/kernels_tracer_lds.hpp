// kernels_tracer_lds.hpp -- tracer right-hand side (centred advection) with the horizontal stencils staged
// through LDS.  Same arithmetic, evaluation order and results as k_tracer_rhs<false,false>
// (kernels_baroclinic.hpp; tracer_update, baroclinic.F90:1981-2300); only the source of the neighbours
// changes.  A workgroup owns a 64 x R tile of columns and marches k; per level every thread loads ITS cell of
// U, V, T, S (curtime) and the two mixing-time tracers plus one halo cell into a double-buffered LDS tile
// and reads the 5-point / corner neighbours from there, with the next level's cells already in flight.
// At tx0.1v3 the direct-load kernel moves 104 GB through the fabric for 43 GB of algorithmic reads and runs at
// the fabric ceiling; the tile cuts the re-fetch to the halo overhead.
#pragma once
#include "kernels_baroclinic.hpp"

namespace pop {

template <int R>
struct TrcTile {
  static constexpr int W = POP_COL_THREADS + 2, H = R + 2, N = W * H, NHALO = N - POP_COL_THREADS * R;
  double u[2][N], v[2][N], tc[2][2][N], tm[2][2][N];
  static_assert(NHALO <= POP_COL_THREADS * R, "every halo cell needs a thread");
};

// FWD: the forward elimination of impvmixt (vertical_mix.F90:1263-1340; pressure-averaging predictor, PSFC = PSURF(cur))
// runs on the right-hand side as it is formed -- level k's value and VDC(k) are in registers anyway -- and E, F are
// stored instead of the right-hand side: the separate solve then only substitutes back (k_impvmixt_back).  Three field
// passes per tracer less than right-hand side + k_impvmixt; the same operations in the same order (bitwise equal, tested).
template <int R, bool FWD = false>
__global__ void __launch_bounds__(POP_COL_THREADS * R, POP_TRC_WAVES)
k_tracer_rhs_lds(DevGrid g, StepParams sp, TracerRhsArgs a) {
  using T = TrcTile<R>;
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
  const bool act = inb && i + 1 <= g.ie && j + 1 <= g.je;
  const int p2 = inb ? j * nxb + i : 0;
  const long long q2 = (long long)b * n2 + p2, base3 = (long long)b * g.n3 + p2;
  const int lc = (ty + 1) * T::W + tx + 1;
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
  int kmt = 0, kmtn = 0, kmts = 0, kmte = 0, kmtw = 0;
  double dtn = 0, dts = 0, dte = 0, dtw = 0, dyu00 = 0, dyu0m = 0, dyum0 = 0, dyumm = 0, dxu00 = 0, dxu0m = 0, dxum0 = 0, dxumm = 0;
  double tarear = 0, psfac = 0, wtk = 0, stf[2] = {0, 0}, tfw[2] = {0, 0};
  double sw_q = 0.0, sw_tkm1 = 1.0;
  int sw_chli = 0;
  if (act) {
    kmt = g.KMT[q2]; kmtn = g.KMTN[q2]; kmts = g.KMTS[q2]; kmte = g.KMTE[q2]; kmtw = g.KMTW[q2];
    dtn = g.DTN[q2]; dts = g.DTS[q2]; dte = g.DTE[q2]; dtw = g.DTW[q2];
    dyu00 = g.DYU[q2]; dyu0m = g.DYU[q2 - nxb]; dyum0 = g.DYU[q2 - 1]; dyumm = g.DYU[q2 - 1 - nxb];
    dxu00 = g.DXU[q2]; dxu0m = g.DXU[q2 - nxb]; dxum0 = g.DXU[q2 - 1]; dxumm = g.DXU[q2 - 1 - nxb];
    tarear = g.TAREA_R[q2];
    psfac = (a.PCUR[q2] - a.POLD[q2]);
    wtk = a.DH[q2];
#pragma unroll
    for (int n = 0; n < 2; ++n) { stf[n] = a.STF[n][q2]; tfw[n] = a.TFW[n][q2]; }
    if (a.sw_on) { sw_q = fmax(a.QSW[q2], 0.0); if (a.sw_type == 2) sw_chli = a.swCHLI[q2]; }
  }
  // KPP's non-local source is +-0 below level KBL (blmix: ghat = 0 there, so the flux difference is 0 - 0), and the sum below starts
  // from +0.0, so a level that is not read adds the same +0.0: rows of a level in which every column is past its KBL are not fetched
  const int ksrc = (act && a.use_kpp_src) ? (a.KBL ? a.KBL[q2] : km) : 0;
  const double ahf_next = (act && a.D2N[0]) ? a.AHF[q2] : 0.0;
  const long long vdcbase = ((long long)b * (km + 2)) * n2 + p2;
  struct Lev { double u, v, tc[2], tm[2], to[2], vdc[2], src[2]; };
  struct Hal { double u, v, tc[2], tm[2]; };
  auto load_cell = [&](int k) {
    Lev L{};
    if (inb) {
      const long long o = base3 + (long long)(k - 1) * n2;
      L.u = a.UCUR[o]; L.v = a.VCUR[o];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        L.tc[n] = a.TCUR[n][o]; L.tm[n] = a.TMIX[n][o]; L.to[n] = a.TOLD[n][o];
        L.vdc[n] = a.VDC[n][vdcbase + (long long)k * n2];
        L.src[n] = (k <= ksrc) ? a.KPP_SRC[n][o] : 0.0;
      }
    }
    return L;
  };
  auto load_halo = [&](int k) {
    Hal Hh{};
    if (hin) {
      const long long o = hbase + (long long)(k - 1) * n2;
      Hh.u = a.UCUR[o]; Hh.v = a.VCUR[o];
#pragma unroll
      for (int n = 0; n < 2; ++n) { Hh.tc[n] = a.TCUR[n][o]; Hh.tm[n] = a.TMIX[n][o]; }
    }
    return Hh;
  };
  Lev cur = load_cell(1);
  Hal hal = load_halo(1);
  double vtf[2] = {0, 0}, tc_km1[2] = {0.0, 0.0};
  double *__restrict__ const TNp[2] = {a.TNEW[0], a.TNEW[1]};
  // forward-elimination state per tracer (FWD)
  const double hfac1 = g.dz[1] / a.c2dtt;
  const double H1 = hfac1 + (act ? a.PCUR[q2] : 0.0) / (sp.grav * a.c2dtt);
  double fwA[2] = {0, 0}, fwB[2] = {0, 0}, fwF[2] = {0, 0};
  for (int k = 1; k <= km; ++k) {
    const int buf = k & 1;
    t.u[buf][lc] = cur.u; t.v[buf][lc] = cur.v;
#pragma unroll
    for (int n = 0; n < 2; ++n) { t.tc[buf][n][lc] = cur.tc[n]; t.tm[buf][n][lc] = cur.tm[n]; }
    if (hl >= 0) {
      t.u[buf][hl] = hal.u; t.v[buf][hl] = hal.v;
#pragma unroll
      for (int n = 0; n < 2; ++n) { t.tc[buf][n][hl] = hal.tc[n]; t.tm[buf][n][hl] = hal.tm[n]; }
    }
    const int kp1 = (k < km) ? k + 1 : km;
    const Lev nxt = load_cell(kp1);
    const Hal nhal = load_halo(kp1);
    __syncthreads();
    if (act) {
      const long long o = base3 + (long long)(k - 1) * n2;
      const double u00 = t.u[buf][lc], u0m = t.u[buf][lc - T::W], um0 = t.u[buf][lc - 1], umm = t.u[buf][lc - 1 - T::W];
      const double v00 = t.v[buf][lc], v0m = t.v[buf][lc - T::W], vm0 = t.v[buf][lc - 1], vmm = t.v[buf][lc - 1 - T::W];
      const double UTE = 0.5 * (u00 * dyu00 + u0m * dyu0m);
      const double UTW = 0.5 * (um0 * dyum0 + umm * dyumm);
      const double VTN = 0.5 * (v00 * dxu00 + vm0 * dxum0);
      const double VTS = 0.5 * (v0m * dxu0m + vmm * dxumm);
      const double hdiv = VTN - VTS + UTE - UTW;
      double wtkb = 0.0;
      if (k < km) { const double FC = hdiv * tarear; wtkb = (k < kmt) ? wtk + g.dz[k] * FC : 0.0; }
      const double CN = (k <= kmtn && k <= kmt) ? dtn : 0.0, CS = (k <= kmts && k <= kmt) ? dts : 0.0;
      const double CE = (k <= kmte && k <= kmt) ? dte : 0.0, CW = (k <= kmtw && k <= kmt) ? dtw : 0.0;
      const double CC = -(CN + CS + CE + CW);
      const double dz2rk = g.dz2r[k], dzrk = g.dzr[k], dzwrk = g.dzwr[k];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const double *TM = t.tm[buf][n], *TC = t.tc[buf][n];
        const double tc_k = cur.tc[n], tc_kp1 = nxt.tc[n], to_k = cur.to[n], to_kp1 = nxt.to[n];
        double FT = sp.ah * (CC * TM[lc] + CN * TM[lc + T::W] + CS * TM[lc - T::W] + CE * TM[lc + 1] + CW * TM[lc - 1]);
        // hdifft_del4's first Laplacian of the current tracers, for the next step (same expression as k_del4_d2t)
        if (a.D2N[0]) a.D2N[n][o] = ahf_next * (CC * TC[lc] + CN * TC[lc + T::W] + CS * TC[lc - T::W] + CE * TC[lc + 1] + CW * TC[lc - 1]);
        double L = 0.5 * (hdiv * tc_k + VTN * TC[lc + T::W] - VTS * TC[lc - T::W] + UTE * TC[lc + 1] - UTW * TC[lc - 1]) * tarear;
        if (k != 1) L = L + dz2rk * wtk * (tc_km1[n] + tc_k);
        if (k < km) L = L - dz2rk * wtkb * (tc_k + tc_kp1);
        FT = FT - L;
        if (k == 1) vtf[n] = (kmt >= 1) ? stf[n] : 0.0;
        const double vtfb = (kmt > k) ? cur.vdc[n] * (to_k - to_kp1) * dzwrk : 0.0;
        const double vd = (k <= kmt) ? (vtf[n] - vtfb) * dzrk : 0.0;
        vtf[n] = vtfb;
        FT = FT + vd;
        if (k == 1) FT = FT + g.dzr[1] * tfw[n];
        double src = 0.0;
        if (a.use_kpp_src) src = src + cur.src[n];
        if (a.sw_on && n == 0) src = src + sw_source(a, sw_q, k, kmt, dzrk, sw_chli, sw_tkm1);
        FT = FT + src;
        if (!FWD) {
          if (k == 1 && sp.pavg) {
            if (kmt > 0) TNp[n][o] = a.c2dtt * FT - 2.0 * tc_k * psfac / (sp.grav * g.dz[1]);
          } else {
            TNp[n][o] = (k <= kmt) ? a.c2dtt * FT : 0.0;
          }
        } else {
          // the value the right-hand side would have stored (land columns keep what TNEW(1) held: the reference's don't-care)
          double rhs;
          if (k == 1 && sp.pavg) rhs = (kmt > 0) ? a.c2dtt * FT - 2.0 * tc_k * psfac / (sp.grav * g.dz[1]) : TNp[n][o];
          else rhs = (k <= kmt) ? a.c2dtt * FT : 0.0;
          if (k == 1) {
            const double A = g.afac_t[1] * cur.vdc[n];
            const double D = H1 + A;
            const double Ek = A / D;
            fwA[n] = A; fwB[n] = H1 * Ek; fwF[n] = hfac1 * rhs / D;
            a.E[n][o] = Ek; a.F[n][o] = fwF[n];
          } else {
            const double C = fwA[n];
            const double hf = g.dz[k] / a.c2dtt;
            const double A = g.afac_t[k] * cur.vdc[n];
            fwA[n] = A;
            if (k > kmt) fwF[n] = 0.0;
            else {
              const double D = (k == kmt) ? hf + fwB[n] : hf + A + fwB[n];
              const double Ek = A / D;
              fwB[n] = (hf + fwB[n]) * Ek;
              fwF[n] = (hf * rhs + C * fwF[n]) / D;
              a.E[n][o] = Ek;
            }
            a.F[n][o] = fwF[n];
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
  if (fwd) hipLaunchKernelGGL((k_tracer_rhs_lds<R, true>), G, B, 0, st, g, sp, a);
  else hipLaunchKernelGGL((k_tracer_rhs_lds<R, false>), G, B, 0, st, g, sp, a);
}

}  // namespace pop
