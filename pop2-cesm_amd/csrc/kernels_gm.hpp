// kernels_gm.hpp -- Gent-McWilliams eddy transport + isopycnal (Redi) diffusion of tracers, hmix_tracer = 3
// (source/hmix_gm.F90:1102-2226 hdifft_gm, hmix_gm_submeso_share.F90:149-432 tracer_diffs_and_isopyc_slopes, init_gm :283-1095) in the
// set-up the code's own hmix_gm_nml defaults give: constant kappa (KAPPA_VERTICAL = 1), kappa_freq 'never', no transition layer,
// use_const_ah_bkg_srfbl, ah_bkg_bottom = 0, slope control 'notanh' | 'tanh'; ah_bolus /= ah or slm_b /= slm_r take the branch
// without cancellation of the skew-flux terms.  kappa type 'bfre' (buoyancy_frequency_dependent_profile :3011-3180; kappa_*_deep = 0.1,
// kappa_freq 'never' | 'every_time_step'): k_gm_kappa_vertical, a column march.  Transition layer: see k_gm_coeffs.  Bolus velocity (diag_gm_bolus): k_gm_bolus.
//
// The reference works level by level and carries whole-block work arrays (TX, TY, TZ, RX, RY, SF_SLX, SF_SLY, FZTOP) from level to
// level; here every value is a function of the mix-time tracers at the cell and its neighbours one level up / down, so two
// 3-D-parallel launches form the same numbers:
//   k_gm_coeffs   per (i,j,k): the slopes SLX, SLY of the four quarter cells of each half (top / bottom) of the T cell, the tapering
//                 factors, and the tapered KAPPA_ISOP, KAPPA_THIC, HOR_DIFF of both halves (14 stored 3-D fields);
//   k_gm_flux     the isopycnal part of the vertical diffusivity added to VDC(k) (:1725-1748) at every cell of the block, and per
//                 physical cell, both tracers together: the fluxes through its six faces (east / north of the cell and of its west / south
//                 neighbour recomputed, the flux through the top face = the bottom-face flux of the level above) -> GTK, which the
//                 tracer right-hand side reads in place of forming del2 mixing (k_tracer_rhs<DEL4 = true>).
// Correctness first (the production grids of this scheme are the 1-degree ones); same operations in the reference's order per value.
#pragma once
#include "kernels_common.hpp"

namespace pop {

struct GmDev {
  double *SLX[4], *SLY[4];          // [face * 2 + half]: face 0 east / north, 1 west / south; half 0 top (ktp), 1 bottom (kbt)
  double *KI[2], *KT[2], *HD[2];    // KAPPA_ISOP, KAPPA_THIC, HOR_DIFF of the two halves
  double *GTK[2];
  double *KV;                       // KAPPA_VERTICAL (kappa type 'bfre'; module state: 1 until computed), nullptr with constant kappa
  // transition layer (transition_layer_on; hmix_gm.F90:3183-3848): nullptr / 0 when off
  int tlt;
  double *SLA[2];                   // SLA_SAVE of the two halves (3-D)
  double *DD, *TH, *ID;             // TLT%DIABATIC_DEPTH, THICKNESS, INTERIOR_DEPTH (2-D)
  int *KL, *ZTW;                    // TLT%K_LEVEL, ZTW
  double *MW[8];                    // merged_streamfunction: WORK1, WORK2 (x) and WORK3, WORK4 (y) of the east|north and west|south side: [2 * w + face], w = 0..3
  const double *HMXL;
  // branch without cancellation: SF_SLX / SF_SLY of every half cell stored by k_gm_sf (as the reference stores them) instead of being
  // re-derived at each of the ~24 places of the flux kernel that read one; [4 * xy + 2 * face + half]; nullptr: formed in place
  double *SF[8];
  double *UISOP, *VISOP, *WISOP;   // diag_gm_bolus: U_ISOP, V_ISOP (east / north face), WTOP_ISOP (top of the cell) of every level; nullptr: off
  const double *HTE, *HTN;
  const double *HYX, *HXY, *RBR, *DXT, *DYT, *HBLT;   // HBLT: nullptr without KPP (BL_DEPTH = zw(1))
  double ah, ah_bolus, ah_bkg_srfbl, slm_r, slm_b;
  int diff_tapering, cancellation, slope_tanh;
  int slope_ctl;                    // 0 notanh, 1 tanh, 2 clip, 3 Gerd
  const double *HUS, *HUW;          // clip
  const double *kdepth;             // kappa type 'depth': KAPPA_VERTICAL(k), nullptr otherwise
  int kappa_bkg;                    // use_const_ah_bkg_srfbl = .false.: HOR_DIFF from the (untapered) KAPPA_ISOP
  double ah_bkg_bottom;             // HOR_DIFF of the bottom half of the bottom cell, 0 = none
};

// tapering factor of DM95 / its polynomial stand-in (:1490-1539)
__device__ __forceinline__ double gm_taper23(double sla, double slm, int ctl) {
  if (ctl == 1) return (sla < slm) ? 0.5 * (1.0 - tanh(10.0 * sla / slm - 4.0)) : 0.0;
  if (ctl == 2) return 1.0;                                                       // clip: the slopes are limited instead
  if (ctl == 3) return (sla > slm) ? (slm / sla) * (slm / sla) : 1.0;             // Gerdes et al. (1991)
  double t = 1.0;
  if (sla > 0.2 * slm && sla < 0.6 * slm) t = 0.5 * (1.0 - (2.5 * sla / slm - 1.0) * (4.0 - fabs(10.0 * sla / slm - 4.0)));
  else if (sla >= 0.6 * slm) t = 0.0;
  return t;
}

// KAPPA_VERTICAL = N^2 / N_ref^2 in [0.1, 1] below the surface diabatic layer SDL (= KPP_HBLT or zw(1)), 1 above (:3011-3180).
// One thread per column of the block (ghost columns included: the coefficients are formed there too); N^2 of the interfaces is
// kept in registers only as far as the march needs it: the reference level K_MIN is the first interface below SDL with N^2 > 0,
// so the march first looks for it, then normalises from there down.
__global__ void __launch_bounds__(256)
k_gm_kappa_vertical(DevGrid g, GmDev w, const double *__restrict__ T, const double *__restrict__ S, double grav) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (p2 >= g.n2) return;
  const int km = g.km;
  const long long n2 = g.n2, q = (long long)b * n2 + p2, base = (long long)b * g.n3 + p2;
  const int kmt = g.KMT[q];
  const double sdl = w.tlt ? w.ID[q] : (w.HBLT ? w.HBLT[q] : g.zw[1]);   // :3081-3083
  // N^2 at the bottom of level k (k < KMT; 0 elsewhere, as the module array's initial value)
  auto bfsq = [&](int k) {
    if (!(k < kmt)) return 0.0;
    const long long o = base + (long long)(k - 1) * n2;
    double rt, rs;
    const MwjfP P = mwjf_level(g.pressz[k + 1]);
    mwjf_rho<true>(P, T[o], S[o], &rt, &rs);
    const double v = -(grav * g.dzwr[k] * (rt * (fmax(-2.0, T[o]) - fmax(-2.0, T[o + n2])) + rs * (S[o] - S[o + n2])));
    return fmax(0.0, v);
  };
  int kmin = (kmt != 0) ? km + 1 : 0;
  double ref = 0.0;
  for (int k = 1; k <= km - 1; ++k) {
    if (kmin == km + 1 && g.zw[k] > sdl && k <= kmt) { const double v = bfsq(k); if (v > 0.0) { ref = v; kmin = k; } }
  }
  // NORM(k) of the interfaces, KAPPA_VERTICAL(k) = NORM(k-1) where k > K_MIN and k <= KMT, else 1
  w.KV[base] = 1.0;
  for (int k = 2; k <= km; ++k) {
    const int ki = k - 1;                              // the interface above level k
    double nr;
    if (ki >= kmin && ki < kmt && ref != 0.0) nr = fmin(fmax(bfsq(ki) / ref, 0.1), 1.0); else nr = 1.0;
    // (:3152-3156 copies NORM(KMT-1) to the interface KMT, which only KAPPA_VERTICAL(KMT+1) would read -- below the bottom, where it is 1)
    w.KV[base + (long long)(k - 1) * n2] = (k > kmin && k <= kmt) ? nr : 1.0;
  }
}

// MODE 0: everything in one launch (no transition layer).  With the transition layer: MODE 1 stores the slopes and SLA_SAVE only (the
// transition-layer march needs SLA of the whole column first), MODE 2 is the full evaluation with the layer's rules: no near-surface
// taper, no slope taper down to the diabatic depth, HOR_DIFF = ah_bkg_srfbl everywhere (:1428-1439, 1591-1600), and at the end the
// vertical profile of apply_vertical_profile_to_isop_hor_diff (:3745-3846), which touches each half cell on its own
template <int MODE>
__global__ void __launch_bounds__(256)
k_gm_coeffs(DevGrid g, GmDev w, const double *__restrict__ T, const double *__restrict__ S) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int kk = blockIdx.y + 1, b = blockIdx.z;
  if (p2 >= g.n2) return;
  const int nxb = g.nxb, nyb = g.nyb, km = g.km, i = p2 % nxb, j = p2 / nxb;
  const long long n2 = g.n2, q = (long long)b * n2 + p2, o = (long long)b * g.n3 + (long long)(kk - 1) * n2 + p2;
  const int kmt = g.KMT[q];
  auto temp = [&](long long oo) { return fmax(-2.0, T[oo]); };
  double sl[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};   // [half][xe, xw, yn, ys]
  if (MODE != 2) {
    // Straight-line (round 4): every neighbour is read at a clamped, always valid index and the conditions select values, so the loads of a
    // cell are requested together (with the loads inside `if (i >= 1)` ... each was a round trip of its own).  Same values, same operations.
    const bool he = i <= nxb - 2, hw = i >= 1, hn = j <= nyb - 2, hs = j >= 1;
    const long long de = he ? 1 : 0, dw = hw ? 1 : 0, dn = hn ? nxb : 0, ds = hs ? nxb : 0;
    const int ke = g.KMT[q + de], kw = g.KMT[q - dw], kn = g.KMT[q + dn], ks = g.KMT[q - ds];
    const double tc = temp(o), sc = S[o];
    const double te = temp(o + de), tw = temp(o - dw), tn = temp(o + dn), ts = temp(o - ds);
    const double se = S[o + de], sw = S[o - dw], sn = S[o + dn], ss = S[o - ds];
    const long long dup = (kk >= 2) ? n2 : 0, ddn = (kk < km) ? n2 : 0;
    const double tu = temp(o - dup), su = S[o - dup], tb = temp(o + ddn), sb = S[o + ddn];
    // horizontal differences of level kk on the east / north face of this cell and of its west / south neighbour
    const double mke = ((kk <= kmt) & (kk <= ke)) ? 1.0 : 0.0, mkw = ((kk <= kw) & (kk <= kmt)) ? 1.0 : 0.0;
    const double mkn = ((kk <= kmt) & (kk <= kn)) ? 1.0 : 0.0, mks = ((kk <= ks) & (kk <= kmt)) ? 1.0 : 0.0;
    const double txp_e = he ? mke * (te - tc) : 0.0, txs_e = he ? mke * (se - sc) : 0.0;
    const double txp_w = hw ? mkw * (tc - tw) : 0.0, txs_w = hw ? mkw * (sc - sw) : 0.0;
    const double typ_n = hn ? mkn * (tn - tc) : 0.0, tys_n = hn ? mkn * (sn - sc) : 0.0;
    const double typ_s = hs ? mks * (tc - ts) : 0.0, tys_s = hs ? mks * (sc - ss) : 0.0;
    double drdt, drds;
    const MwjfP P = mwjf_level(g.pressz[kk]);
    mwjf_rho<true>(P, T[o], S[o], &drdt, &drds);
    const double rxe = drdt * txp_e + drds * txs_e, rxw = hw ? drdt * txp_w + drds * txs_w : 0.0;
    const double ryn = drdt * typ_n + drds * tys_n, rys = hs ? drdt * typ_s + drds * tys_s : 0.0;
    {                                 // top half: the vertical difference to the level above, with this level's expansion coefficients (share :383-400)
      double rz = drdt * (tu - tc) + drds * (su - sc);
      rz = fmin(rz, -1.0e-20);
      if (kk >= 2 && kk <= kmt) { sl[0][0] = rxe / rz; sl[0][1] = rxw / rz; sl[0][2] = ryn / rz; sl[0][3] = rys / rz; }
    }
    {                                 // bottom half: the difference to the level below (share :278-291)
      double rz = drdt * (tc - tb) + drds * (sc - sb);
      rz = fmin(rz, -1.0e-20);
      const double mk = (kk < kmt) ? 1.0 : 0.0;
      if (kk < km) { sl[1][0] = mk * rxe / rz; sl[1][1] = mk * rxw / rz; sl[1][2] = mk * ryn / rz; sl[1][3] = mk * rys / rz; }
    }
  } else {
    // MODE 2 follows MODE 1 on the same tracers: the slopes MODE 1 stored (unclipped) are the ones this launch would form again
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) { sl[s2][0] = w.SLX[0 + s2][o]; sl[s2][1] = w.SLX[2 + s2][o]; sl[s2][2] = w.SLY[0 + s2][o]; sl[s2][3] = w.SLY[2 + s2][o]; }
  }
  const double dxt = w.DXT[q], dyt = w.DYT[q], rbr = w.RBR[q];
  const double bl = w.HBLT ? w.HBLT[q] : g.zw[1];
  const double dz_bottom = (kk == 1) ? 0.0 : g.zt[kk - 1];
  const int kp1r = min(kk + 1, km);
  const double refdepth[2] = {(kk == km) ? g.zw[kp1r] : g.zt[kp1r], g.zw[kp1r]};   // :1408-1411
  const double ddq = (MODE == 2) ? w.DD[q] : 0.0;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int kid = kk + s - 1;
    const double sla = (MODE == 2) ? w.SLA[s][o]      // (stored by MODE 1: the same expression of the same slopes)
                                   : g.dzw[kid] * sqrt(0.5 * ((sl[s][0] * sl[s][0] + sl[s][1] * sl[s][1]) / (dxt * dxt) + (sl[s][2] * sl[s][2] + sl[s][3] * sl[s][3]) / (dyt * dyt))) + 1.0e-10;
    if (MODE == 1) {
      w.SLA[s][o] = sla;
      w.SLX[0 + s][o] = sl[s][0]; w.SLX[2 + s][o] = sl[s][1]; w.SLY[0 + s][o] = sl[s][2]; w.SLY[2 + s][o] = sl[s][3];
      continue;
    }
    const double w1 = fmin(1.0, g.zt[kk] * rbr / sla);
    const double t1f = w.slope_tanh ? 0.5 * (1.0 + sin(3.14159265358979323846 * (w1 - 0.5))) : (0.5 + 2.0 * (w1 - 0.5) * (1.0 - fabs(w1 - 0.5)));
    const double taper1 = (MODE == 2) ? 1.0 : ((dz_bottom <= bl) ? t1f : 1.0);
    double taper2 = gm_taper23(sla, w.slm_r, w.slope_ctl);
    double taper3 = w.diff_tapering ? gm_taper23(sla, w.slm_b, w.slope_ctl) : taper2;
    if (w.slope_ctl == 2) {   // :1541-1573: slope clipping (after SLA, which keeps the unclipped slopes)
      const double dzwk = g.dzw[kid], dzwrk = g.dzwr[kid], hus = w.HUS[q], huw = w.HUW[q];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        if (fabs(sl[s][n] * dzwk / hus) > w.slm_r) sl[s][n] = copysign(w.slm_r * hus * dzwrk, sl[s][n]);
        if (fabs(sl[s][2 + n] * dzwk / huw) > w.slm_r) sl[s][2 + n] = copysign(w.slm_r * huw * dzwrk, sl[s][2 + n]);
      }
    }
    if (MODE == 2 && refdepth[s] <= ddq) { taper2 = 1.0; taper3 = 1.0; }
    const double kv = w.kdepth ? w.kdepth[kk] : (w.KV ? w.KV[o] : 1.0);
    // KAPPA_LATERAL * max(KAPPA_VERTICAL, kappa_*_deep) with 'bfre' (:1353-1358, 1382-1387), the constants otherwise
    const double kis = w.kdepth ? w.ah * kv : (w.KV ? w.ah * fmax(kv, 0.1) : w.ah), kts = w.kdepth ? w.ah_bolus * kv : (w.KV ? w.ah_bolus * fmax(kv, 0.1) : w.ah_bolus);   // 'depth': :1360-1366, 1375-1381
    double hd = w.kappa_bkg ? ((dz_bottom <= bl) ? kis * (1.0 - taper1 * taper2) : 0.0)                                // :1624-1631
                            : ((dz_bottom <= bl) ? w.ah_bkg_srfbl * (1.0 - taper1 * taper2) * kv : 0.0);
    double ki = taper1 * taper2 * kis, kt = taper1 * taper3 * kts;
    if (MODE == 2) hd = w.kappa_bkg ? kis : w.ah_bkg_srfbl;               // :1598-1608
    if (kk == 1 && s == 0) { hd = w.kappa_bkg ? kis : w.ah_bkg_srfbl; ki = 0.0; kt = 0.0; }   // :1208, :1369-1370, :1663-1664
    if (s == 1 && kk == kmt) { ki = 0.0; kt = 0.0; }                       // :1654-1657
    if (MODE == 2 && kk <= kmt) {                                          // apply_vertical_profile_to_isop_hor_diff
      const double rd = (s == 0) ? g.zt[kk] - 0.25 * g.dz[kk] : g.zt[kk] + 0.25 * g.dz[kk], th = w.TH[q], idq = w.ID[q];
      if (rd <= ddq) ki = 0.0;
      if (rd > ddq && rd <= idq && th > 1.0e-10) { hd = (idq - rd) * hd / th; ki = (rd - ddq) * ki / th; }
      if (rd > idq) hd = 0.0;
    }
    if (w.ah_bkg_bottom != 0.0 && s == 1 && kk == kmt) hd = w.ah_bkg_bottom;   // :1757-1761 (at the level's turn, after everything above)
    w.KI[s][o] = ki; w.KT[s][o] = kt; w.HD[s][o] = hd;
    if (MODE != 2 || w.slope_ctl == 2) {   // (MODE 2 without clipping: the stored slopes are already these)
      w.SLX[0 + s][o] = sl[s][0]; w.SLX[2 + s][o] = sl[s][1]; w.SLY[0 + s][o] = sl[s][2]; w.SLY[2 + s][o] = sl[s][3];
    }
  }
}

// TLT%DIABATIC_DEPTH: smooth_hblt(.false., .true.) (vmix_kpp.F90:3699-3881) -- one pass of the 1-1-4-1-1 filter over HMXL with land
// neighbours' weights folded into the centre, capped at the depth of the bottom T point; the outermost ring of the block keeps HMXL
__global__ void __launch_bounds__(256)
k_gm_diabatic_depth(DevGrid g, GmDev w) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (p2 >= g.n2) return;
  const int nxb = g.nxb, nyb = g.nyb, i = p2 % nxb, j = p2 / nxb;
  const long long q = (long long)b * g.n2 + p2;
  if (!w.HMXL) { w.DD[q] = g.zw[1]; return; }
  double out = w.HMXL[q];
  if (i >= 1 && i <= nxb - 2 && j >= 1 && j <= nyb - 2) {
    const int kmt = g.KMT[q];
    if (kmt != 0) {
      double cw = 0.125, ce = 0.125, cn = 0.125, cs = 0.125, cc = 0.5;
      if (g.KMT[q - 1] == 0) { cc = cc + cw; cw = 0.0; }
      if (g.KMT[q + 1] == 0) { cc = cc + ce; ce = 0.0; }
      if (g.KMT[q - nxb] == 0) { cc = cc + cs; cs = 0.0; }
      if (g.KMT[q + nxb] == 0) { cc = cc + cn; cn = 0.0; }
      out = cw * w.HMXL[q - 1] + ce * w.HMXL[q + 1] + cs * w.HMXL[q - nxb] + cn * w.HMXL[q + nxb] + cc * w.HMXL[q];
    }
    if (kmt >= 1 && kmt <= g.km && out > g.zt[kmt]) out = g.zt[kmt];
  }
  w.DD[q] = out;
}

// transition_layer (:3183-3440): one thread per column, the reference's state machine over the levels
__global__ void __launch_bounds__(256)
k_gm_transition_layer(DevGrid g, GmDev w) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (p2 >= g.n2) return;
  const int km = g.km;
  const long long n2 = g.n2, q = (long long)b * n2 + p2, base = (long long)b * g.n3 + p2;
  const int kmt = g.KMT[q];
  const double dd = w.DD[q], rb = 1.0 / w.RBR[q];
  auto sla = [&](int s, int kk) { return w.SLA[s][base + (long long)(kk - 1) * n2]; };
  int kstart = 0, ksub = 0, klev = 0, ztw = 0;
  bool compute = kmt != 0;
  double thick = 0.0, interior = 0.0;
  for (int k = 1; k <= km; ++k) {
    if (compute && dd < g.zw[k]) { kstart = k + 1; ksub = 1; thick = g.zw[k] - dd; klev = k; ztw = 2; compute = false; }
    if (k != 1 && kstart == k + 1 && dd < g.zt[k]) { kstart = k; ksub = 2; thick = g.zt[k] - dd; klev = k; ztw = 1; }
  }
  compute = !(kmt == 0 || kstart > kmt || (kstart == kmt && ksub == 2));
  for (int k = 1; k <= km - 1; ++k) {
    double work = 0.0;
    if (compute && ksub == 2 && kstart < kmt && kstart == k) work = fmax(sla(1, k), sla(0, k + 1)) * rb;
    if (work != 0.0 && dd < (g.zw[k] - work)) compute = false;
    if (work != 0.0 && dd >= (g.zw[k] - work)) { kstart = kstart + 1; ksub = 1; thick = g.zw[k] - dd; klev = k; ztw = 2; }
  }
  for (int k = 2; k <= km; ++k) {
    for (int kk = 1; kk <= 2; ++kk) {
      const double refd = (kk == 1) ? g.zt[k] : g.zw[k];
      double work = 0.0;
      if (kk == 1) {
        if (compute && kstart <= kmt && kstart == k) work = fmax(sla(0, k), sla(1, k)) * rb;
      } else {
        if (k < km && compute && kstart < kmt && kstart == k) work = fmax(sla(1, k), sla(0, k + 1)) * rb;
        if (compute && kstart == kmt && kstart == k) work = sla(1, k) * rb;
      }
      if (work != 0.0 && dd < (refd - work)) compute = false;
      if (work != 0.0 && dd >= (refd - work)) { thick = refd - dd; klev = k; ztw = kk; }
    }
    if (compute && kstart == k) kstart = kstart + 1;
  }
  if (klev >= 1 && ztw == 1) interior = g.zt[klev];
  if (klev >= 1 && ztw == 2) interior = g.zw[klev];
  w.TH[q] = thick; w.ID[q] = interior; w.KL[q] = klev; w.ZTW[q] = ztw;
}

// merged_streamfunction, first part (:3488-3566): the interior streamfunction and its first derivative at the interior depth of the
// column, for the east / north (face 0) and west / south (face 1) quarter cells -> MW[2 * {0: WORK1, 1: WORK2, 2: WORK3, 3: WORK4} + face]
__global__ void __launch_bounds__(256)
k_gm_msf_column(DevGrid g, GmDev w) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (p2 >= g.n2) return;
  const int km = g.km;
  const long long n2 = g.n2, q = (long long)b * n2 + p2, base = (long long)b * g.n3 + p2;
  const int kmt = g.KMT[q], k = w.KL[q], zt = w.ZTW[q];
  double W1[2] = {0.0, 0.0}, W2[2] = {0.0, 0.0}, W3[2] = {0.0, 0.0}, W4[2] = {0.0, 0.0};
  if (k >= 1 && k <= km - 1 && k < kmt) {
    const long long o = base + (long long)(k - 1) * n2, o1 = o + n2, o2 = o1 + n2;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const double *SXt = w.SLX[2 * n], *SXb = w.SLX[2 * n + 1], *SYt = w.SLY[2 * n], *SYb = w.SLY[2 * n + 1];
      if (zt == 1) {
        W1[n] = w.KT[1][o] * SXb[o] * g.dz[k];
        W2[n] = 2.0 * g.dzwr[k] * (W1[n] - w.KT[0][o1] * SXt[o1] * g.dz[k + 1]);
        const double w2n = 2.0 * (w.KT[0][o1] * SXt[o1] - w.KT[1][o1] * SXb[o1]);
        W3[n] = w.KT[1][o] * SYb[o] * g.dz[k];
        W4[n] = 2.0 * g.dzwr[k] * (W3[n] - w.KT[0][o1] * SYt[o1] * g.dz[k + 1]);
        const double w4n = 2.0 * (w.KT[0][o1] * SYt[o1] - w.KT[1][o1] * SYb[o1]);
        if (fabs(w2n) < fabs(W2[n])) W2[n] = w2n;
        if (fabs(w4n) < fabs(W4[n])) W4[n] = w4n;
      } else if (zt == 2) {
        W1[n] = w.KT[0][o1] * SXt[o1];
        W2[n] = 2.0 * (W1[n] - (w.KT[1][o1] * SXb[o1]));
        W1[n] = W1[n] * g.dz[k + 1];
        W3[n] = w.KT[0][o1] * SYt[o1];
        W4[n] = 2.0 * (W3[n] - (w.KT[1][o1] * SYb[o1]));
        W3[n] = W3[n] * g.dz[k + 1];
        if (k + 1 < kmt && k < km - 1) {
          const double w2n = 2.0 * g.dzwr[k + 1] * (w.KT[1][o1] * SXb[o1] * g.dz[k + 1] - w.KT[0][o2] * SXt[o2] * g.dz[k + 2]);
          const double w4n = 2.0 * g.dzwr[k + 1] * (w.KT[1][o1] * SYb[o1] * g.dz[k + 1] - w.KT[0][o2] * SYt[o2] * g.dz[k + 2]);
          if (fabs(w2n) < fabs(W2[n])) W2[n] = w2n;
          if (fabs(w4n) < fabs(W4[n])) W4[n] = w4n;
        }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) { w.MW[0 + n][q] = W1[n]; w.MW[2 + n][q] = W2[n]; w.MW[4 + n][q] = W3[n]; w.MW[6 + n][q] = W4[n]; }
}

// horizontal differences of tracer X at level kk (TX, TY of the reference) on the east / north face of 2-D cell q (3-D index o)
__device__ __forceinline__ double gm_tx(const DevGrid &g, const double *__restrict__ X, int kk, long long q, long long o) {
  return ((kk <= g.KMT[q] && kk <= g.KMT[q + 1]) ? 1.0 : 0.0) * (X[o + 1] - X[o]);
}
__device__ __forceinline__ double gm_ty(const DevGrid &g, const double *__restrict__ X, int kk, long long q, long long o) {
  return ((kk <= g.KMT[q] && kk <= g.KMT[q + g.nxb]) ? 1.0 : 0.0) * (X[o + g.nxb] - X[o]);
}
// TZ(kk) = X(kk-1) - X(kk), 0 at kk = 1 (never assigned in the reference)
__device__ __forceinline__ double gm_tz(const DevGrid &g, const double *__restrict__ X, int kk, long long o) { return (kk >= 2) ? X[o - g.n2] - X[o] : 0.0; }
// SF_SLX / SF_SLY of cell q (3-D index o) at level kk: xy 0 = x (SLX), 1 = y (SLY); face 0 = east / north, 1 = west / south; half 0 =
// top, 1 = bottom.  Without the transition layer (:1680-1700): KAPPA_THIC * slope * dz where kk <= KMT, else 0.  With it
// (merged_streamfunction, second part :3585-3741): linear in the diabatic region, quadratic in the transition layer, the plain
// product in the interior, by the depth of the middle of the half cell
__device__ __forceinline__ double gm_sf_form(const DevGrid &g, const GmDev &w, int xy, int face, int half, int kk, long long q, long long o) {
  if (!(kk <= g.KMT[q])) return 0.0;
  const double *__restrict__ SL = xy ? w.SLY[2 * face + half] : w.SLX[2 * face + half];
  if (!w.tlt) return w.KT[half][o] * SL[o] * g.dz[kk];
  const double rd = half ? g.zt[kk] + 0.25 * g.dz[kk] : g.zt[kk] - 0.25 * g.dz[kk];
  const double dd = w.DD[q], th = w.TH[q], id = w.ID[q];
  if (rd > id) return w.KT[half][o] * SL[o] * g.dz[kk];
  const double wa = w.MW[4 * xy + face][q], wb = w.MW[4 * xy + 2 + face][q];      // WORK1 | WORK3 and WORK2 | WORK4
  const double w5 = 1.0 / (2.0 * dd + th);                                        // KMT /= 0 here
  const double lin = rd * w5 * (2.0 * wa + th * wb);
  if (rd <= dd) return lin;
  const double w6 = (th > 1.0e-10) ? w5 / th : 0.0, w7 = (dd - rd) * (dd - rd);
  return -w7 * w6 * (wa + id * wb) + lin;
}

__device__ __forceinline__ double gm_sf(const DevGrid &g, const GmDev &w, int xy, int face, int half, int kk, long long q, long long o) {
  if (w.SF[0]) return w.SF[4 * xy + 2 * face + half][o];
  return gm_sf_form(g, w, xy, face, half, kk, q, o);
}
__global__ void __launch_bounds__(256)
k_gm_sf(DevGrid g, GmDev w) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int kk = blockIdx.y + 1, b = blockIdx.z;
  if (p2 >= g.n2) return;
  const long long q = (long long)b * g.n2 + p2, o = (long long)b * g.n3 + (long long)(kk - 1) * g.n2 + p2;
#pragma unroll
  for (int xy = 0; xy < 2; ++xy)
#pragma unroll
    for (int face = 0; face < 2; ++face)
#pragma unroll
      for (int half = 0; half < 2; ++half) w.SF[4 * xy + 2 * face + half][o] = gm_sf_form(g, w, xy, face, half, kk, q, o);
}

// Both tracers of the path go through every flux function together: the coefficients (diffusivities, slopes, masks, metrics) are
// loaded and combined once, the tracer differences enter linearly -- per tracer the operations and their order are the reference's.
struct Gm2 { double a, b; };
// flux through the east face of cell q at level k (:1827, :1832-1868); the caller guarantees i <= nxb - 2
__device__ __forceinline__ Gm2 gm_fx(const DevGrid &g, const GmDev &w, const double *__restrict__ X0, const double *__restrict__ X1, int k, long long q, long long o) {
  const int km = g.km;
  const double cx = (k <= g.KMT[q] && k <= g.KMT[q + 1]) ? w.HYX[q] * 0.25 : 0.0;
  const double work3 = w.KI[0][o] + w.HD[0][o] + w.KI[1][o] + w.HD[1][o] + w.KI[0][o + 1] + w.HD[0][o + 1] + w.KI[1][o + 1] + w.HD[1][o + 1];
  Gm2 f = {g.dz[k] * cx * gm_tx(g, X0, k, q, o) * work3, g.dz[k] * cx * gm_tx(g, X1, k, q, o) * work3};
  if (!w.cancellation) {
    const int kp1 = (k == km) ? k : k + 1;
    const long long okp = o + (long long)(kp1 - k) * g.n2;
    const double w1 = w.KI[0][o] * w.SLX[0][o] * g.dz[k] - gm_sf(g, w, 0, 0, 0, k, q, o);
    const double w2 = w.KI[1][o] * w.SLX[1][o] * g.dz[k] - gm_sf(g, w, 0, 0, 1, k, q, o);
    const double w3 = w.KI[0][o + 1] * w.SLX[2][o + 1] * g.dz[k] - gm_sf(g, w, 0, 1, 0, k, q + 1, o + 1);
    const double w4 = w.KI[1][o + 1] * w.SLX[3][o + 1] * g.dz[k] - gm_sf(g, w, 0, 1, 1, k, q + 1, o + 1);
    f.a = f.a - cx * (w1 * gm_tz(g, X0, k, o) + w2 * gm_tz(g, X0, kp1, okp) + w3 * gm_tz(g, X0, k, o + 1) + w4 * gm_tz(g, X0, kp1, okp + 1));
    f.b = f.b - cx * (w1 * gm_tz(g, X1, k, o) + w2 * gm_tz(g, X1, kp1, okp) + w3 * gm_tz(g, X1, k, o + 1) + w4 * gm_tz(g, X1, kp1, okp + 1));
  }
  return f;
}
__device__ __forceinline__ Gm2 gm_fy(const DevGrid &g, const GmDev &w, const double *__restrict__ X0, const double *__restrict__ X1, int k, long long q, long long o) {
  const int km = g.km, nxb = g.nxb;
  const double cy = (k <= g.KMT[q] && k <= g.KMT[q + nxb]) ? w.HXY[q] * 0.25 : 0.0;
  const double work4 = w.KI[0][o] + w.HD[0][o] + w.KI[1][o] + w.HD[1][o] + w.KI[0][o + nxb] + w.HD[0][o + nxb] + w.KI[1][o + nxb] + w.HD[1][o + nxb];
  Gm2 f = {g.dz[k] * cy * gm_ty(g, X0, k, q, o) * work4, g.dz[k] * cy * gm_ty(g, X1, k, q, o) * work4};
  if (!w.cancellation) {
    const int kp1 = (k == km) ? k : k + 1;
    const long long okp = o + (long long)(kp1 - k) * g.n2;
    const double w1 = w.KI[0][o] * w.SLY[0][o] * g.dz[k] - gm_sf(g, w, 1, 0, 0, k, q, o);
    const double w2 = w.KI[1][o] * w.SLY[1][o] * g.dz[k] - gm_sf(g, w, 1, 0, 1, k, q, o);
    const double w3 = w.KI[0][o + nxb] * w.SLY[2][o + nxb] * g.dz[k] - gm_sf(g, w, 1, 1, 0, k, q + nxb, o + nxb);
    const double w4 = w.KI[1][o + nxb] * w.SLY[3][o + nxb] * g.dz[k] - gm_sf(g, w, 1, 1, 1, k, q + nxb, o + nxb);
    f.a = f.a - cy * (w1 * gm_tz(g, X0, k, o) + w2 * gm_tz(g, X0, kp1, okp) + w3 * gm_tz(g, X0, k, o + nxb) + w4 * gm_tz(g, X0, kp1, okp + nxb));
    f.b = f.b - cy * (w1 * gm_tz(g, X1, k, o) + w2 * gm_tz(g, X1, kp1, okp) + w3 * gm_tz(g, X1, k, o + nxb) + w4 * gm_tz(g, X1, kp1, okp + nxb));
  }
  return f;
}
// flux through the bottom face of level k < km of physical cell q (:1910-2050)
__device__ __forceinline__ Gm2 gm_fz(const DevGrid &g, const GmDev &w, const double *__restrict__ X0, const double *__restrict__ X1, int k, long long q, long long o) {
  const int nxb = g.nxb, kp1 = k + 1;
  const long long okp = o + g.n2;
  const double kmask = (k < g.KMT[q]) ? 1.0 : 0.0;
  const double hyx = w.HYX[q], hxy = w.HXY[q], hyxw = w.HYX[q - 1], hxys = w.HXY[q - nxb];
  // east, north, west, south terms of one half cell at level kk (3-D index oo): coefficient * metric * horizontal difference
  auto faces = [&](const double *__restrict__ X, double ce, double cn, double cw, double cs, int kk, long long oo) {
    return ce * hyx * gm_tx(g, X, kk, q, oo) + cn * hxy * gm_ty(g, X, kk, q, oo) + cw * hyxw * gm_tx(g, X, kk, q - 1, oo - 1) + cs * hxys * gm_ty(g, X, kk, q - nxb, oo - nxb);
  };
  const double sb[4] = {w.SLX[1][o], w.SLY[1][o], w.SLX[3][o], w.SLY[3][o]};           // bottom half of level k: east, north, west, south
  const double st[4] = {w.SLX[0][okp], w.SLY[0][okp], w.SLX[2][okp], w.SLY[2][okp]};   // top half of level k + 1
  const double cb = g.dz[k] * w.KI[1][o], ct = g.dz[kp1] * w.KI[0][okp];
  Gm2 r;
  if (!w.cancellation) {
    // SF_SLX / SF_SLY of the bottom half of level k and of the top half of level k + 1: east, north, west, south
    const double fb[4] = {gm_sf(g, w, 0, 0, 1, k, q, o), gm_sf(g, w, 1, 0, 1, k, q, o), gm_sf(g, w, 0, 1, 1, k, q, o), gm_sf(g, w, 1, 1, 1, k, q, o)};
    const double ft[4] = {gm_sf(g, w, 0, 0, 0, kp1, q, okp), gm_sf(g, w, 1, 0, 0, kp1, q, okp), gm_sf(g, w, 0, 1, 0, kp1, q, okp), gm_sf(g, w, 1, 1, 0, kp1, q, okp)};
    auto one = [&](const double *__restrict__ X) {
      double w3 = 0.0;
      w3 = w3 + (cb * faces(X, sb[0], sb[1], sb[2], sb[3], k, o));
      w3 = w3 + faces(X, fb[0], fb[1], fb[2], fb[3], k, o);
      w3 = w3 + (ct * faces(X, st[0], st[1], st[2], st[3], kp1, okp));
      w3 = w3 + (1.0 * faces(X, ft[0], ft[1], ft[2], ft[3], kp1, okp));
      return -kmask * 0.25 * w3;
    };
    r.a = one(X0); r.b = one(X1);
    return r;
  }
  auto one = [&](const double *__restrict__ X) {
    double w3 = (cb * faces(X, sb[0], sb[1], sb[2], sb[3], k, o));
    w3 = w3 + (ct * faces(X, st[0], st[1], st[2], st[3], kp1, okp));
    return -kmask * 0.5 * w3;
  };
  r.a = one(X0); r.b = one(X1);
  return r;
}

// tendency of both tracers at every physical cell (0 elsewhere), and -- the coefficients being in registers -- the isopycnal part
// of the vertical diffusivity added to VDC(k) at EVERY cell of the block (:1725-1748; VDC1: the second array, nullptr when the two
// tracer classes share one)
#define POP_GM_KC 4   // levels per thread: the level above / below a level is then mostly the same thread's own earlier / later read
__global__ void __launch_bounds__(256)
k_gm_flux(DevGrid g, GmDev w, const double *__restrict__ X0, const double *__restrict__ X1, double *__restrict__ VDC0, double *__restrict__ VDC1) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int k0 = blockIdx.y * POP_GM_KC + 1, b = blockIdx.z;
  if (p2 >= g.n2) return;
  const int nxb = g.nxb, i = p2 % nxb, j = p2 / nxb, km = g.km;
  const long long n2 = g.n2, q = (long long)b * n2 + p2;
  const bool phys = i + 1 >= g.ib && i + 1 <= blk_ie(g, b) && j + 1 >= g.jb && j + 1 <= blk_je(g, b);
  const double hyx = w.HYX[q], hxy = w.HXY[q], hyxw = (i >= 1) ? w.HYX[q - 1] : 0.0, hxys = (j >= 1) ? w.HXY[q - nxb] : 0.0;
  const double tar = g.TAREA_R[q];
  const int kmt = g.KMT[q];
  Gm2 fztop = {0.0, 0.0};
  if (phys && k0 >= 2) fztop = gm_fz(g, w, X0, X1, k0 - 1, q, (long long)b * g.n3 + (long long)(k0 - 2) * n2 + p2);
  for (int k = k0; k < k0 + POP_GM_KC && k <= km; ++k) {
    const long long o = (long long)b * g.n3 + (long long)(k - 1) * n2 + p2;
    if (k < km) {
      const long long ok = o + n2;
      const double kmask = (k < kmt) ? 1.0 : 0.0;
      auto sq = [](double x) { return x * x; };
      const double add = g.dzw[k] * kmask * tar *
        (g.dz[k] * 0.25 * w.KI[1][o] * (hyx * sq(w.SLX[1][o]) + hyxw * sq(w.SLX[3][o]) + hxy * sq(w.SLY[1][o]) + hxys * sq(w.SLY[3][o])) +
         g.dz[k + 1] * 0.25 * w.KI[0][ok] * (hyx * sq(w.SLX[0][ok]) + hyxw * sq(w.SLX[2][ok]) + hxy * sq(w.SLY[0][ok]) + hxys * sq(w.SLY[2][ok])));
      const long long v = ((long long)b * (km + 2) + k) * n2 + p2;
      VDC0[v] = VDC0[v] + add;
      if (VDC1) VDC1[v] = VDC1[v] + add;
    }
    Gm2 gt = {0.0, 0.0};
    if (phys) {
      const Gm2 fxe = gm_fx(g, w, X0, X1, k, q, o), fxw = gm_fx(g, w, X0, X1, k, q - 1, o - 1);
      const Gm2 fyn = gm_fy(g, w, X0, X1, k, q, o), fys = gm_fy(g, w, X0, X1, k, q - nxb, o - nxb);
      const double sc = g.dzr[k];
      if (k < km) {
        const Gm2 fz = gm_fz(g, w, X0, X1, k, q, o);
        gt.a = (fxe.a - fxw.a + fyn.a - fys.a + fztop.a - fz.a) * sc * tar;
        gt.b = (fxe.b - fxw.b + fyn.b - fys.b + fztop.b - fz.b) * sc * tar;
        fztop = fz;                                  // FZTOP of the next level (:2003)
      } else {
        gt.a = (fxe.a - fxw.a + fyn.a - fys.a + fztop.a) * sc * tar;
        gt.b = (fxe.b - fxw.b + fyn.b - fys.b + fztop.b) * sc * tar;
      }
    }
    w.GTK[0][o] = gt.a; w.GTK[1][o] = gt.b;
  }
}

// ---- the flux functions once more, straight-line (round 4) ---------------------------------------------------------------------------
// gm_fx / gm_fy / gm_fz above guard loads with conditions (`kk >= 2 ? X[o - n2] ...`, `a && b` on two loaded values, the stored / re-derived
// choice of SF inside gm_sf): 522 loads behind 494 waits and 420 branches in k_gm_flux, every load a round trip of its own -- the kernel
// took 1.6 ms for 2 GB at gx1v7.  Here every load is unconditional at a clamped (always valid) address and conditions select VALUES, the
// two uniform choices (cancellation of the skew terms, SF stored by k_gm_sf) are template parameters.  Per value the same operations in the
// same order as above (tests/test_gpu_gm.py::test_gm_flux_tile_is_bitwise_the_cell_kernel compares the two sets bit for bit).
__device__ __forceinline__ double gmb_mask(int kk, int ka, int kb) { return ((kk <= ka) & (kk <= kb)) ? 1.0 : 0.0; }
__device__ __forceinline__ double gmb_tz(const double *__restrict__ X, int kk, long long o, long long n2) {
  const double d = X[o - (kk >= 2 ? n2 : 0)] - X[o];
  return (kk >= 2) ? d : 0.0;
}
// east-face flux of cell q (needs i <= nxb - 2)
template <bool CANCEL>
__device__ __forceinline__ Gm2 gmb_fx(const DevGrid &g, const GmDev &w, const double *__restrict__ X0, const double *__restrict__ X1, int k, long long q, long long o) {
  const long long n2 = g.n2;
  const int ka = g.KMT[q], kb = g.KMT[q + 1];
  const double hyx = w.HYX[q];
  const double cx = ((k <= ka) & (k <= kb)) ? hyx * 0.25 : 0.0;
  const double ki0 = w.KI[0][o], ki1 = w.KI[1][o], ki0e = w.KI[0][o + 1], ki1e = w.KI[1][o + 1];
  const double work3 = ki0 + w.HD[0][o] + ki1 + w.HD[1][o] + ki0e + w.HD[0][o + 1] + ki1e + w.HD[1][o + 1];
  const double m = gmb_mask(k, ka, kb), dz = g.dz[k];
  const double tx0 = m * (X0[o + 1] - X0[o]), tx1 = m * (X1[o + 1] - X1[o]);
  Gm2 f = {dz * cx * tx0 * work3, dz * cx * tx1 * work3};
  if (!CANCEL) {
    const int kp1 = (k == g.km) ? k : k + 1;
    const long long okp = o + (long long)(kp1 - k) * n2;
    const double w1 = ki0 * w.SLX[0][o] * dz - w.SF[0][o];              // SF[4 xy + 2 face + half]
    const double w2 = ki1 * w.SLX[1][o] * dz - w.SF[1][o];
    const double w3 = ki0e * w.SLX[2][o + 1] * dz - w.SF[2][o + 1];
    const double w4 = ki1e * w.SLX[3][o + 1] * dz - w.SF[3][o + 1];
    f.a = f.a - cx * (w1 * gmb_tz(X0, k, o, n2) + w2 * gmb_tz(X0, kp1, okp, n2) + w3 * gmb_tz(X0, k, o + 1, n2) + w4 * gmb_tz(X0, kp1, okp + 1, n2));
    f.b = f.b - cx * (w1 * gmb_tz(X1, k, o, n2) + w2 * gmb_tz(X1, kp1, okp, n2) + w3 * gmb_tz(X1, k, o + 1, n2) + w4 * gmb_tz(X1, kp1, okp + 1, n2));
  }
  return f;
}
// north-face flux of cell q (needs j <= nyb - 2)
template <bool CANCEL>
__device__ __forceinline__ Gm2 gmb_fy(const DevGrid &g, const GmDev &w, const double *__restrict__ X0, const double *__restrict__ X1, int k, long long q, long long o) {
  const long long n2 = g.n2;
  const int nxb = g.nxb;
  const int ka = g.KMT[q], kb = g.KMT[q + nxb];
  const double hxy = w.HXY[q];
  const double cy = ((k <= ka) & (k <= kb)) ? hxy * 0.25 : 0.0;
  const double ki0 = w.KI[0][o], ki1 = w.KI[1][o], ki0n = w.KI[0][o + nxb], ki1n = w.KI[1][o + nxb];
  const double work4 = ki0 + w.HD[0][o] + ki1 + w.HD[1][o] + ki0n + w.HD[0][o + nxb] + ki1n + w.HD[1][o + nxb];
  const double m = gmb_mask(k, ka, kb), dz = g.dz[k];
  const double ty0 = m * (X0[o + nxb] - X0[o]), ty1 = m * (X1[o + nxb] - X1[o]);
  Gm2 f = {dz * cy * ty0 * work4, dz * cy * ty1 * work4};
  if (!CANCEL) {
    const int kp1 = (k == g.km) ? k : k + 1;
    const long long okp = o + (long long)(kp1 - k) * n2;
    const double w1 = ki0 * w.SLY[0][o] * dz - w.SF[4][o];
    const double w2 = ki1 * w.SLY[1][o] * dz - w.SF[5][o];
    const double w3 = ki0n * w.SLY[2][o + nxb] * dz - w.SF[6][o + nxb];
    const double w4 = ki1n * w.SLY[3][o + nxb] * dz - w.SF[7][o + nxb];
    f.a = f.a - cy * (w1 * gmb_tz(X0, k, o, n2) + w2 * gmb_tz(X0, kp1, okp, n2) + w3 * gmb_tz(X0, k, o + nxb, n2) + w4 * gmb_tz(X0, kp1, okp + nxb, n2));
    f.b = f.b - cy * (w1 * gmb_tz(X1, k, o, n2) + w2 * gmb_tz(X1, kp1, okp, n2) + w3 * gmb_tz(X1, k, o + nxb, n2) + w4 * gmb_tz(X1, kp1, okp + nxb, n2));
  }
  return f;
}
// flux through the bottom face of level k < km of a PHYSICAL cell q (its west and south neighbours exist)
template <bool CANCEL>
__device__ __forceinline__ Gm2 gmb_fz(const DevGrid &g, const GmDev &w, const double *__restrict__ X0, const double *__restrict__ X1, int k, long long q, long long o) {
  const int nxb = g.nxb, kp1 = k + 1;
  const long long okp = o + g.n2;
  const int kc = g.KMT[q], ke = g.KMT[q + 1], kn = g.KMT[q + nxb], kw = g.KMT[q - 1], ks = g.KMT[q - nxb];
  const double kmask = (k < kc) ? 1.0 : 0.0;
  const double hyx = w.HYX[q], hxy = w.HXY[q], hyxw = w.HYX[q - 1], hxys = w.HXY[q - nxb];
  // masked horizontal differences of both tracers at level k and k + 1: east, north, west, south (gm_tx / gm_ty)
  const double me = gmb_mask(k, kc, ke), mn = gmb_mask(k, kc, kn), mw = gmb_mask(k, kw, kc), ms = gmb_mask(k, ks, kc);
  const double pe = gmb_mask(kp1, kc, ke), pn = gmb_mask(kp1, kc, kn), pw = gmb_mask(kp1, kw, kc), ps = gmb_mask(kp1, ks, kc);
  const double sb[4] = {w.SLX[1][o], w.SLY[1][o], w.SLX[3][o], w.SLY[3][o]};           // bottom half of level k: east, north, west, south
  const double st[4] = {w.SLX[0][okp], w.SLY[0][okp], w.SLX[2][okp], w.SLY[2][okp]};   // top half of level k + 1
  const double cb = g.dz[k] * w.KI[1][o], ct = g.dz[kp1] * w.KI[0][okp];
  double fb[4] = {0.0, 0.0, 0.0, 0.0}, ft[4] = {0.0, 0.0, 0.0, 0.0};
  if (!CANCEL) {
    fb[0] = w.SF[1][o]; fb[1] = w.SF[5][o]; fb[2] = w.SF[3][o]; fb[3] = w.SF[7][o];
    ft[0] = w.SF[0][okp]; ft[1] = w.SF[4][okp]; ft[2] = w.SF[2][okp]; ft[3] = w.SF[6][okp];
  }
  auto one = [&](const double *__restrict__ X) {
    const double xc = X[o], xp = X[okp];
    const double de = me * (X[o + 1] - xc), dn = mn * (X[o + nxb] - xc), dw = mw * (xc - X[o - 1]), ds = ms * (xc - X[o - nxb]);
    const double qe = pe * (X[okp + 1] - xp), qn = pn * (X[okp + nxb] - xp), qw = pw * (xp - X[okp - 1]), qs = ps * (xp - X[okp - nxb]);
    auto faces = [&](double ce, double cn, double cw, double cs, double e, double n, double w_, double s_) {
      return ce * hyx * e + cn * hxy * n + cw * hyxw * w_ + cs * hxys * s_;
    };
    if (!CANCEL) {
      double w3 = 0.0;
      w3 = w3 + (cb * faces(sb[0], sb[1], sb[2], sb[3], de, dn, dw, ds));
      w3 = w3 + faces(fb[0], fb[1], fb[2], fb[3], de, dn, dw, ds);
      w3 = w3 + (ct * faces(st[0], st[1], st[2], st[3], qe, qn, qw, qs));
      w3 = w3 + (1.0 * faces(ft[0], ft[1], ft[2], ft[3], qe, qn, qw, qs));
      return -kmask * 0.25 * w3;
    }
    double w3 = (cb * faces(sb[0], sb[1], sb[2], sb[3], de, dn, dw, ds));
    w3 = w3 + (ct * faces(st[0], st[1], st[2], st[3], qe, qn, qw, qs));
    return -kmask * 0.5 * w3;
  };
  Gm2 r;
  r.a = one(X0); r.b = one(X1);
  return r;
}

// k_gm_flux with the straight-line flux functions and every horizontal face flux formed ONCE (round 4).  A workgroup owns a 64 x R
// patch of the block of which it COMPUTES the tendency on the 63 x (R - 1) cells right of its first column and above its first row: a
// thread forms the east and the north flux of its own cell, takes the west flux from the lane to its left (a wave is one patch row: DPP
// shuffle) and the south flux from the row below through LDS; the first column and the first row only supply those fluxes, so no lane
// evaluates a second flux and no wave diverges (a first form let lane 0 / row 0 evaluate their neighbour's flux themselves: the whole
// wave waited for it, slower than k_gm_flux).  The isopycnal part of VDC is added by the thread that computes the cell.
template <int R, bool CANCEL>
__global__ void __launch_bounds__(64 * R)
k_gm_flux_tile(DevGrid g, GmDev w, const double *__restrict__ X0, const double *__restrict__ X1, double *__restrict__ VDC0, double *__restrict__ VDC1) {
  __shared__ double fy_a[2][R][64], fy_b[2][R][64];
  const int nxb = g.nxb, nyb = g.nyb, km = g.km;
  const int tiles_i = (nxb + 62) / 63;
  const int ti = blockIdx.x % tiles_i, tj = blockIdx.x / tiles_i;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int i = ti * 63 - 1 + tx, j = tj * (R - 1) - 1 + ty;              // the first column / row of the patch belongs to the neighbour
  const int k0 = blockIdx.y * POP_GM_KC + 1, b = blockIdx.z;
  const bool inb = i >= 0 && i < nxb && j >= 0 && j < nyb;
  const bool own = inb && tx >= 1 && ty >= 1;                             // this thread computes the cell
  const int p2 = inb ? j * nxb + i : 0;
  const long long n2 = g.n2, q = (long long)b * n2 + p2;
  const bool phys = own && i + 1 >= g.ib && i + 1 <= blk_ie(g, b) && j + 1 >= g.jb && j + 1 <= blk_je(g, b);
  const bool has_e = inb && i <= nxb - 2, has_n = inb && j <= nyb - 2;
  // lanes whose face does not exist evaluate the flux of a cell where it does (clamped index) and discard it: no divergent block around the loads
  const int pe2 = has_e ? p2 : 0, pn2 = has_n ? p2 : 0;
  const long long qe_ = (long long)b * n2 + pe2, qn_ = (long long)b * n2 + pn2;
  const long long qz = phys ? q : (long long)b * n2 + (long long)(g.jb - 1) * nxb + (g.ib - 1);   // a physical cell for the lanes that compute no tendency
  const int pz2 = (int)(qz - (long long)b * n2);
  const double hyx = w.HYX[q], hxy = w.HXY[q], hyxw = w.HYX[q - ((inb && i >= 1) ? 1 : 0)], hxys = w.HXY[q - ((inb && j >= 1) ? nxb : 0)];
  const double hyxw_ = (inb && i >= 1) ? hyxw : 0.0, hxys_ = (inb && j >= 1) ? hxys : 0.0;
  const double tar = g.TAREA_R[q];
  const int kmt = g.KMT[q];
  Gm2 fztop = {0.0, 0.0};
  if (k0 >= 2) {
    const Gm2 t = gmb_fz<CANCEL>(g, w, X0, X1, k0 - 1, qz, (long long)b * g.n3 + (long long)(k0 - 2) * n2 + pz2);
    if (phys) fztop = t;
  }
  for (int k = k0; k < k0 + POP_GM_KC && k <= km; ++k) {
    const long long lev = (long long)b * g.n3 + (long long)(k - 1) * n2;
    const long long o = lev + p2;
    const int buf = k & 1;
    {
      const int kq = (k < km) ? k : km - 1;                               // (k = km: computed at the level above and discarded)
      const long long oq = (long long)b * g.n3 + (long long)(kq - 1) * n2 + p2, ok = oq + n2;
      const double kmask = (kq < kmt) ? 1.0 : 0.0;
      auto sq = [](double x) { return x * x; };
      const double add = g.dzw[kq] * kmask * tar *
        (g.dz[kq] * 0.25 * w.KI[1][oq] * (hyx * sq(w.SLX[1][oq]) + hyxw_ * sq(w.SLX[3][oq]) + hxy * sq(w.SLY[1][oq]) + hxys_ * sq(w.SLY[3][oq])) +
         g.dz[kq + 1] * 0.25 * w.KI[0][ok] * (hyx * sq(w.SLX[0][ok]) + hyxw_ * sq(w.SLX[2][ok]) + hxy * sq(w.SLY[0][ok]) + hxys_ * sq(w.SLY[2][ok])));
      if (own && k < km) {
        const long long v = ((long long)b * (km + 2) + k) * n2 + p2;
        VDC0[v] = VDC0[v] + add;
        if (VDC1) VDC1[v] = VDC1[v] + add;
      }
    }
    Gm2 fxe = gmb_fx<CANCEL>(g, w, X0, X1, k, qe_, lev + pe2);
    Gm2 fyn = gmb_fy<CANCEL>(g, w, X0, X1, k, qn_, lev + pn2);
    if (!has_e) { fxe.a = 0.0; fxe.b = 0.0; }
    if (!has_n) { fyn.a = 0.0; fyn.b = 0.0; }
    fy_a[buf][ty][tx] = fyn.a; fy_b[buf][ty][tx] = fyn.b;
    const Gm2 fxw = {__shfl_up(fxe.a, 1), __shfl_up(fxe.b, 1)};
    const int kz = (k < km) ? k : km - 1;
    const Gm2 fzr = gmb_fz<CANCEL>(g, w, X0, X1, kz, qz, (long long)b * g.n3 + (long long)(kz - 1) * n2 + pz2);
    __syncthreads();                                         // (the buffer of level k - 1 is free again: everybody has passed this barrier since)
    Gm2 gt = {0.0, 0.0};
    if (phys) {
      const Gm2 fys = {fy_a[buf][ty - 1][tx], fy_b[buf][ty - 1][tx]};
      const double sc = g.dzr[k];
      if (k < km) {
        gt.a = (fxe.a - fxw.a + fyn.a - fys.a + fztop.a - fzr.a) * sc * tar;
        gt.b = (fxe.b - fxw.b + fyn.b - fys.b + fztop.b - fzr.b) * sc * tar;
        fztop = fzr;
      } else {
        gt.a = (fxe.a - fxw.a + fyn.a - fys.a + fztop.a) * sc * tar;
        gt.b = (fxe.b - fxw.b + fyn.b - fys.b + fztop.b) * sc * tar;
      }
    }
    if (own) { w.GTK[0][o] = gt.a; w.GTK[1][o] = gt.b; }
  }
}

// diag_gm_bolus (:2079-2151): the eddy-induced velocity.  One thread per column of the block marches the levels, carrying the stream
// function at the top of the level (UIT, VIT of the reference) on the four faces it needs: its own east and north face (U_ISOP,
// V_ISOP of the cell) and the east face of its west / north face of its south neighbour (the divergence that integrates to WISOP).
__global__ void __launch_bounds__(256)
k_gm_bolus(DevGrid g, GmDev w) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (p2 >= g.n2) return;
  const int nxb = g.nxb, nyb = g.nyb, km = g.km, i = p2 % nxb, j = p2 / nxb;
  const long long n2 = g.n2, q = (long long)b * n2 + p2, base = (long long)b * g.n3 + p2;
  const bool phys = i + 1 >= g.ib && i + 1 <= blk_ie(g, b) && j + 1 >= g.jb && j + 1 <= blk_je(g, b);
  const bool has_e = i <= nxb - 2, has_n = j <= nyb - 2;
  const int kmt = g.KMT[q];
  // stream function on the east face of cell c (2-D index qq, 3-D index oo) at the bottom of level k, and the same on its north face
  auto psi_e = [&](long long qq, long long oo, int k) {
    const double factor = (k < km) ? 1.0 : 0.0;
    const int kp1 = (k < km) ? k + 1 : k;
    const long long okp = oo + (long long)(kp1 - k) * n2;
    const double v = (gm_sf(g, w, 0, 0, 1, k, qq, oo) + factor * gm_sf(g, w, 0, 0, 0, kp1, qq, okp) +
                      gm_sf(g, w, 0, 1, 1, k, qq + 1, oo + 1) + factor * gm_sf(g, w, 0, 1, 0, kp1, qq + 1, okp + 1)) * 0.25 * w.HYX[qq];
    return (k < g.KMT[qq] && k < g.KMT[qq + 1]) ? v : 0.0;
  };
  auto psi_n = [&](long long qq, long long oo, int k) {
    const double factor = (k < km) ? 1.0 : 0.0;
    const int kp1 = (k < km) ? k + 1 : k;
    const long long okp = oo + (long long)(kp1 - k) * n2;
    const double v = (gm_sf(g, w, 1, 0, 1, k, qq, oo) + factor * gm_sf(g, w, 1, 0, 0, kp1, qq, okp) +
                      gm_sf(g, w, 1, 1, 1, k, qq + nxb, oo + nxb) + factor * gm_sf(g, w, 1, 1, 0, kp1, qq + nxb, okp + nxb)) * 0.25 * w.HXY[qq];
    return (k < g.KMT[qq] && k < g.KMT[qq + nxb]) ? v : 0.0;
  };
  double uit_e = 0.0, uit_w = 0.0, vit_n = 0.0, vit_s = 0.0, wtop = 0.0;
  const double hte = w.HTE[q], htn = w.HTN[q], tar = g.TAREA_R[q];
  for (int k = 1; k <= km; ++k) {
    const long long o = base + (long long)(k - 1) * n2;
    const double uib_e = has_e ? psi_e(q, o, k) : 0.0, vib_n = has_n ? psi_n(q, o, k) : 0.0;
    const double w1 = (has_e && k <= kmt && k <= g.KMT[q + 1]) ? uit_e - uib_e : 0.0;
    const double w2 = (has_n && k <= kmt && k <= g.KMT[q + nxb]) ? vit_n - vib_n : 0.0;
    w.UISOP[o] = w1 * g.dzr[k] / hte;
    w.VISOP[o] = w2 * g.dzr[k] / htn;
    w.WISOP[o] = wtop;
    if (phys) {
      const double uib_w = psi_e(q - 1, o - 1, k), vib_s = psi_n(q - nxb, o - nxb, k);
      const double w1w = (k <= g.KMT[q - 1] && k <= kmt) ? uit_w - uib_w : 0.0;
      const double w2s = (k <= g.KMT[q - nxb] && k <= kmt) ? vit_s - vib_s : 0.0;
      wtop = (k < kmt) ? wtop + tar * (w1 - w1w + w2 - w2s) : 0.0;      // WBOT_ISOP of this level = WTOP_ISOP of the next
      uit_w = uib_w; vit_s = vib_s;
    }
    uit_e = uib_e; vit_n = vib_n;
  }
}

}  // namespace pop
