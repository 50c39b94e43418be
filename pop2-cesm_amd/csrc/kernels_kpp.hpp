// kernels_kpp.hpp -- KPP vertical mixing (Large, McWilliams & Doney 1994) as the reference's
// native path computes it (the `.not. lcvmix` branches of source/vmix_kpp.F90).
//
// Column mapping: one thread per (i,j) of the WHOLE block array (ghost columns included,
// because the boundary-layer depth is smoothed horizontally and the viscosity is averaged to U
// points afterwards).  Six launches per step:
//   k_kpp_buoydiff   buoydiff      :3575-3621   DBLOC, DBSFC (O(km*kref) EOS evaluations/column)
//   k_kpp_interior   ri_iwmix      :1505-1603,1774-1929 + ddmix :3404-3497  interior VISC, VDC
//   k_kpp_ushear     bldepth       :2293-2325   reference-depth shear^2 at U points
//   k_kpp_bldepth    bldepth       :2140-2700   bulk Richardson march -> HBLT, KBL (unsmoothed)
//   k_kpp_blmix      smooth_hblt   :3797-3870 + blmix :2900-3222 + convection / masks :1218-1262
//                                  + non-local source :1283-1306
//   k_kpp_vvc        tgrid_to_ugrid of VISC -> VVC    :1253-1262
// Selections: no tidal / near-inertial-wave / Langmuir mixing, no short-wave penetration
// (lshort_wave=.false.), lcheckekmo=.false., SMFT available, no partial bottom cells.
// Integer powers use the usual expansion x**3=(x*x)*x, x**4=(x*x)*(x*x).
// Diagnostic outputs HMXL, HMXL_DR: k_kpp_hmxl, with pop_config kpp_ml_diagnostics = 1; tavg fields are not computed.
#pragma once

namespace pop {

struct KppDev {
  CArr zgrid, hwide, bckgrnd_vdc, bckgrnd_vvc;             // zgrid/hwide: 0..km+1 (ConstArr: wave-uniform level index -> scalar loads)
  CArrI kref;                                               // 1..km: surface-layer reference level
  CArr eosP;                                                // 6 (km + 2): mwjf_level of every level (k_kpp_level_table)
  // r3: bit k-1 of CONVB[column] = the interface below level k is not stably stratified (DBLOC(k) <= 0: the reference's N2 > 0 test,
  // vmix_kpp.F90:1218-1225, divides by a positive thickness).  Written by k_kpp_buoy_interior_march, read by the sparse form of
  // k_kpp_blmix instead of the DBLOC column; nullptr when the interior kernel of this evaluation does not form it
  unsigned long long *CONVB;
  int src_clear_all;                                        // the non-local source may hold non-zeros below the KBL stored with it (a caller wrote it)
  double *HBLT0, *USTAR, *BFSFC;                              // 2-D scratch
  int *KBL0, *KBL;
  double Vtc, cg, rich_mix;
  int lrich, ldbl_diff, nsmooth;
  // shear kernel limited to the levels the boundary-layer march is expected to reach (k_kpp_ushear_col with a hint): WUK[q] = number
  // of levels of WU that are valid at U point q; the march computes a level beyond it itself (kpp_ushear_point, same operations).
  // nullptr: every level of WU is valid
  int *WUK;
  int wu_margin;
  int vdc_same;            // the two tracer classes share ONE diffusivity array (VDC1 == VDC2): the kernels touch VDC1 only.  A flag, not a
                           // pointer comparison: both parameters are __restrict__, which lets the compiler assume they differ

  // vmix_kpp_nml lshort_wave (sw_absorption_type: 0 'top-layer', 1 'jerlov' with water type jerlov = 1..5), lcheckekmo
  int lshort_wave, sw_type, jerlov, lcheckekmo;
  const double *FCORT, *SHF_QSW;     // T-point Coriolis parameter (Ekman depth), surface short-wave flux
  // sw_absorption_type 'chlorophyll' (sw_type 2): transmission table Tr[(2 km + 1) * n + k] over the levels ztr[0 .. 2 km] and 401
  // chlorophyll amounts (sw_absorption.F90:525-728); CHLI = column of the table per cell (set_chl :500-512, computed on the host)
  const double *Tr, *ztr;
  const int *CHLI;
  int ksol;
  double *BO, *BOSOL;                // surface buoyancy forcing without / from the short-wave flux (lshort_wave: blmix needs both)
};

constexpr double KPP_EPSSFC = 0.1, KPP_RIINFTY = 0.8, KPP_RRHO0 = 2.55, KPP_DSFMAX = 1.0, KPP_CSTAR = 10.0;
constexpr double KPP_ZETA_M = -0.2, KPP_ZETA_S = -1.0, KPP_C_M = 8.38, KPP_C_S = 98.96, KPP_A_M = 1.26, KPP_A_S = -28.86;
constexpr double KPP_CONCV = 1.7, KPP_VONKAR = 0.4, KPP_EPS = 1.0e-10, KPP_EPS2 = 1.0e-20, KPP_RICR = 0.3;

// wscale (vmix_kpp.F90:3296-3337)
template <bool WANT_M>
__device__ __forceinline__ void kpp_wscale(double sigma, double hbl, double ustar, double bfsfc, double &wm, double &ws) {
  const double zetah = sigma * hbl * KPP_VONKAR * bfsfc;
  const double u3 = (ustar * ustar) * ustar;
  const double zeta = zetah / (u3 + KPP_EPS);
  if (WANT_M) {
    if (zeta >= 0.0) wm = KPP_VONKAR * ustar / (1.0 + 5.0 * zeta);
    else if (zeta >= KPP_ZETA_M) wm = KPP_VONKAR * ustar * pow(1.0 - 16.0 * zeta, 0.25);
    else wm = KPP_VONKAR * pow(KPP_A_M * u3 - KPP_C_M * zetah, 1.0 / 3.0);
  }
  if (zeta >= 0.0) ws = KPP_VONKAR * ustar / (1.0 + 5.0 * zeta);
  else if (zeta >= KPP_ZETA_S) ws = KPP_VONKAR * ustar * sqrt(1.0 - 16.0 * zeta);
  else ws = KPP_VONKAR * pow(KPP_A_S * u3 - KPP_C_S * zetah, 1.0 / 3.0);
}

__device__ __forceinline__ double tmask(double t) { return (t < -2.0) ? -2.0 : t; }
constexpr double KPP_CEKMAN = 0.7, KPP_CMONOB = 1.0;
// sw_absorb_frac (sw_absorption.F90:736-811): share of the surface short-wave flux that reaches `depth` (cm) in Jerlov water
// type jt = 1..5 (two exponentials, Simpson and Paulson 1977; zero below 200 m)
__device__ __forceinline__ double kpp_sw_absorb_frac(double depth, int jt) {
  const double rfac[5] = {0.58, 0.62, 0.67, 0.77, 0.78}, depth1[5] = {0.35, 0.60, 1.00, 1.50, 1.40}, depth2[5] = {23.0, 20.0, 17.0, 14.0, 7.90};
  const double dm = -depth * 0.01;
  if (dm < -200.0) return 0.0;
  return rfac[jt - 1] * exp(dm / depth1[jt - 1]) + (1.0 - rfac[jt - 1]) * exp(dm / depth2[jt - 1]);
}
// sw_trans_chl (sw_absorption.F90:951-1047): kin > 0: transmission to level kin of ztr; kin = 0: interpolated to depth ztrans
__device__ __forceinline__ double kpp_sw_trans_chl(const KppDev &kp, int kin, double ztrans, int idx) {
  const double *Trn = kp.Tr + (long long)idx * (kp.ksol + 1);
  if (kin > 0) return Trn[kin];
  int kindx = kp.ksol - 1;
  double w1 = 0.0, w2 = 0.0;
  for (int k = 1; k <= kp.ksol; ++k)
    if (kp.ztr[k - 1] <= ztrans && ztrans < kp.ztr[k]) {
      w2 = (ztrans - kp.ztr[k - 1]) / (kp.ztr[k] - kp.ztr[k - 1]);
      w1 = 1.0 - w2;
      kindx = k - 1;
    }
  return w1 * Trn[kindx] + w2 * Trn[kindx + 1];
}
// surface buoyancy forcing with the radiative contribution down to `depth` (vmix_kpp.F90:2236-2256, 2387-2412, 2707-2742);
// kin: the level of ztr that `depth` is (chlorophyll table), 0 = interpolate
__device__ __forceinline__ double kpp_bfsfc(const KppDev &kp, double bo, double bosol, double depth, int kin, int chlidx) {
  if (!kp.lshort_wave) return bo;
  if (kp.sw_type == 0) return bo + bosol;
  if (kp.sw_type == 2) return bo + bosol * (1.0 - kpp_sw_trans_chl(kp, kin, depth, chlidx));
  return bo + bosol * (1.0 - kpp_sw_absorb_frac(depth, kp.jerlov));
}

// ---- buoydiff: 3-D parallel, one thread per (i,j,k); level k yields DBSFC(k) and DBLOC(k-1) ------
// SFC = false: DBLOC only (k_kpp_bldepth<true, .> evaluates DBSFC on demand)
template <bool SFC>
__global__ void __launch_bounds__(256)
k_kpp_buoydiff(DevGrid g, KppDev kp, const double *__restrict__ T, const double *__restrict__ S,
               double *__restrict__ DBLOC, double *__restrict__ DBSFC) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y + 1, b = blockIdx.z;
  if (p2 >= g.n2) return;
  const int km = g.km;
  const long long n2 = g.n2;
  const long long base3 = (long long)b * g.n3 + p2;
  const long long o = base3 + (long long)(k - 1) * n2;
  if (k == 1) { if (SFC) DBSFC[o] = 0.0; return; }
  const int kmt = g.KMT[(long long)b * n2 + p2];
  const MwjfP P = mwjf_level(g.pressz[k]);
  const double rhokm = mwjf_rho<false>(P, tmask(T[o - n2]), S[o - n2], nullptr, nullptr);
  const double rhok = mwjf_rho<false>(P, tmask(T[o]), S[o], nullptr, nullptr);
  if (!SFC) {
    double dbl = 0.0;
    if (rhok != 0.0) dbl = GRAV * (1.0 - rhokm / rhok);
    if (k - 1 >= kmt) dbl = 0.0;
    DBLOC[o - n2] = dbl;
    if (k == km) DBLOC[o] = 0.0;
    return;
  }
  const double surfthick = KPP_EPSSFC * g.zt[k];
  const int kref = kp.kref[k];
  const long long orf = base3 + (long long)(kref - 1) * n2;
  double rhoavg = mwjf_rho<false>(P, tmask(T[orf]), S[orf], nullptr, nullptr);
  if (kref != 1) {
    rhoavg = rhoavg * (surfthick - g.zw[kref - 1]);
    int kt = 1;
    for (; kt + 3 <= kref - 1; kt += 4) {      // loads of 4 levels in flight, same order of additions
      double tt[4], ss[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { const long long ot = base3 + (long long)(kt + t - 1) * n2; tt[t] = T[ot]; ss[t] = S[ot]; }
#pragma unroll
      for (int t = 0; t < 4; ++t) rhoavg = rhoavg + g.dz[kt + t] * mwjf_rho<false>(P, tmask(tt[t]), ss[t], nullptr, nullptr);
    }
    for (; kt <= kref - 1; ++kt) {
      const long long ot = base3 + (long long)(kt - 1) * n2;
      rhoavg = rhoavg + g.dz[kt] * mwjf_rho<false>(P, tmask(T[ot]), S[ot], nullptr, nullptr);
    }
    rhoavg = rhoavg / surfthick;
  }
  double dbs = 0.0, dbl = 0.0;
  if (rhok != 0.0) { dbs = GRAV * (1.0 - rhoavg / rhok); dbl = GRAV * (1.0 - rhokm / rhok); }
  if (k - 1 >= kmt) dbl = 0.0;
  DBSFC[o] = dbs;
  DBLOC[o - n2] = dbl;
  if (k == km) DBLOC[o] = 0.0;
}

// ---- buoydiff, column form for bandwidth-bound grids ------------------------------------------
// The 3-D-parallel kernel re-reads the top kref levels of T and S for every level (66 GB through the
// fabric at tx0.1v3 for 17 GB of algorithmic traffic) and repeats the pressure-independent half of
// every equation-of-state evaluation.  Here one thread owns a column: the clamped T, 1000*S and the
// pressure-independent second term of the denominator (mwjf_prep2) of the top KR levels sit in registers
// (KR >= max kref, checked by the host), every field is read once, and only the pressure-dependent
// polynomials are re-evaluated.  Same operations in the same order as k_kpp_buoydiff.
template <int KR, int WAVES>
__global__ void __launch_bounds__(POP_COL_THREADS, WAVES)
k_kpp_buoydiff_col(DevGrid g, KppDev kp, const double *__restrict__ T, const double *__restrict__ S,
                   double *__restrict__ DBLOC, double *__restrict__ DBSFC) {
  Col c;
  if (!col_setup(g, c, false)) return;
  const int km = g.km;
  const long long n2 = g.n2;
  const int kmt = g.KMT[c.q2];
  MwjfTS2 top[KR + 1];
#pragma unroll
  for (int t = 1; t <= KR; ++t) {
    const int kk = (t <= km) ? t : km;
    const long long o = c.base3 + (long long)(kk - 1) * n2;
    top[t] = mwjf_prep2(tmask(T[o]), S[o]);
  }
  DBSFC[c.base3] = 0.0;
  MwjfTS2 xkm = top[1];
  for (int k = 2; k <= km; ++k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    const MwjfTS2 xk = mwjf_prep2(tmask(T[o]), S[o]);
    const MwjfP P = mwjf_level(g.pressz[k]);
    const double rhokm = mwjf_eval2(P, xkm);
    const double rhok = mwjf_eval2(P, xk);
    const double surfthick = KPP_EPSSFC * g.zt[k];
    const int kref = kp.kref[k];
    // kref is wave-uniform: the predicated iterations below are skipped as a whole
    MwjfTS2 xr = top[1];
#pragma unroll
    for (int t = 2; t <= KR; ++t) if (t == kref) xr = top[t];
    double rhoavg = mwjf_eval2(P, xr);
    if (kref != 1) {
      rhoavg = rhoavg * (surfthick - g.zw[kref - 1]);
#pragma unroll
      for (int kt = 1; kt <= KR - 1; ++kt)
        if (kt <= kref - 1) rhoavg = rhoavg + g.dz[kt] * mwjf_eval2(P, top[kt]);
      rhoavg = rhoavg / surfthick;
    }
    double dbs = 0.0, dbl = 0.0;
    if (rhok != 0.0) { dbs = GRAV * (1.0 - rhoavg / rhok); dbl = GRAV * (1.0 - rhokm / rhok); }
    if (k - 1 >= kmt) dbl = 0.0;
    DBSFC[o] = dbs;
    DBLOC[o - n2] = dbl;
    if (k == km) DBLOC[o] = 0.0;
    xkm = xk;
  }
}

// ---- buoydiff, level-parallel form with the surface layer in LDS ---------------------------------------------------
// The column form above is VALU-bound at one or two waves per SIMD (60 register doubles for the prepared surface-layer
// levels).  Here a workgroup owns 64 columns and NG waves share them: the prepared top KR levels (clamped T, 1000 S,
// pressure-independent denominator term) are formed once into LDS (KR * 3 * 512 B = 30 KB for KR = 20), and wave g takes the
// levels k = 2 + g, 2 + g + NG, ... (interleaved: the work per level grows with kref).  A thread keeps no column state, so
// five workgroups fit a CU and the divisions of one wave hide behind the others.  Every (i,j,k) value is formed by the same
// operations in the same order as in k_kpp_buoydiff / k_kpp_buoydiff_col: bitwise equal (tested).
template <int KR, int NG>
__global__ void __launch_bounds__(POP_COL_THREADS * NG)
k_kpp_buoydiff_lds(DevGrid g, KppDev kp, const double *__restrict__ T, const double *__restrict__ S,
                   double *__restrict__ DBLOC, double *__restrict__ DBSFC) {
  __shared__ double top[3][KR][POP_COL_THREADS];
  const int tx = threadIdx.x, gy = threadIdx.y;
  // same column order as col_setup (xcd_remap 0 / 1); surplus threads keep running to the barrier with a clamped column
  const int tile = g.xcd_remap ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const long long p2raw = (long long)tile * POP_COL_THREADS + tx;
  if (land_run(g, blockIdx.y, (long long)tile * POP_COL_THREADS, POP_COL_THREADS)) return;   // whole workgroup: before any barrier
  const bool live = p2raw < g.n2;
  const int p2 = live ? (int)p2raw : 0;
  const int b = blockIdx.y, km = g.km;
  const long long n2 = g.n2, base3 = (long long)b * g.n3 + p2;
  for (int t = 1 + gy; t <= KR; t += NG) {
    const int kk = (t <= km) ? t : km;
    const long long o = base3 + (long long)(kk - 1) * n2;
    const MwjfTS2 x = mwjf_prep2(tmask(T[o]), S[o]);
    top[0][t - 1][tx] = x.TQ; top[1][t - 1][tx] = x.SQ; top[2][t - 1][tx] = x.A2;
  }
  __syncthreads();
  if (!live) return;
  const int kmt = g.KMT[(long long)b * n2 + p2];
  if (gy == 0) DBSFC[base3] = 0.0;
  auto top_at = [&](int t) { MwjfTS2 x; x.TQ = top[0][t - 1][tx]; x.SQ = top[1][t - 1][tx]; x.A2 = top[2][t - 1][tx]; return x; };
  for (int k = 2 + gy; k <= km; k += NG) {
    const long long o = base3 + (long long)(k - 1) * n2;
    const double tk = T[o], sk = S[o], tm = T[o - n2], sm = S[o - n2];
    const MwjfTS2 xk = mwjf_prep2(tmask(tk), sk), xkm = mwjf_prep2(tmask(tm), sm);
    const MwjfP P = mwjf_level(g.pressz[k]);
    const double rhokm = mwjf_eval2(P, xkm);
    const double rhok = mwjf_eval2(P, xk);
    const double surfthick = KPP_EPSSFC * g.zt[k];
    const int kref = kp.kref[k];                 // wave-uniform (k is)
    double rhoavg = mwjf_eval2(P, top_at(kref));
    if (kref != 1) {
      rhoavg = rhoavg * (surfthick - g.zw[kref - 1]);
      for (int kt = 1; kt <= kref - 1; ++kt) rhoavg = rhoavg + g.dz[kt] * mwjf_eval2(P, top_at(kt));
      rhoavg = rhoavg / surfthick;
    }
    double dbs = 0.0, dbl = 0.0;
    if (rhok != 0.0) { dbs = GRAV * (1.0 - rhoavg / rhok); dbl = GRAV * (1.0 - rhokm / rhok); }
    if (k - 1 >= kmt) dbl = 0.0;
    DBSFC[o] = dbs;
    DBLOC[o - n2] = dbl;
    if (k == km) DBLOC[o] = 0.0;
  }
}

// ---- buoydiff + ri_iwmix + ddmix in one level-parallel launch ------------------------------------------------------
// k_kpp_buoydiff_lds continued through the interior coefficients: the wave that forms DBLOC(k-1) also forms the local
// Richardson number of that interface (velocity shear at the four surrounding U points, two levels) into a second LDS
// array; after a barrier the "carry below the bottom" rule, the 1-2-1 smoothing and the coefficients -- all local in
// k +- 1 -- are evaluated level-parallel from it.  T, S, U, V are read and DBLOC, DBSFC, VISC, VDC written once; the
// Richardson scratch field and the second pass over T, S, U, V of the column kernels are gone.  All loops stay rolled (a
// fully unrolled variant was 153 KB of code and ran at instruction-cache speed: 20 ms instead of 9).  Same operations
// in the same order per value as k_kpp_buoydiff_col + k_kpp_interior(_reg): bitwise equal (tested).  km <= 64.
// SFC = false: DBSFC is not formed here (k_kpp_bldepth_lazy evaluates it on demand, level by level, until the boundary-layer
// depth is found); the surface-layer levels are then not prepared and the rhoavg sums -- most of this kernel's arithmetic -- are gone.
template <int KR, int NG, bool SFC = true>
__global__ void __launch_bounds__(POP_COL_THREADS * NG)
k_kpp_buoy_interior_lds(DevGrid g, KppDev kp, const double *__restrict__ T, const double *__restrict__ S,
                        const double *__restrict__ U, const double *__restrict__ V, double *__restrict__ DBLOC,
                        double *__restrict__ DBSFC, double *__restrict__ VISC, double *__restrict__ VDC1, double *__restrict__ VDC2) {
  constexpr int KMR = 64, TOPR = (3 * KR > KMR) ? 3 * KR : KMR;
  __shared__ double shtop[TOPR][POP_COL_THREADS];   // prepared surface-layer levels; later the smoothing ping-pong buffer
  __shared__ double shri[KMR][POP_COL_THREADS];     // Richardson number of level k at row k - 1
  const int tx = threadIdx.x, gy = threadIdx.y;
  const int tile = g.xcd_remap ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const long long p2raw = (long long)tile * POP_COL_THREADS + tx;
  if (land_run(g, blockIdx.y, (long long)tile * POP_COL_THREADS, POP_COL_THREADS)) return;   // whole workgroup: before any barrier
  const bool live = p2raw < g.n2;
  const int p2 = live ? (int)p2raw : 0;
  const int b = blockIdx.y, km = g.km, nxb = g.nxb;
  const long long n2 = g.n2, base3 = (long long)b * g.n3 + p2;
  const int ci = p2 % nxb, cj = p2 / nxb;
  const bool edge = (ci == 0 || cj == 0);          // ugrid_to_tgrid zeroes the first row and column
  if (SFC) {
    for (int t = 1 + gy; t <= KR; t += NG) {
      const int kk = (t <= km) ? t : km;
      const long long o = base3 + (long long)(kk - 1) * n2;
      const MwjfTS2 x = mwjf_prep2(tmask(T[o]), S[o]);
      shtop[t - 1][tx] = x.TQ; shtop[KR + t - 1][tx] = x.SQ; shtop[2 * KR + t - 1][tx] = x.A2;
    }
    __syncthreads();
  }
  const int kmt = live ? g.KMT[(long long)b * n2 + p2] : 0;
  auto top_at = [&](int t) { MwjfTS2 x; x.TQ = shtop[t - 1][tx]; x.SQ = shtop[KR + t - 1][tx]; x.A2 = shtop[2 * KR + t - 1][tx]; return x; };
  const long long off4[4] = {0, -(long long)nxb, -1, -1 - (long long)nxb};
  if (SFC && live && gy == 0) DBSFC[base3] = 0.0;
#pragma unroll 1
  for (int k = 2 + gy; k <= km; k += NG) {
    double ri = 0.0;
    if (live) {
      const long long o = base3 + (long long)(k - 1) * n2;
      const double tk = T[o], sk = S[o], tm = T[o - n2], sm = S[o - n2];
      double sh4[4] = {0.0, 0.0, 0.0, 0.0};
      if (!edge) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const long long q = o + off4[t];
          const double du = U[q - n2] - U[q], dv = V[q - n2] - V[q];
          sh4[t] = du * du + dv * dv;
        }
      }
      const MwjfTS2 xk = mwjf_prep2(tmask(tk), sk), xkm = mwjf_prep2(tmask(tm), sm);
      const MwjfP P = mwjf_level(g.pressz[k]);
      const double rhokm = mwjf_eval2(P, xkm);
      const double rhok = mwjf_eval2(P, xk);
      double dbl = 0.0;
      if (SFC) {
        const double surfthick = KPP_EPSSFC * g.zt[k];
        const int kref = kp.kref[k];                 // wave-uniform (k is)
        double rhoavg = mwjf_eval2(P, top_at(kref));
        if (kref != 1) {
          rhoavg = rhoavg * (surfthick - g.zw[kref - 1]);
#pragma unroll 1
          for (int kt = 1; kt <= kref - 1; ++kt) rhoavg = rhoavg + g.dz[kt] * mwjf_eval2(P, top_at(kt));
          rhoavg = rhoavg / surfthick;
        }
        double dbs = 0.0;
        if (rhok != 0.0) dbs = GRAV * (1.0 - rhoavg / rhok);
        DBSFC[o] = dbs;
      }
      if (rhok != 0.0) dbl = GRAV * (1.0 - rhokm / rhok);
      if (k - 1 >= kmt) dbl = 0.0;
      DBLOC[o - n2] = dbl;
      if (k == km) DBLOC[o] = 0.0;
      double vsh = 0.0;
      if (!edge) vsh = 0.25 * sh4[0] + 0.25 * sh4[1] + 0.25 * sh4[2] + 0.25 * sh4[3];
      ri = dbl * (kp.zgrid[k - 1] - kp.zgrid[k]) / (vsh + KPP_EPS);
    }
    shri[k - 2][tx] = ri;                           // level k - 1
  }
  if (gy == 0) shri[km - 1][tx] = 0.0 * (kp.zgrid[km] - kp.zgrid[km + 1]) / (0.0 + KPP_EPS);   // DBLOC(km) = 0, no shear below
  __syncthreads();                                  // Richardson column complete; the surface-layer levels are dead
  // WORK0 of ri_iwmix: below the bottom the last ocean value is carried (0 on land)
  const double carry = (kmt >= 1) ? shri[kmt - 1][tx] : 0.0;
  auto w0 = [&](int k) { return (k <= kmt) ? shri[k - 1][tx] : carry; };
  // smoothing passes that are not the last write a full column into the other buffer (old values are never overwritten)
  double (*src)[POP_COL_THREADS] = nullptr;          // nullptr: read through w0()
  double (*dst)[POP_COL_THREADS] = shtop;
  for (int pass = 0; pass + 1 < kp.nsmooth; ++pass) {
#pragma unroll 1
    for (int k = 1 + gy; k <= km; k += NG) {
      const double cur = src ? src[k - 1][tx] : w0(k);
      double v = cur;
      if (kmt >= 3) {
        const double w1 = 0.25 * ((k > 1) ? (src ? src[k - 2][tx] : w0(k - 1)) : cur);
        const double nxt = (k < km) ? (src ? src[k][tx] : w0(k + 1)) : cur;
        v = w1 + 0.5 * cur + 0.25 * nxt;
      }
      dst[k - 1][tx] = v;
    }
    __syncthreads();
    src = dst; dst = (dst == shtop) ? shri : shtop;
  }
  if (!live) return;
  // last smoothing pass + coefficients (+ double diffusion), level by level
  const long long vb = ((long long)b * (km + 2)) * n2 + p2;
#pragma unroll 1
  for (int k = 1 + gy; k <= km; k += NG) {
    const long long o = base3 + (long long)(k - 1) * n2;
    const double cur = src ? src[k - 1][tx] : w0(k);
    double riw = cur;
    if (kp.nsmooth >= 1 && kmt >= 3) {
      const double w1 = 0.25 * ((k > 1) ? (src ? src[k - 2][tx] : w0(k - 1)) : cur);
      const double nxt = (k < km) ? (src ? src[k][tx] : w0(k + 1)) : cur;
      riw = w1 + 0.5 * cur + 0.25 * nxt;
    }
    double fri = fmax(riw, 0.0) / KPP_RIINFTY;
    fri = fmin(fri, 1.0);
    double visc, vd1 = 0.0, vd2 = 0.0;
    if (kp.lrich) {
      const double f = 1.0 - fri * fri;
      const double f3 = (f * f) * f;
      visc = kp.bckgrnd_vvc[k] + kp.rich_mix * f3;
      if (k < km) { vd2 = kp.bckgrnd_vdc[k] + kp.rich_mix * f3; vd1 = vd2; }
    } else {
      visc = kp.bckgrnd_vvc[k];
      if (k < km) { vd2 = kp.bckgrnd_vdc[k]; vd1 = vd2; }
    }
    if (k >= kmt) { visc = 0.0; vd1 = 0.0; vd2 = 0.0; }
    if (kp.ldbl_diff) {
      const double t_k = T[o], s_k = S[o];
      double ta_u = 0.0, sb_u = 0.0;
      { const MwjfP Pk = mwjf_level(g.pressz[k]); (void)mwjf_rho<true>(Pk, tmask(t_k), s_k, &ta_u, &sb_u); }
      double alphadt = 0.0, betads = 0.0, ta_n = 0.0, sb_n = 0.0, t_n = 0.0, s_n = 0.0;
      if (k < km) {
        t_n = T[o + n2]; s_n = S[o + n2];
        const MwjfP Pn = mwjf_level(g.pressz[k + 1]);
        (void)mwjf_rho<true>(Pn, tmask(t_n), s_n, &ta_n, &sb_n);
        alphadt = -0.5 * (ta_u + ta_n) * (t_k - t_n);
        betads = 0.5 * (sb_u + sb_n) * (s_k - s_n);
      }
      if (alphadt > betads && betads > 0.0) {
        const double rrho = fmin(alphadt / betads, KPP_RRHO0);
        const double f = 1.0 - (rrho - 1.0) / (KPP_RRHO0 - 1.0);
        const double diffdd = KPP_DSFMAX * ((f * f) * f);
        vd1 = vd1 + 0.7 * diffdd; vd2 = vd2 + diffdd;
      }
      double rrho = 0.0, diffdd = 0.0, prandtl = 0.0;
      if (alphadt < 0.0 && betads < 0.0 && alphadt > betads) {
        rrho = alphadt / betads;
        diffdd = 1.5e-2 * 0.909 * exp(4.6 * exp(-0.54 * (1.0 / rrho - 1.0)));
        prandtl = 0.15 * rrho;
      }
      if (rrho > 0.5) prandtl = (1.85 - 0.85 / rrho) * rrho;
      vd1 = vd1 + diffdd; vd2 = vd2 + prandtl * diffdd;
    }
    VISC[o] = visc;
    VDC1[vb + (long long)k * n2] = vd1;
    if (!kp.vdc_same) VDC2[vb + (long long)k * n2] = vd2;   // one array when the two classes share their values (no double diffusion)
  }
}

// ---- buoydiff (DBLOC) + ri_iwmix as ONE column march, no LDS, no barrier (round 3) ------------------------------------
// With the surface-layer buoyancy difference formed on demand by the boundary-layer march (SFC = false above) every level
// costs the same, and nothing is left that needs the level-parallel split of k_kpp_buoy_interior_lds: a thread owns a column
// and walks down it once.  What it carries between levels is what the level-parallel form recomputes or re-reads -- the
// prepared (T, S) of the level above (one square root and half an equation of state per cell), U and V of the level above at the
// four surrounding U points (8 of the 16 velocity loads), and a three-value window of the local Richardson number for
// the 1-2-1 smoothing (the LDS column and its three barriers per workgroup).  The pressure polynomials of the equation of state
// come from a per-level table (KppDev::eosP, formed on the device by k_kpp_level_table with mwjf_level itself) through scalar
// loads instead of 14 vector operations per level.  The raw operands of level k + 2 are in flight while level k + 1 is
// evaluated.  One smoothing pass (num_v_smooth_Ri = 1, the reference's default), no double diffusion, no partial bottom
// cells: every other configuration keeps the kernels above.  Same operations in the same order per value: bitwise equal
// to k_kpp_buoy_interior_lds (tested).
__global__ void k_kpp_level_table(DevGrid g, double *__restrict__ tab) {
  const int k = threadIdx.x;
  if (k < 1 || k > g.km) return;
  const MwjfP P = mwjf_level(g.pressz[k]);
  double *t = tab + 6 * k;
  t[0] = P.n0; t[1] = P.n2; t[2] = P.ns1t0; t[3] = P.d0; t[4] = P.d1; t[5] = P.d3;
}
struct KppRaw { double t, s, u[4], v[4]; };
// PBC: partial bottom cells (vmix_kpp.F90:1531-1533, 1553-1563): the shear of each U corner over its own thickness, the Richardson
// number over the T column's own thickness
template <bool PBC = false>
__global__ void __launch_bounds__(POP_COL_THREADS)
k_kpp_buoy_interior_march(DevGrid g, KppDev kp, const double *__restrict__ T, const double *__restrict__ S,
                          const double *__restrict__ U, const double *__restrict__ V, double *__restrict__ DBLOC,
                          double *__restrict__ VISC, double *__restrict__ VDC1, double *__restrict__ VDC2) {
  Col c;
  if (!col_setup(g, c, false)) return;
  const int km = g.km, nxb = g.nxb;
  const long long n2 = g.n2;
  const int kmt = g.KMT[c.q2];
  const bool edge = (c.i == 0 || c.j == 0);          // ugrid_to_tgrid zeroes the first row and column (their loads go to the own cell)
  const long long off4[4] = {0, edge ? 0 : -(long long)nxb, edge ? 0 : -1, edge ? 0 : -1 - (long long)nxb};
  const long long vb = ((long long)c.b * (km + 2)) * n2 + c.p2;
  int kmu4[4] = {0, 0, 0, 0}; double dzub4[4] = {0.0, 0.0, 0.0, 0.0}; const double dzbc = PBC ? g.DZBC[c.q2] : 0.0;
  if (PBC) {
#pragma unroll
    for (int t = 0; t < 4; ++t) { kmu4[t] = g.KMU[c.q2 + off4[t]]; dzub4[t] = g.DZUB[c.q2 + off4[t]]; }
  }
  auto dzq = [&](int k, int kbot, double dzbot) { return (k < 1 || k > km) ? 0.0 : ((k == kbot) ? dzbot : g.dz.u(k)); };   // pbc_dz, scalar loads
  auto load = [&](int k) {                            // level k = 1 .. km
    KppRaw r;
    const long long o = c.base3 + (long long)(k - 1) * n2;
    r.t = T[o]; r.s = S[o];
#pragma unroll
    for (int t = 0; t < 4; ++t) { r.u[t] = U[o + off4[t]]; r.v[t] = V[o + off4[t]]; }
    return r;
  };
  // coefficients of level k from the window (w0(k-1), w0(k), w0(k+1))
  auto emit = [&](int k, double wprev, double cur, double wnext) {
    double riw = cur;
    if (kmt >= 3) {
      const double w1 = 0.25 * ((k > 1) ? wprev : cur);
      const double nxt = (k < km) ? wnext : cur;
      riw = w1 + 0.5 * cur + 0.25 * nxt;
    }
    double fri = fmax(riw, 0.0) / KPP_RIINFTY;
    fri = fmin(fri, 1.0);
    double visc, vd1 = 0.0, vd2 = 0.0;
    if (kp.lrich) {
      const double f = 1.0 - fri * fri;
      const double f3 = (f * f) * f;
      visc = kp.bckgrnd_vvc[k] + kp.rich_mix * f3;
      if (k < km) { vd2 = kp.bckgrnd_vdc[k] + kp.rich_mix * f3; vd1 = vd2; }
    } else {
      visc = kp.bckgrnd_vvc[k];
      if (k < km) { vd2 = kp.bckgrnd_vdc[k]; vd1 = vd2; }
    }
    if (k >= kmt) { visc = 0.0; vd1 = 0.0; vd2 = 0.0; }
    VISC[c.base3 + (long long)(k - 1) * n2] = visc;
    VDC1[vb + (long long)k * n2] = vd1;
    if (!kp.vdc_same) VDC2[vb + (long long)k * n2] = vd2;
  };
  KppRaw up = load(1), A = load(2 <= km ? 2 : km), B;
  MwjfTS2 xkm = mwjf_prep2(tmask(up.t), up.s);
  double wpp = 0.0, wp = 0.0, carry = 0.0;            // w0(k-3), w0(k-2); the value carried below the bottom
  unsigned long long convb = 0ull;                    // KppDev::CONVB of this column
  // level k: forms DBLOC(k-1), Ri(k-1) and the coefficients of level k-2.  cu holds the operands of level k; those of level k+1
  // are requested into nx first and are in flight while this level is evaluated.  The loop below alternates two operand sets
  // (a rotation by register moves would have to wait for the loads it moves)
  auto level = [&](int k, const KppRaw &cu, KppRaw &nx) {
    nx = load(k + 1 <= km ? k + 1 : km);
    const MwjfTS2 xk = mwjf_prep2(tmask(cu.t), cu.s);
    MwjfP P;
    { const int e = 6 * k; P.n0 = kp.eosP[e]; P.n2 = kp.eosP[e + 1]; P.ns1t0 = kp.eosP[e + 2]; P.d0 = kp.eosP[e + 3]; P.d1 = kp.eosP[e + 4]; P.d3 = kp.eosP[e + 5]; }
    const double rhokm = mwjf_eval2(P, xkm);
    const double rhok = mwjf_eval2(P, xk);
    double dbl = 0.0;
    if (rhok != 0.0) dbl = GRAV * (1.0 - rhokm / rhok);
    if (k - 1 >= kmt) dbl = 0.0;
    DBLOC[c.base3 + (long long)(k - 2) * n2] = dbl;
    if (!(dbl > 0.0)) convb |= 1ull << (k - 2);
    double sh4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double du = up.u[t] - cu.u[t], dv = up.v[t] - cu.v[t];
      sh4[t] = du * du + dv * dv;
      if (PBC) { const double h = 0.5 * (dzq(k - 1, kmu4[t], dzub4[t]) + dzq(k, kmu4[t], dzub4[t])); sh4[t] = sh4[t] / (h * h); }
    }
    double vsh = 0.25 * sh4[0] + 0.25 * sh4[1] + 0.25 * sh4[2] + 0.25 * sh4[3];
    if (edge) vsh = 0.0;
    double ri = dbl * (kp.zgrid[k - 1] - kp.zgrid[k]) / (vsh + KPP_EPS);
    if (PBC) { const double h = 0.5 * (dzq(k - 1, kmt, dzbc) + dzq(k, kmt, dzbc)); ri = dbl / (vsh + KPP_EPS / (h * h)) / h; }
    const int m = k - 1;                              // the level ri belongs to
    const double w = (m <= kmt) ? ri : carry;
    if (m == kmt) carry = ri;
    if (m >= 2) emit(m - 1, wpp, wp, w);
    wpp = wp; wp = w;
#pragma unroll
    for (int t = 0; t < 4; ++t) { up.u[t] = cu.u[t]; up.v[t] = cu.v[t]; }
    xkm = xk;
  };
  int k = 2;
#pragma unroll 1
  for (; k + 1 <= km; k += 2) { level(k, A, B); level(k + 1, B, A); }
  if (k <= km) level(k, A, B);
  DBLOC[c.base3 + (long long)(km - 1) * n2] = 0.0;
  if (kp.CONVB) kp.CONVB[c.q2] = convb;
  {
    double ri = 0.0 * (kp.zgrid[km] - kp.zgrid[km + 1]) / (0.0 + KPP_EPS);   // DBLOC(km) = 0, no shear below
    if (PBC) { const double h = 0.5 * dzq(km, kmt, dzbc); ri = 0.0 / (0.0 + KPP_EPS / (h * h)) / h; }
    const double w = (km <= kmt) ? ri : carry;
    emit(km - 1, wpp, wp, w);
    emit(km, wp, w, w);
  }
}

// ---- bldepth part 1, column form: U, V of the top KR levels in registers, each level read once ----
template <int KR, bool PBC = false>
__global__ void __launch_bounds__(POP_COL_THREADS)
k_kpp_ushear_col(DevGrid g, KppDev kp, const double *__restrict__ U, const double *__restrict__ V, double *__restrict__ WU) {
  // the two barriers below follow an early return of some lanes: defined only because the workgroup is exactly one wavefront
  // (launched with dim3(POP_COL_THREADS), never as a 2-D tile)
  static_assert(POP_COL_THREADS == 64, "k_kpp_ushear_col: one wavefront per workgroup");
  __shared__ int s_cap;
  if (kp.WUK && threadIdx.x == 0) s_cap = 0;
  Col c;
  if (!col_setup(g, c, false)) return;
  const int km = g.km;
  const long long n2 = g.n2;
  // With a hint (kp.WUK): the march of k_kpp_bldepth<true> stops, wave by wave, at the deepest first crossing of its columns, which
  // moves little from one evaluation to the next.  This wave forms WU down to the deepest KBL the previous evaluation left at the
  // T cells that use its U points, plus a margin, and records that level per column; the march computes what is missing itself.
  int kcap = km, krcap = KR;
  if (kp.WUK) {
    int h = kp.KBL[c.q2];
    if (c.i + 1 < g.nxb) h = max(h, kp.KBL[c.q2 + 1]);
    if (c.j + 1 < g.nyb) { h = max(h, kp.KBL[c.q2 + g.nxb]); if (c.i + 1 < g.nxb) h = max(h, kp.KBL[c.q2 + g.nxb + 1]); }
    __syncthreads();                               // one wave per workgroup: s_cap = 0 is visible
    atomicMax(&s_cap, h);
    __syncthreads();
    kcap = min(km, max(2, s_cap + kp.wu_margin));
    kp.WUK[c.q2] = kcap;
    krcap = kp.kref[kcap];                         // kref is non-decreasing in the level
  }
  const int kmu_c = PBC ? g.KMU[c.q2] : 0; const double dzub_c = PBC ? g.DZUB[c.q2] : 0.0;
  double ur[KR + 1], vr[KR + 1];
#pragma unroll
  for (int t = 1; t <= KR; ++t) {
    ur[t] = 0.0; vr[t] = 0.0;
    if (t <= krcap) {
      const int kk = (t <= km) ? t : km;
      const long long o = c.base3 + (long long)(kk - 1) * n2;
      ur[t] = U[o]; vr[t] = V[o];
    }
  }
  for (int kl = 2; kl <= kcap; ++kl) {
    const long long o = c.base3 + (long long)(kl - 1) * n2;
    const double ukl = U[o], vkl = V[o];
    const double surfthick = KPP_EPSSFC * g.zt[kl];
    const int kref = kp.kref[kl];
    double uref, vref;
    if (kref > 1) {
      double uk = ur[1], vk = vr[1];
#pragma unroll
      for (int t = 2; t <= KR; ++t) if (t == kref) { uk = ur[t]; vk = vr[t]; }
      uref = uk * (surfthick - g.zw[kref - 1]);
      vref = vk * (surfthick - g.zw[kref - 1]);
#pragma unroll
      for (int kt = 1; kt <= KR - 1; ++kt)
        if (kt <= kref - 1) { uref = uref + g.dz[kt] * ur[kt]; vref = vref + g.dz[kt] * vr[kt]; }
      uref = uref / surfthick; vref = vref / surfthick;
    } else { uref = ur[1]; vref = vr[1]; }
    const double du = uref - ukl, dv = vref - vkl;
    if (PBC) {   // vmix_kpp.F90:2359-2362
      const double h = -kp.zgrid[kl - 1] + 0.5 * (pbc_dz(g, kl, kmu_c, dzub_c) + pbc_dz(g, kl - 1, kmu_c, dzub_c) - pbc_dz(g, 1, kmu_c, dzub_c));
      WU[o] = (du * du + dv * dv) / (h * h);
    } else WU[o] = du * du + dv * dv;
  }
}

// ---- ri_iwmix + ddmix: interior coefficients -----------------------------------------------
// VISC: (nxb,nyb,km,block) scratch; VDC1/VDC2: (nxb,nyb,0:km+1,block), levels 0 and km+1 stay 0.
// RIW: scratch for the (smoothed) Richardson number.
// PBC: partial bottom cells (vmix_kpp.F90:1531-1533, 1553-1563)
template <bool PBC = false>
__global__ void __launch_bounds__(POP_COL_THREADS)
k_kpp_interior(DevGrid g, KppDev kp, const double *__restrict__ T, const double *__restrict__ S,
               const double *__restrict__ U, const double *__restrict__ V, const double *__restrict__ DBLOC,
               double *__restrict__ RIW, double *__restrict__ VISC, double *__restrict__ VDC1, double *__restrict__ VDC2) {
  Col c;
  if (!col_setup(g, c, false)) return;
  const int km = g.km, nxb = g.nxb;
  const long long n2 = g.n2;
  const int kmt = g.KMT[c.q2];
  const bool edge = (c.i == 0 || c.j == 0);   // ugrid_to_tgrid zeroes the first row and column
  // PBC: bottom level / thickness of the four U cells around the T cell (offsets 0, -nxb, -1, -1-nxb) and of the T cell
  int kmu4[4] = {0, 0, 0, 0}; double dzub4[4] = {0, 0, 0, 0}; const double dzbc = PBC ? g.DZBC[c.q2] : 0.0;
  if (PBC && !edge) {
    const long long off[4] = {0, -nxb, -1, -1 - nxb};
#pragma unroll
    for (int t = 0; t < 4; ++t) { kmu4[t] = g.KMU[c.q2 + off[t]]; dzub4[t] = g.DZUB[c.q2 + off[t]]; }
  }
  // pass 1: local Richardson number
  double prev = 0.0;   // WORK0(k-1)
  for (int k = 1; k <= km; ++k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    double vsh = 0.0;
    if (k < km && !edge) {
      auto sh = [&](long long q, int t) {
        const double du = U[q] - U[q + n2], dv = V[q] - V[q + n2];
        if (PBC) { const double h = 0.5 * (pbc_dz(g, k, kmu4[t], dzub4[t]) + pbc_dz(g, k + 1, kmu4[t], dzub4[t])); return (du * du + dv * dv) / (h * h); }
        return du * du + dv * dv;
      };
      vsh = 0.25 * sh(o, 0) + 0.25 * sh(o - nxb, 1) + 0.25 * sh(o - 1, 2) + 0.25 * sh(o - 1 - nxb, 3);
    }
    double ri = DBLOC[o] * (kp.zgrid[k] - kp.zgrid[k + 1]) / (vsh + KPP_EPS);
    if (PBC) {
      const double h = (k < km) ? 0.5 * (pbc_dz(g, k, kmt, dzbc) + pbc_dz(g, k + 1, kmt, dzbc)) : 0.5 * pbc_dz(g, k, kmt, dzbc);
      ri = DBLOC[o] / (vsh + KPP_EPS / (h * h)) / h;
    }
    const double w0 = (k <= kmt) ? ri : prev;
    RIW[o] = w0;
    prev = w0;
  }
  // pass 2: 1-2-1 vertical smoothing, nsmooth times (in place, old values carried)
  for (int n = 0; n < kp.nsmooth; ++n) {
    double w1 = 0.25 * RIW[c.base3];
    if (kmt >= 3) {
      double cur = RIW[c.base3];
      for (int k = 1; k <= km; ++k) {
        const long long o = c.base3 + (long long)(k - 1) * n2;
        const double nxt = (k < km) ? RIW[o + n2] : cur;    // WORK0(km+1) = WORK0(km) (old value)
        RIW[o] = w1 + 0.5 * cur + 0.25 * nxt;
        w1 = 0.25 * cur;
        cur = nxt;
      }
    }
  }
  // pass 3: coefficients (+ double diffusion)
  const long long vb = ((long long)c.b * (km + 2)) * n2 + c.p2;
  double ta_u = 0.0, sb_u = 0.0, t_k = T[c.base3], s_k = S[c.base3];
  if (kp.ldbl_diff) { const MwjfP P = mwjf_level(g.pressz[1]); (void)mwjf_rho<true>(P, tmask(t_k), s_k, &ta_u, &sb_u); }
  for (int k = 1; k <= km; ++k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    double fri = fmax(RIW[o], 0.0) / KPP_RIINFTY;
    fri = fmin(fri, 1.0);
    double visc, vd1 = 0.0, vd2 = 0.0;
    if (kp.lrich) {
      const double f = 1.0 - fri * fri;
      const double f3 = (f * f) * f;
      visc = kp.bckgrnd_vvc[k] + kp.rich_mix * f3;
      if (k < km) { vd2 = kp.bckgrnd_vdc[k] + kp.rich_mix * f3; vd1 = vd2; }
    } else {
      visc = kp.bckgrnd_vvc[k];
      if (k < km) { vd2 = kp.bckgrnd_vdc[k]; vd1 = vd2; }
    }
    if (k >= kmt) { visc = 0.0; vd1 = 0.0; vd2 = 0.0; }
    if (kp.ldbl_diff) {
      double alphadt = 0.0, betads = 0.0, ta_n = 0.0, sb_n = 0.0, t_n = 0.0, s_n = 0.0;
      if (k < km) {
        t_n = T[o + n2]; s_n = S[o + n2];
        const MwjfP P = mwjf_level(g.pressz[k + 1]);
        (void)mwjf_rho<true>(P, tmask(t_n), s_n, &ta_n, &sb_n);
        alphadt = -0.5 * (ta_u + ta_n) * (t_k - t_n);
        betads = 0.5 * (sb_u + sb_n) * (s_k - s_n);
      }
      if (alphadt > betads && betads > 0.0) {
        const double rrho = fmin(alphadt / betads, KPP_RRHO0);
        const double f = 1.0 - (rrho - 1.0) / (KPP_RRHO0 - 1.0);
        const double diffdd = KPP_DSFMAX * ((f * f) * f);
        vd1 = vd1 + 0.7 * diffdd; vd2 = vd2 + diffdd;
      }
      double rrho = 0.0, diffdd = 0.0, prandtl = 0.0;
      if (alphadt < 0.0 && betads < 0.0 && alphadt > betads) {
        rrho = alphadt / betads;
        diffdd = 1.5e-2 * 0.909 * exp(4.6 * exp(-0.54 * (1.0 / rrho - 1.0)));
        prandtl = 0.15 * rrho;
      }
      if (rrho > 0.5) prandtl = (1.85 - 0.85 / rrho) * rrho;
      vd1 = vd1 + diffdd; vd2 = vd2 + prandtl * diffdd;
      ta_u = ta_n; sb_u = sb_n; t_k = t_n; s_k = s_n;
    }
    VISC[o] = visc;
    VDC1[vb + (long long)k * n2] = vd1;
    if (!kp.vdc_same) VDC2[vb + (long long)k * n2] = vd2;   // one array when the two classes share their values (no double diffusion)
  }
}

// ---- ri_iwmix + ddmix with the Richardson-number column in registers (km = 60, 62) -------------------
// Same operations as k_kpp_interior; the three passes over the column (local Ri, 1-2-1 smoothing, coefficients)
// keep Ri in a fully unrolled register array instead of a scratch field, and the velocity differences carry
// level k+1 into the next iteration: one read of each input, no RIW traffic.
template <int KM>
__global__ void __launch_bounds__(POP_COL_THREADS)
k_kpp_interior_reg(DevGrid g, KppDev kp, const double *__restrict__ T, const double *__restrict__ S,
                   const double *__restrict__ U, const double *__restrict__ V, const double *__restrict__ DBLOC,
                   double *__restrict__ VISC, double *__restrict__ VDC1, double *__restrict__ VDC2) {
  Col c;
  if (!col_setup(g, c, false)) return;
  const int nxb = g.nxb;
  const long long n2 = g.n2;
  const int kmt = g.KMT[c.q2];
  const bool edge = (c.i == 0 || c.j == 0);
  double riw[KM];
  {
    // pass 1: local Richardson number; U, V at the four surrounding U points, level k carried from k+1
    const long long off4[4] = {0, -(long long)nxb, -1, -1 - (long long)nxb};
    double uk[4] = {0, 0, 0, 0}, vk[4] = {0, 0, 0, 0};
    if (!edge) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { uk[t] = U[c.base3 + off4[t]]; vk[t] = V[c.base3 + off4[t]]; }
    }
    double prev = 0.0;
#pragma unroll
    for (int k = 1; k <= KM; ++k) {
      const long long o = c.base3 + (long long)(k - 1) * n2;
      double vsh = 0.0;
      if (k < KM && !edge) {
        double sh[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double un = U[o + off4[t] + n2], vn = V[o + off4[t] + n2];
          const double du = uk[t] - un, dv = vk[t] - vn;
          sh[t] = du * du + dv * dv;
          uk[t] = un; vk[t] = vn;
        }
        vsh = 0.25 * sh[0] + 0.25 * sh[1] + 0.25 * sh[2] + 0.25 * sh[3];
      }
      const double ri = DBLOC[o] * (kp.zgrid[k] - kp.zgrid[k + 1]) / (vsh + KPP_EPS);
      const double w0 = (k <= kmt) ? ri : prev;
      riw[k - 1] = w0;
      prev = w0;
    }
  }
  // pass 2: 1-2-1 vertical smoothing, nsmooth times (old values carried)
  for (int n = 0; n < kp.nsmooth; ++n) {
    if (kmt >= 3) {
      double w1 = 0.25 * riw[0];
      double cur = riw[0];
#pragma unroll
      for (int k = 1; k <= KM; ++k) {
        const double nxt = (k < KM) ? riw[k] : cur;
        riw[k - 1] = w1 + 0.5 * cur + 0.25 * nxt;
        w1 = 0.25 * cur;
        cur = nxt;
      }
    }
  }
  // pass 3: coefficients (+ double diffusion)
  const long long vb = ((long long)c.b * (KM + 2)) * n2 + c.p2;
  double ta_u = 0.0, sb_u = 0.0, t_k = T[c.base3], s_k = S[c.base3];
  if (kp.ldbl_diff) { const MwjfP P = mwjf_level(g.pressz[1]); (void)mwjf_rho<true>(P, tmask(t_k), s_k, &ta_u, &sb_u); }
#pragma unroll
  for (int k = 1; k <= KM; ++k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    double fri = fmax(riw[k - 1], 0.0) / KPP_RIINFTY;
    fri = fmin(fri, 1.0);
    double visc, vd1 = 0.0, vd2 = 0.0;
    if (kp.lrich) {
      const double f = 1.0 - fri * fri;
      const double f3 = (f * f) * f;
      visc = kp.bckgrnd_vvc[k] + kp.rich_mix * f3;
      if (k < KM) { vd2 = kp.bckgrnd_vdc[k] + kp.rich_mix * f3; vd1 = vd2; }
    } else {
      visc = kp.bckgrnd_vvc[k];
      if (k < KM) { vd2 = kp.bckgrnd_vdc[k]; vd1 = vd2; }
    }
    if (k >= kmt) { visc = 0.0; vd1 = 0.0; vd2 = 0.0; }
    if (kp.ldbl_diff) {
      double alphadt = 0.0, betads = 0.0, ta_n = 0.0, sb_n = 0.0, t_n = 0.0, s_n = 0.0;
      if (k < KM) {
        t_n = T[o + n2]; s_n = S[o + n2];
        const MwjfP P = mwjf_level(g.pressz[k + 1]);
        (void)mwjf_rho<true>(P, tmask(t_n), s_n, &ta_n, &sb_n);
        alphadt = -0.5 * (ta_u + ta_n) * (t_k - t_n);
        betads = 0.5 * (sb_u + sb_n) * (s_k - s_n);
      }
      if (alphadt > betads && betads > 0.0) {
        const double rrho = fmin(alphadt / betads, KPP_RRHO0);
        const double f = 1.0 - (rrho - 1.0) / (KPP_RRHO0 - 1.0);
        const double diffdd = KPP_DSFMAX * ((f * f) * f);
        vd1 = vd1 + 0.7 * diffdd; vd2 = vd2 + diffdd;
      }
      double rrho = 0.0, diffdd = 0.0, prandtl = 0.0;
      if (alphadt < 0.0 && betads < 0.0 && alphadt > betads) {
        rrho = alphadt / betads;
        diffdd = 1.5e-2 * 0.909 * exp(4.6 * exp(-0.54 * (1.0 / rrho - 1.0)));
        prandtl = 0.15 * rrho;
      }
      if (rrho > 0.5) prandtl = (1.85 - 0.85 / rrho) * rrho;
      vd1 = vd1 + diffdd; vd2 = vd2 + prandtl * diffdd;
      ta_u = ta_n; sb_u = sb_n; t_k = t_n; s_k = s_n;
    }
    VISC[o] = visc;
    VDC1[vb + (long long)k * n2] = vd1;
    if (!kp.vdc_same) VDC2[vb + (long long)k * n2] = vd2;   // one array when the two classes share their values (no double diffusion)
  }
}

// ---- bldepth, part 1: shear^2 between the surface-layer reference velocity and level kl at U
// points; 3-D parallel, one thread per (i,j,kl)
// PBC: partial bottom cells (vmix_kpp.F90:2359-2362): divided by the squared distance to the reference level
template <bool PBC = false>
__global__ void __launch_bounds__(256)
k_kpp_ushear(DevGrid g, KppDev kp, const double *__restrict__ U, const double *__restrict__ V, double *__restrict__ WU) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int kl = blockIdx.y + 1, b = blockIdx.z;
  if (p2 >= g.n2 || kl < 2) return;
  const long long n2 = g.n2;
  const long long base3 = (long long)b * g.n3 + p2;
  const long long o = base3 + (long long)(kl - 1) * n2;
  const double surfthick = KPP_EPSSFC * g.zt[kl];
  const int kref = kp.kref[kl];
  double uref, vref;
  if (kref > 1) {
    const long long orf = base3 + (long long)(kref - 1) * n2;
    uref = U[orf] * (surfthick - g.zw[kref - 1]);
    vref = V[orf] * (surfthick - g.zw[kref - 1]);
    // same left-to-right sum; the loads of 8 levels are issued before their adds (the chain otherwise
    // pays one L2 round trip per level)
    int kt = 1;
    for (; kt + 7 <= kref - 1; kt += 8) {
      double uu[8], vv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) { const long long ot = base3 + (long long)(kt + t - 1) * n2; uu[t] = U[ot]; vv[t] = V[ot]; }
#pragma unroll
      for (int t = 0; t < 8; ++t) { uref = uref + g.dz[kt + t] * uu[t]; vref = vref + g.dz[kt + t] * vv[t]; }
    }
    for (; kt <= kref - 1; ++kt) {
      const long long ot = base3 + (long long)(kt - 1) * n2;
      uref = uref + g.dz[kt] * U[ot];
      vref = vref + g.dz[kt] * V[ot];
    }
    uref = uref / surfthick; vref = vref / surfthick;
  } else { uref = U[base3]; vref = V[base3]; }
  const double du = uref - U[o], dv = vref - V[o];
  if (PBC) {
    const long long q2 = (long long)b * n2 + p2;
    const int kmu = g.KMU[q2]; const double dzub = g.DZUB[q2];
    const double h = -kp.zgrid[kl - 1] + 0.5 * (pbc_dz(g, kl, kmu, dzub) + pbc_dz(g, kl - 1, kmu, dzub) - pbc_dz(g, 1, kmu, dzub));
    WU[o] = (du * du + dv * dv) / (h * h);
  } else
  WU[o] = du * du + dv * dv;
}

// the same for one U point (column base3) and level: what k_kpp_ushear / k_kpp_ushear_col store in WU
// PBC: divided by the squared distance to the reference level with the U column's own thicknesses (vmix_kpp.F90:2359-2362); q2 = the
// U point's 2-D index
template <bool PBC = false>
__device__ __forceinline__ double kpp_ushear_point(const DevGrid &g, const KppDev &kp, const double *__restrict__ U, const double *__restrict__ V,
                                                   long long base3, int kl, long long q2 = 0) {
  const long long n2 = g.n2;
  const long long o = base3 + (long long)(kl - 1) * n2;
  const double surfthick = KPP_EPSSFC * g.zt[kl];
  const int kref = kp.kref[kl];
  double uref, vref;
  if (kref > 1) {
    const long long orf = base3 + (long long)(kref - 1) * n2;
    uref = U[orf] * (surfthick - g.zw[kref - 1]);
    vref = V[orf] * (surfthick - g.zw[kref - 1]);
#pragma unroll 1
    for (int kt = 1; kt <= kref - 1; ++kt) {
      const long long ot = base3 + (long long)(kt - 1) * n2;
      uref = uref + g.dz[kt] * U[ot];
      vref = vref + g.dz[kt] * V[ot];
    }
    uref = uref / surfthick; vref = vref / surfthick;
  } else { uref = U[base3]; vref = V[base3]; }
  const double du = uref - U[o], dv = vref - V[o];
  if (PBC) {
    const int kmu = g.KMU[q2]; const double dzub = g.DZUB[q2];
    const double h = -kp.zgrid[kl - 1] + 0.5 * (pbc_dz(g, kl, kmu, dzub) + pbc_dz(g, kl - 1, kmu, dzub) - pbc_dz(g, 1, kmu, dzub));
    return (du * du + dv * dv) / (h * h);
  }
  return du * du + dv * dv;
}

// ---- bldepth, part 2: bulk Richardson number march -> unsmoothed HBLT, KBL ---------------------
// LAZY: the buoyancy difference against the surface layer (DBSFC of buoydiff: up to 1 + kref equation-of-state evaluations per
// level, the bulk of KPP's arithmetic) is formed here, on demand, for the level the march is at -- same operations in the same
// order as k_kpp_buoy_interior_lds, from surface-layer levels prepared into LDS as kref grows -- and the wave leaves the march
// once none of its 64 columns can change any more: a column's (HBLT, KBL) are final after its first crossing of the critical bulk
// Richardson number, and no crossing happens below its bottom (wk = 0 there), so the levels that are skipped only rotated
// rib_* / z_*.  BFSFC ("value of the last pass") is formed for kl = km directly.  Results are bitwise those of the full march
// (tested); the reference's array form (vmix_kpp.F90 bldepth :2280-2520) has no such exit.  Not with lcheckekmo (its Ekman / Monin-
// Obukhov limits march every level) and not with the mixed-layer-depth diagnostics (they read DBSFC at every level).
// PBC: partial bottom cells (vmix_kpp.F90:2212-2220, 2363-2366, 2486-2496, 2561-2575), with either form of the march (r3)
template <bool LAZY, int KR, bool PBC = false>
__global__ void __launch_bounds__(POP_COL_THREADS)
k_kpp_bldepth(DevGrid g, KppDev kp, const double *__restrict__ T, const double *__restrict__ S,
              const double *__restrict__ STF1, const double *__restrict__ STF2, const double *__restrict__ DBLOC,
              const double *__restrict__ DBSFC, const double *__restrict__ WU, const double *__restrict__ UU, const double *__restrict__ VV) {
  static_assert(POP_COL_THREADS == 64, "the lazy march votes over one 64-lane wave");
  __shared__ double shtop[LAZY ? 3 * KR : 1][POP_COL_THREADS];
  int nprep = 0;                                   // surface-layer levels prepared so far (wave-uniform)
  Col c;
  if (!col_setup(g, c, false)) return;
  const int km = g.km, nxb = g.nxb;
  const long long n2 = g.n2;
  const int kmt = g.KMT[c.q2];
  const bool edge = (c.i == 0 || c.j == 0);
  int wuk[4] = {km, km, km, km};                  // valid levels of WU at the four U points around this T point
  if (LAZY && kp.WUK && !edge) { wuk[0] = kp.WUK[c.q2]; wuk[1] = kp.WUK[c.q2 - 1]; wuk[2] = kp.WUK[c.q2 - nxb]; wuk[3] = kp.WUK[c.q2 - 1 - nxb]; }
  const double s1 = g.SMFT1[c.q2], s2 = g.SMFT2[c.q2];
  double ustar = sqrt(sqrt(s1 * s1 + s2 * s2));
  ustar = fmax(ustar, KPP_EPS);
  double talpha, sbeta;
  const MwjfP P1 = mwjf_level(g.pressz[1]);
  const double rho1 = mwjf_rho<true>(P1, tmask(T[c.base3]), S[c.base3], &talpha, &sbeta);
  double bo = 0.0, bosol = 0.0;
  const int chli = (kp.lshort_wave && kp.sw_type == 2) ? kp.CHLI[c.q2] : 0;
  if (rho1 != 0.0) {
    bo = GRAV * (-talpha * STF1[c.q2] - sbeta * STF2[c.q2]) / rho1;
    if (kp.lshort_wave) bosol = -GRAV * talpha * kp.SHF_QSW[c.q2] / rho1;
  }
  int kbl = (kmt > 1) ? kmt : 1;
  double hblt = -kp.zgrid[kbl];
  const double dzbc = PBC ? g.DZBC[c.q2] : 0.0, dzub = PBC ? g.DZUB[c.q2] : 0.0;
  const int kmu_c = PBC ? g.KMU[c.q2] : 0;
  // ZKL of level kl: -zgrid(kl), or with partial bottom cells -zgrid(kl-1) + p5*(DZT(kl) + DZT(kl-1)) for kl > 1
  auto zkl_of = [&](int kl) { return (PBC && kl > 1) ? -kp.zgrid[kl - 1] + 0.5 * (pbc_dz(g, kl, kmt, dzbc) + pbc_dz(g, kl - 1, kmt, dzbc)) : -kp.zgrid[kl]; };
  if (PBC) hblt = zkl_of(kbl);
  double rib_upper = 0.0, rib_up = 0.0, z_upper = 0.0, z_up = kp.zgrid[1];
  double bfsfc = bo;
  // lcheckekmo (:2231-2265): Ekman and Monin-Obukhov depth limits under stable forcing; hmon_up / hmon_dn rotate like rib_*
  double hekman = 0.0, hlimit = 0.0, hmon_up = 0.0;
  if (kp.lcheckekmo) {
    hekman = -kp.zgrid[km] + KPP_EPS; hlimit = -kp.zgrid[km] + KPP_EPS;
    double bf = kpp_bfsfc(kp, bo, bosol, -z_up, 1, chli);
    const double st = (bf >= 0.0) ? 1.0 : 0.0;
    bf = bf + st * KPP_EPS;
    const double w = st * KPP_CMONOB * ustar * ustar * ustar / KPP_VONKAR / bf + (st - 1.0) * kp.zgrid[km];
    hmon_up = (w <= -z_up) ? -z_up + KPP_EPS : w;
  }
  for (int kl = 2; kl <= km; ++kl) {
    if (LAZY && __builtin_amdgcn_ballot_w64(kbl == kmt && kl <= kmt) == 0) break;
    const long long o = c.base3 + (long long)(kl - 1) * n2;
    const double surfthick = KPP_EPSSFC * g.zt[kl];
    const double zkl = zkl_of(kl);
    double dbsfc;
    if (LAZY) {
      const int kref = kp.kref[kl];                // wave-uniform
      for (; nprep < kref; ++nprep) {
        const int kk = (nprep + 1 <= km) ? nprep + 1 : km;
        const long long ot = c.base3 + (long long)(kk - 1) * n2;
        const MwjfTS2 x = mwjf_prep2(tmask(T[ot]), S[ot]);
        shtop[nprep][threadIdx.x] = x.TQ; shtop[KR + nprep][threadIdx.x] = x.SQ; shtop[2 * KR + nprep][threadIdx.x] = x.A2;
      }
      auto top_at = [&](int t) { MwjfTS2 x; x.TQ = shtop[t - 1][threadIdx.x]; x.SQ = shtop[KR + t - 1][threadIdx.x]; x.A2 = shtop[2 * KR + t - 1][threadIdx.x]; return x; };
      const MwjfTS2 xk = mwjf_prep2(tmask(T[o]), S[o]);
      const MwjfP P = mwjf_level(g.pressz[kl]);
      const double rhok = mwjf_eval2(P, xk);
      double rhoavg = mwjf_eval2(P, top_at(kref));
      if (kref != 1) {
        rhoavg = rhoavg * (surfthick - g.zw[kref - 1]);
#pragma unroll 1
        for (int kt = 1; kt <= kref - 1; ++kt) rhoavg = rhoavg + g.dz[kt] * mwjf_eval2(P, top_at(kt));
        rhoavg = rhoavg / surfthick;
      }
      dbsfc = 0.0;
      if (rhok != 0.0) dbsfc = GRAV * (1.0 - rhoavg / rhok);
    } else dbsfc = DBSFC[o];
    double vshear = 0.0;
    if (!edge) {
      if (LAZY && kp.WUK) {
        const double w00 = (kl <= wuk[0]) ? WU[o] : kpp_ushear_point<PBC>(g, kp, UU, VV, c.base3, kl, c.q2);
        const double w10 = (kl <= wuk[1]) ? WU[o - 1] : kpp_ushear_point<PBC>(g, kp, UU, VV, c.base3 - 1, kl, c.q2 - 1);
        const double w01 = (kl <= wuk[2]) ? WU[o - nxb] : kpp_ushear_point<PBC>(g, kp, UU, VV, c.base3 - nxb, kl, c.q2 - nxb);
        const double w11 = (kl <= wuk[3]) ? WU[o - 1 - nxb] : kpp_ushear_point<PBC>(g, kp, UU, VV, c.base3 - 1 - nxb, kl, c.q2 - 1 - nxb);
        vshear = fmax(fmax(w00, w10), fmax(w01, w11));
      } else vshear = fmax(fmax(WU[o], WU[o - 1]), fmax(WU[o - nxb], WU[o - 1 - nxb]));
    }
    bfsfc = kpp_bfsfc(kp, bo, bosol, zkl, 2 * kl - 1, chli);
    const double stable = (bfsfc >= 0.0) ? 1.0 : 0.0;
    bfsfc = bfsfc + stable * KPP_EPS;
    if (kp.lcheckekmo) {   // :2426-2455
      if (stable > 0.5 && hekman >= -kp.zgrid[km]) hekman = fmax(zkl, KPP_CEKMAN * ustar / (fabs(kp.FCORT[c.q2]) + KPP_EPS));
      const double hmon_dn = stable * KPP_CMONOB * ustar * ustar * ustar / KPP_VONKAR / bfsfc + (stable - 1.0) * kp.zgrid[km];
      if (hmon_dn <= zkl && hmon_up > -z_up) {
        const double w = (hmon_dn - hmon_up) / (z_up + zkl);
        hlimit = (hmon_dn - w * zkl) / (1.0 - w);
      }
      hmon_up = hmon_dn;
    }
    double wm_unused = 0.0, ws;
    kpp_wscale<false>(KPP_EPSSFC, zkl, ustar, bfsfc, wm_unused, ws);
    const double db = DBLOC[o];
    double bfr = sqrt(0.5 * (db + fabs(db) + KPP_EPS2) / (kp.zgrid[kl] - kp.zgrid[kl + 1]));
    if (PBC) bfr = (kl < km) ? sqrt(0.5 * (db + fabs(db) + KPP_EPS2) / (0.5 * (pbc_dz(g, kl, kmt, dzbc) + pbc_dz(g, kl + 1, kmt, dzbc))))
                             : sqrt(0.5 * (db + fabs(db) + KPP_EPS2) / pbc_dz(g, kl, kmt, dzbc));
    const double zref = -surfthick / 2.0;
    double wmm = zkl * ws * bfr * ((kp.Vtc / KPP_RICR) * fmax(2.1 - 200.0 * bfr, KPP_CONCV));
    double wk = (kmt >= kl) ? (zref - kp.zgrid[kl]) * dbsfc : 0.0;
    double rib_dn;
    if (PBC) {
      const double ht = -kp.zgrid[kl - 1] + 0.5 * (pbc_dz(g, kl - 1, kmt, dzbc) + pbc_dz(g, kl, kmt, dzbc) - pbc_dz(g, 1, kmt, dzbc));
      const double hu = -kp.zgrid[kl - 1] + 0.5 * (pbc_dz(g, kl, kmu_c, dzub) + pbc_dz(g, kl - 1, kmu_c, dzub) - pbc_dz(g, 1, kmu_c, dzub));
      wk = (kmt >= kl) ? dbsfc / ht : 0.0;
      wmm = wmm / (ht * ht);
      rib_dn = wk / (vshear + wmm + KPP_EPS / (hu * hu));
    } else rib_dn = wk / (vshear + wmm + KPP_EPS);
    if (kbl == kmt && rib_dn > KPP_RICR) {
      const double slope_up = (rib_upper - rib_up) / (z_up - z_upper);
      const double d = z_up + zkl;
      const double a_co = (rib_dn - rib_up - slope_up * (zkl + z_up)) / (d * d);
      const double b_co = slope_up + 2.0 * a_co * z_up;
      const double c_co = rib_up + z_up * (a_co * z_up + slope_up) - KPP_RICR;
      const double sqrt_arg = b_co * b_co - 4.0 * a_co * c_co;
      if ((fabs(b_co) > KPP_EPS && fabs(a_co) / fabs(b_co) <= KPP_EPS) || sqrt_arg <= 0.0)
        hblt = -z_up + (z_up + zkl) * (KPP_RICR - rib_up) / (rib_dn - rib_up);
      else hblt = (-b_co + sqrt(sqrt_arg)) / (2.0 * a_co);
      kbl = kl;
    }
    rib_upper = rib_up; rib_up = rib_dn;
    z_upper = z_up; z_up = kp.zgrid[kl];
  }
  if (kp.lcheckekmo) {   // :2676-2690; the reference tests against ZKL of the last pass of the march, -zgrid(km)
    if (hekman < hlimit) hlimit = hekman;
    for (int kl = 2; kl <= km; ++kl)
      if (hlimit < hblt && hlimit > -kp.zgrid[kl - 1] && hlimit <= zkl_of(km)) { hblt = hlimit; kbl = kl; }
  }
  if (LAZY && km >= 2) {      // what the pass kl = km leaves in bfsfc
    bfsfc = kpp_bfsfc(kp, bo, bosol, -kp.zgrid[km], 2 * km - 1, chli);
    bfsfc = bfsfc + ((bfsfc >= 0.0) ? 1.0 : 0.0) * KPP_EPS;
  }
  kp.HBLT0[c.q2] = hblt;
  kp.KBL0[c.q2] = kbl;
  kp.USTAR[c.q2] = ustar;
  kp.BFSFC[c.q2] = bfsfc;    // value of the last kl pass (vmix_kpp.F90: no short-wave branch)
  if (kp.lshort_wave) { kp.BO[c.q2] = bo; kp.BOSOL[c.q2] = bosol; }
}

// ---- smooth_hblt + blmix + interior convection + masks + non-local source -------------------
// PBC: partial bottom cells (vmix_kpp.F90:3835-3864, 2911-2923, 2948-2973, 3075-3083, 3155-3165, 1220-1222, 1296-1302)
// SAME: the two tracer classes share one diffusivity array (KppDev::vdc_same; a template flag so that the level march below holds no
// branch around a load or a store).  The march keeps the operands of the next level in flight (two operand sets used in turn).
// SPARSE (r3; with KppDev::CONVB from the column-march interior kernel, flat bottom): part (1) of the level march touches only what
// changes.  Below the boundary layer a level is rewritten only where it is convectively unstable (visc + 0.0 is visc, and the
// zeros below the bottom are already there), found from the 64-bit mask of the column instead of its DBLOC values; the non-local
// source is written down to KBL and cleared only as deep as the previous evaluation into the same buffers left it non-zero
// (the KBL stored with them; every level after a caller wrote KPP_SRC).  A level no lane of the wave has to touch is skipped
// by the whole wave.  Same values as the streaming form (tests/test_gpu_parity.py).
struct KppBlRaw { double visc, vd1, vd2, db; };
template <bool PBC = false, bool SAME = false, bool SPARSE = false>
__global__ void __launch_bounds__(POP_COL_THREADS)
k_kpp_blmix(DevGrid g, StepParams sp, KppDev kp, const double *__restrict__ DBLOC, const double *__restrict__ STF1,
            const double *__restrict__ STF2, double *__restrict__ VISC, double *__restrict__ VDC1, double *__restrict__ VDC2,
            double *__restrict__ SRC1, double *__restrict__ SRC2, double *__restrict__ HBLT_OUT) {
  Col c;
  if (!col_setup(g, c, false)) return;
  const int km = g.km, nxb = g.nxb;
  const long long n2 = g.n2;
  const int kmt = g.KMT[c.q2];
  const CArr zgrid = kp.zgrid, hwide = kp.hwide;
  // smooth_hblt(overwrite_hblt=.true., use_hmxl=.false.)
  double w2 = kp.HBLT0[c.q2];
  int kbl = kp.KBL0[c.q2];
  const bool inner = (c.i >= 1 && c.i <= g.nxb - 2 && c.j >= 1 && c.j <= g.nyb - 2);
  if (inner && kmt != 0) {
    double cw = 0.125, ce = 0.125, cn = 0.125, cs = 0.125, cc = 0.5;
    if (g.KMT[c.q2 - 1] == 0) { cc = cc + cw; cw = 0.0; }
    if (g.KMT[c.q2 + 1] == 0) { cc = cc + ce; ce = 0.0; }
    if (g.KMT[c.q2 - nxb] == 0) { cc = cc + cs; cs = 0.0; }
    if (g.KMT[c.q2 + nxb] == 0) { cc = cc + cn; cn = 0.0; }
    w2 = cw * kp.HBLT0[c.q2 - 1] + ce * kp.HBLT0[c.q2 + 1] + cs * kp.HBLT0[c.q2 - nxb] + cn * kp.HBLT0[c.q2 + nxb] + cc * kp.HBLT0[c.q2];
  }
  const double dzbc = PBC ? g.DZBC[c.q2] : 0.0;
  auto dzt_at = [&](int k) { return pbc_dz(g, k, kmt, dzbc); };
  // ztmp(k) = -zgrid(k), or with partial bottom cells -zgrid(k-1) + p5*(DZT(k-1) + DZT(k))
  auto ztmp = [&](int k) { return PBC ? -zgrid[k - 1] + 0.5 * (dzt_at(k - 1) + dzt_at(k)) : -zgrid[k]; };
  if (inner && kmt >= 1 && w2 > ztmp(kmt)) w2 = ztmp(kmt);
  const double hblt = fmax(w2, -zgrid[1]);
  if (inner && kmt != 0)
    for (int k = 1; k <= km; ++k)
      if (hblt > -zgrid[k - 1] && hblt <= ztmp(k)) kbl = max(k, 2);
  HBLT_OUT[c.q2] = hblt;
  const int kbl_prev = SPARSE ? kp.KBL[c.q2] : 0;   // KBL of the previous evaluation into this set of output buffers
  kp.KBL[c.q2] = kbl;
  const double ustar = kp.USTAR[c.q2];
  double bfsfc = kp.BFSFC[c.q2];
  if (kp.lshort_wave) bfsfc = kpp_bfsfc(kp, kp.BO[c.q2], kp.BOSOL[c.q2], hblt, 0, kp.sw_type == 2 ? kp.CHLI[c.q2] : 0);   // forcing down to the boundary layer depth (:2707-2742)
  const double stable = (bfsfc >= 0.0) ? 1.0 : 0.0;
  bfsfc = bfsfc + stable * KPP_EPS;
  // blmix: matching at the boundary-layer base
  double wm, ws;
  kpp_wscale<true>(KPP_EPSSFC, hblt, ustar, bfsfc, wm, ws);
  double casea_x = -zgrid[kbl] - 0.5 * hwide[kbl] - hblt;
  if (PBC) casea_x = (kbl == 1) ? -zgrid[0] - hblt : -zgrid[kbl - 1] + 0.5 * dzt_at(kbl - 1) - hblt;
  const double casea = 0.5 + (casea_x >= 0.0 ? 0.5 : -0.5);
  const int nc = (casea > 0.5) ? 1 : 0;
  const int kn = nc * (kbl - 1) + (1 - nc) * kbl;
  const double u2 = ustar * ustar;
  const double f1 = stable * 5.0 * bfsfc / (u2 * u2 + KPP_EPS);
  const long long vb = ((long long)c.b * (km + 2)) * n2 + c.p2;
  auto visc_at = [&](int k) { return (k >= 1 && k <= km) ? VISC[c.base3 + (long long)(k - 1) * n2] : 0.0; };
  double gat1[3] = {0.0, 0.0, 0.0}, dat1[3] = {0.0, 0.0, 0.0};
  if (kn >= 1 && kn <= km) {
    const int k = kn;
    double dh = 0.5 * hwide[k] - zgrid[k] - hblt;
    double R = 1.0 - dh / hwide[k];
    double hup = hwide[k], hdn = hwide[k + 1];
    if (PBC) {
      const double w1 = (k == 1) ? 0.0 : dzt_at(k - 1);
      hdn = (k == km) ? KPP_EPS : dzt_at(k + 1);
      hup = dzt_at(k);
      dh = -zgrid[k - 1] + dzt_at(k) + 0.5 * w1 - hblt;
      R = 1.0 - dh / dzt_at(k);
    }
    double fm[3], f0[3], fp[3];
    fm[0] = visc_at(k - 1); f0[0] = visc_at(k); fp[0] = visc_at(k + 1);
    fm[2] = VDC1[vb + (long long)(k - 1) * n2]; f0[2] = VDC1[vb + (long long)k * n2]; fp[2] = VDC1[vb + (long long)(k + 1) * n2];
    if (!SAME) { fm[1] = VDC2[vb + (long long)(k - 1) * n2]; f0[1] = VDC2[vb + (long long)k * n2]; fp[1] = VDC2[vb + (long long)(k + 1) * n2]; }
    else { fm[1] = fm[2]; f0[1] = f0[2]; fp[1] = fp[2]; }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double up = (fm[q] - f0[q]) / hup, dn = (f0[q] - fp[q]) / hdn;
      const double Pq = 0.5 * ((1.0 - R) * (up + fabs(up)) + R * (dn + fabs(dn)));
      const double Hq = f0[q] + Pq * dh;
      const double wv = (q == 0) ? wm : ws;
      gat1[q] = Hq / hblt / (wv + KPP_EPS);
      dat1[q] = fmin(-Pq / (wv + KPP_EPS) + f1 * Hq, 0.0);
    }
  }
  auto shape = [&](double sig, double wv, int q) {
    return hblt * wv * sig * (1.0 + sig * ((sig - 2.0) + (3.0 - 2.0 * sig) * gat1[q] + (sig - 1.0) * dat1[q]));
  };
  double dkm1[3];
  {
    const int k = kbl - 1;
    const double sig = -zgrid[k] / hblt;
    double wm1, ws1;
    kpp_wscale<true>(fmin(sig, KPP_EPSSFC), hblt, ustar, bfsfc, wm1, ws1);
    dkm1[0] = shape(sig, wm1, 0); dkm1[1] = shape(sig, ws1, 1); dkm1[2] = shape(sig, ws1, 2);
  }
  // The level march in two parts (round 3).
  // (1) Every level as an INTERIOR level -- convection + masks, and the non-local source of a level below the boundary layer
  //     (ghat = 0) -- in one branch-free streaming loop with the next level's operands in flight: 3 loads, 4 stores and three
  //     divisions per level, no divergent block around a memory operation.
  // (2) The levels k < KBL again, now with the boundary-layer coefficients (similarity functions, shape function, the blending at
  //     KBL - 1, which needs the interior values of that level: read before (1) overwrites them) and the non-local source down to
  //     level KBL.  This is the divergent, arithmetic-heavy part; it touches KBL levels only.
  // Every value is formed by the operations of the single loop it replaces (the reference's order): a level k < KBL takes its
  // coefficients from (2), where the convective terms are +0.0 exactly as before, a level k >= KBL from (1).
  const double stf1 = STF1[c.q2], stf2 = STF2[c.q2];
  const int kb1 = (kbl - 1 >= 1) ? kbl - 1 : 1;                 // the level of the blending (clamped: KBL = 1 on land has none)
  const double visc_b = VISC[c.base3 + (long long)(kb1 - 1) * n2], vd1_b = VDC1[vb + (long long)kb1 * n2];
  const double vd2_b = SAME ? vd1_b : VDC2[vb + (long long)kb1 * n2];
  if (SPARSE) {
    const unsigned long long bits = kp.CONVB[c.q2];
    const int kdeep = kp.src_clear_all ? km : kbl_prev;
    const int ktop = (kbl > 1) ? kbl : 1;
#pragma unroll 1
    for (int k = 1; k <= km; ++k) {
      const long long o = c.base3 + (long long)(k - 1) * n2;
      const bool conv = k >= ktop && k >= kbl && k <= km - 1 && k < kmt && ((bits >> (k - 1)) & 1ull);
      if (__any(conv)) {      // convection (vmix_kpp.F90:1218-1240) where the column is unstable; the other lanes rewrite their value
        const double visc = VISC[o], vd1 = VDC1[vb + (long long)k * n2];
        const double cvv = conv ? sp.convect_visc * 1.0 : 0.0, cvd = conv ? sp.convect_diff * 1.0 : 0.0;
        VISC[o] = conv ? visc + cvv : visc;
        VDC1[vb + (long long)k * n2] = conv ? vd1 + cvd : vd1;
        if (!SAME) { const double vd2 = VDC2[vb + (long long)k * n2]; VDC2[vb + (long long)k * n2] = conv ? vd2 + cvd : vd2; }
      }
      const bool clr = k > kbl && k <= kdeep;
      if (__any(clr)) {       // +-0 below the boundary layer: stf / dz * (0 - 0); lanes above it are rewritten by part (2)
        const double dzk = (PBC && k > 1) ? dzt_at(k) : g.dz.u(k);
        SRC1[o] = stf1 / dzk * (0.0 - 0.0); SRC2[o] = stf2 / dzk * (0.0 - 0.0);
      }
    }
  } else {
  auto load = [&](int k) {
    KppBlRaw r;
    const long long o = c.base3 + (long long)(k - 1) * n2;
    r.visc = VISC[o]; r.vd1 = VDC1[vb + (long long)k * n2]; r.vd2 = SAME ? 0.0 : VDC2[vb + (long long)k * n2]; r.db = DBLOC[o];
    return r;
  };
  auto level = [&](int k, const KppBlRaw &cu, KppBlRaw &nx) {
    nx = load(k + 1 <= km ? k + 1 : km);
    const long long o = c.base3 + (long long)(k - 1) * n2;
    const double zk = zgrid.u(k), zk1 = zgrid.u(k + 1), dz_k = g.dz.u(k);
    double visc = cu.visc, vd1 = cu.vd1, vd2 = SAME ? cu.vd1 : cu.vd2;
    if (k <= km - 1) {
      const double N2 = PBC ? cu.db / (0.5 * (dzt_at(k) + dzt_at(k + 1))) : cu.db / (zk - zk1);
      const double fcon = (N2 > 0.0) ? 0.0 : 1.0;
      double cvv = 0.0, cvd = 0.0;
      if (k >= kbl) { cvv = sp.convect_visc * fcon; cvd = sp.convect_diff * fcon; }
      if (k < kmt) { visc = visc + cvv; vd1 = vd1 + cvd; vd2 = vd2 + cvd; }
      else { visc = 0.0; vd1 = 0.0; vd2 = 0.0; }
    } else { vd1 = 0.0; vd2 = 0.0; }
    VISC[o] = visc;
    VDC1[vb + (long long)k * n2] = vd1;
    if (!SAME) VDC2[vb + (long long)k * n2] = vd2;   // one array when the two classes share their values (no double diffusion)
    const double fl1 = vd1 * 0.0, fl2 = vd2 * 0.0;  // ghat = 0 below the boundary layer
    const double dzk = (PBC && k > 1) ? dzt_at(k) : dz_k;
    const double p1 = (k == 1) ? -fl1 : 0.0 - fl1, p2 = (k == 1) ? -fl2 : 0.0 - fl2;
    SRC1[o] = stf1 / dzk * p1; SRC2[o] = stf2 / dzk * p2;
  };
  {
    KppBlRaw A = load(1), B;
    int k = 1;
#pragma unroll 1
    for (; k + 1 <= km; k += 2) { level(k, A, B); level(k + 1, B, A); }
    if (k <= km) level(k, A, B);
  }
  }
  // (2) the boundary layer
  double flux1_prev = 0.0, flux2_prev = 0.0;   // VDC(k-1)*GHAT(k-1) per tracer class
#pragma unroll 1
  for (int k = 1; k <= kbl && k <= km; ++k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    const double dzk = (PBC && k > 1) ? dzt_at(k) : g.dz[k];
    if (k == kbl) {     // first level below: its source still feels the flux through its upper face
      if (k > 1) { SRC1[o] = stf1 / dzk * (flux1_prev - 0.0); SRC2[o] = stf2 / dzk * (flux2_prev - 0.0); }
      else if (SPARSE) { SRC1[o] = stf1 / dzk * (-(0.0)); SRC2[o] = stf2 / dzk * (-(0.0)); }   // KBL = 1 (land): -(vd * ghat), ghat = 0
      break;
    }
    double sig = (-zgrid[k] + 0.5 * hwide[k]) / hblt;
    if (PBC && k > 1) sig = (-zgrid[k - 1] + 0.5 * dzt_at(k - 1) + dzt_at(k)) / hblt;
    double wmk, wsk;
    kpp_wscale<true>(fmin(sig, KPP_EPSSFC), hblt, ustar, bfsfc, wmk, wsk);
    double b0 = shape(sig, wmk, 0), b1 = shape(sig, wsk, 1), b2 = shape(sig, wsk, 2);
    double ghat = (1.0 - stable) * kp.cg / (wsk * hblt + KPP_EPS);
    if (k == kbl - 1 && k <= km - 1) {
      double dh = (hblt + zgrid[k]) / (zgrid[k] - zgrid[k + 1]);
      if (PBC) {
        const double w1 = (k == 1) ? -0.5 * dzt_at(k) : zgrid[k - 1] - 0.5 * (dzt_at(k - 1) + dzt_at(k));
        dh = (hblt + w1) / (0.5 * (dzt_at(k) + dzt_at(k + 1)));
      }
      const double omd = 1.0 - dh;
      b0 = omd * visc_b + dh * ((omd * omd) * dkm1[0] + (dh * dh) * (casea * visc_b + (1.0 - casea) * b0));
      b1 = omd * vd2_b + dh * ((omd * omd) * dkm1[1] + (dh * dh) * (casea * vd2_b + (1.0 - casea) * b1));
      b2 = omd * vd1_b + dh * ((omd * omd) * dkm1[2] + (dh * dh) * (casea * vd1_b + (1.0 - casea) * b2));
      ghat = (1.0 - casea) * ghat;
    }
    double visc = b0, vd2 = b1, vd1 = b2;
    if (k <= km - 1) {
      if (k < kmt) { visc = visc + 0.0; vd1 = vd1 + 0.0; vd2 = vd2 + 0.0; }   // the convective terms of a level above KBL
      else { visc = 0.0; vd1 = 0.0; vd2 = 0.0; }
    } else { vd1 = 0.0; vd2 = 0.0; }
    VISC[o] = visc;
    VDC1[vb + (long long)k * n2] = vd1;
    if (!SAME) VDC2[vb + (long long)k * n2] = vd2;
    const double fl1 = vd1 * ghat, fl2 = vd2 * ghat;
    if (k == 1) { SRC1[o] = stf1 / dzk * (-fl1); SRC2[o] = stf2 / dzk * (-fl2); }
    else { SRC1[o] = stf1 / dzk * (flux1_prev - fl1); SRC2[o] = stf2 / dzk * (flux2_prev - fl2); }
    flux1_prev = fl1; flux2_prev = fl2;
  }
}

// ---- VVC = tgrid_to_ugrid(VISC) masked by k < KMU; VVC(km) = 0.  3-D parallel ---------------
#define POP_VVC_KC 8   // levels per thread: the four averaging weights and KMU are loaded once per chunk
// ---- diagnostic mixed-layer depths (vmix_kpp.F90:1310-1418; pop_config kpp_ml_diagnostics = 1): HMXL, the depth of the maximum
// buoyancy gradient, and HMXL_DR, the depth where the potential density exceeds its surface value by 3e-5 g/cm^3.  Every cell
// of the block, as the reference's whole-array statements (T, S are 0 below the bottom and on land there too).
__global__ void __launch_bounds__(POP_COL_THREADS)
k_kpp_hmxl(DevGrid g, KppDev kp, const double *__restrict__ T, const double *__restrict__ S, const double *__restrict__ DBSFC,
           double *__restrict__ HMXL, double *__restrict__ HMXL_DR) {
  Col c;
  if (!col_setup(g, c, false)) return;
  const int km = g.km, kmt = g.KMT[c.q2];
  const long long n2 = g.n2;
  // partial bottom cells (:1326-1356): depths and spacings from the cell thicknesses DZT(k) (DZT(0) = 0)
  const bool pbc = g.pbc != 0;
  const double dzb = pbc ? g.DZBC[c.q2] : 0.0;
  auto dzt = [&](int k) { return pbc_dz(g, k, kmt, dzb); };
  double ustar = 0.0, hmxl = (kmt == 1) ? g.zt[1] : 0.0;
  for (int k = 2; k <= km && k <= kmt; ++k) {
    const double zk = pbc ? g.zt[k - 1] + 0.5 * (dzt(k - 1) + dzt(k)) : g.zt[k];
    const double q = DBSFC[c.base3 + (long long)(k - 1) * n2] / zk;
    ustar = (q > ustar) ? q : ustar;
    hmxl = zk;
  }
  double gm1 = 0.0, dbm1 = DBSFC[c.base3];
  for (int k = 2; k <= km; ++k) {
    const double db = DBSFC[c.base3 + (long long)(k - 1) * n2];
    const double dzk = pbc ? 0.5 * (dzt(k) + dzt(k - 1)) : (g.zt[k] - g.zt[k - 1]);
    const double v = (ustar > 0.0) ? (db - dbm1) / dzk : 0.0;
    if (v >= ustar && (v - gm1) != 0.0 && ustar > 0.0) {
      const double bf = (v - ustar) / (v - gm1);
      hmxl = pbc ? (g.zt[k - 1] + 0.25 * (dzt(k - 1) + dzt(k))) * (1.0 - bf) + (g.zt[k - 1] - 0.25 * (dzt(k - 2) + dzt(k - 1))) * bf
                 : -0.5 * (kp.zgrid[k] + kp.zgrid[k - 1]) * (1.0 - bf) - 0.5 * (kp.zgrid[k - 1] + kp.zgrid[k - 2]) * bf;
      ustar = 0.0;
    }
    gm1 = v; dbm1 = db;
  }
  HMXL[c.q2] = hmxl;
  const MwjfP P1 = mwjf_level(g.pressz[1]);
  const double rho1 = mwjf_rho<false>(P1, tmask(T[c.base3]), S[c.base3], nullptr, nullptr);
  const double target = rho1 + 3.0e-05;
  double rhok = rho1, hdr = (kmt == 1) ? g.zt[1] : 0.0;
  bool found = (kmt == 1);
  for (int k = 1; k <= km - 1; ++k) {
    const long long o = c.base3 + (long long)k * n2;
    const double rkp1 = mwjf_rho<false>(P1, tmask(T[o]), S[o], nullptr, nullptr);
    if (target > rhok && target <= rkp1 && !found) {
      hdr = g.zt[k] + (target - rhok) * (g.zt[k + 1] - g.zt[k]) / (rkp1 - rhok + KPP_EPS);
      found = true;
    }
    rhok = rkp1;
  }
  HMXL_DR[c.q2] = hdr;
}

__global__ void k_kpp_vvc(DevGrid g, const double *__restrict__ VISC, double *__restrict__ VVC, int patch) {
  int p2;
  const int k0 = blockIdx.y * POP_VVC_KC + 1, b = blockIdx.z;
  if (!patch_cell(g, patch, b, p2)) return;
  const int i = p2 % g.nxb, j = p2 / g.nxb, nxb = g.nxb;
  const long long q2 = (long long)b * g.n2 + p2;
  const bool in = i < g.nxb - 1 && j < g.nyb - 1;
  const int kmu = g.KMU[q2];
  const double au0 = g.AU0[q2], aun = g.AUN[q2], aue = g.AUE[q2], aune = g.AUNE[q2];
  const int k1 = min(k0 + POP_VVC_KC - 1, g.km);
  for (int k = k0; k <= k1; ++k) {
    const long long o = (long long)b * g.n3 + (long long)(k - 1) * g.n2 + p2;
    double v = 0.0;
    if (k <= g.km - 1 && k < kmu && in)
      v = au0 * VISC[o] + aun * VISC[o + nxb] + aue * VISC[o + 1] + aune * VISC[o + nxb + 1];
    VVC[o] = v;
  }
}

// VVC from VISC in 64 x R patches on bandwidth-bound grids (a row-marching form that fetches every VISC value once was measured slower:
// profiles/r04_ab_kpp_vvc_rows.txt)
inline void launch_kpp_vvc(const DevGrid &g, const HostModel &h, hipStream_t st, const double *VISC, double *VVC) {
  const int vp = patch_rows(g, h.tun.del4_tile);
  hipLaunchKernelGGL(k_kpp_vvc, dim3(patch_grid_x(g, vp), (g.km + POP_VVC_KC - 1) / POP_VVC_KC, g.nblocks), dim3(vp ? 64 * vp : 256), 0, st, g, VISC, VVC, vp);
}

// ---- host side ---------------------------------------------------------------------------------
// per-context KPP state (MixDev::kpp)
// col: bit 0 = ushear, bit 1 = buoydiff in column form.  side / ev_*: second HIP stream on which the shear kernel (needs only
// U, V; consumed by bldepth) runs beside buoydiff + interior (POP_KPP_SIDE_STREAM=0 keeps everything on one stream)
struct KppHost { KppDev dev; int max_kref = 1; int col = 0; int *wuk = nullptr; unsigned long long *convb = nullptr; hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_bd = nullptr; };
inline void kpp_destroy(MixDev &m) {
  KppHost *k = (KppHost *)m.kpp;
  if (k) {
    if (k->ev_fork) hipEventDestroy(k->ev_fork);
    if (k->ev_join) hipEventDestroy(k->ev_join);
    if (k->ev_bd) hipEventDestroy(k->ev_bd);
    if (k->side) hipStreamDestroy(k->side);
  }
  delete k; m.kpp = nullptr;
}

// set_chl (sw_absorption.F90:500-512): column of the transmission table for a chlorophyll amount (mg/m^3); the reference
// assigns the quotient to an integer array, which truncates
inline int sw_chl_index(const SwTab &T, double chl) {
  chl = std::max(chl, T.chlmin); chl = std::min(chl, T.chlmax);
  int idx = (int)(std::log10(chl / T.chlmin) / T.dlogchl);
  return std::min(std::max(idx, 0), 400);
}
// device tables of the short-wave absorption (SwTab): sw_absorb(0:km) for sw_absorption_type 0 / 1, the chlorophyll
// transmission table for 2 (set_chl_trn, sw_absorption.F90:650-716, on the table of Ohlmann (2003), :135-216)
inline int sw_tables_create(HostModel &h, std::vector<void *> &allocs, std::string &err) {
  if (h.sw.swabs) return 0;
  const pop_config &c = h.c;
  const int km = h.km, type = c.sw_absorption_type, jt = c.jerlov_water_type ? c.jerlov_water_type : 3;
  const size_t a2 = h.n2 * h.nblocks;
  void *p;
  auto up = [&](const void *src, size_t bytes, void **dst) -> int {
    if (hipMalloc(dst, bytes) != hipSuccess) return 1;
    allocs.push_back(*dst);
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess ? 1 : 0;
  };
  {   // init_sw_absorption :355-370
    std::vector<double> sa(km + 1, 0.0);
    sa[0] = 1.0;
    if (type == 1) {
      const double rfac[5] = {0.58, 0.62, 0.67, 0.77, 0.78}, depth1[5] = {0.35, 0.60, 1.00, 1.50, 1.40}, depth2[5] = {23.0, 20.0, 17.0, 14.0, 7.90};
      for (int kk = 1; kk <= km - 1; ++kk) {
        const double dm = -h.zw[kk] * 0.01;
        sa[kk] = (dm < -200.0) ? 0.0 : rfac[jt - 1] * std::exp(dm / depth1[jt - 1]) + (1.0 - rfac[jt - 1]) * std::exp(dm / depth2[jt - 1]);
      }
    }
    if (up(sa.data(), sa.size() * 8, &p)) { err = "sw alloc"; return 1; } h.sw.swabs = (double *)p;
  }
  if (type == 2) {
    static const double cnc[31] = {.001, .005, .01, .02, .03, .05, .10, .15, .20, .25, .30, .35, .40, .45, .50, .60, .70, .80, .90, 1.00, 1.50,
      2.00, 2.50, 3.00, 4.00, 5.00, 6.00, 7.00, 8.00, 9.00, 10.00};
    static const double A1t[31] = {0.4421, 0.4451, 0.4488, 0.4563, 0.4622, 0.4715, 0.4877, 0.4993, 0.5084, 0.5159, 0.5223, 0.5278, 0.5326, 0.5369,
      0.5408, 0.5474, 0.5529, 0.5576, 0.5615, 0.5649, 0.5757, 0.5802, 0.5808, 0.5788, 0.56965, 0.55638, 0.54091, 0.52442, 0.50766, 0.49110, 0.47505};
    static const double A2t[31] = {0.2981, 0.2963, 0.2940, 0.2894, 0.2858, 0.2800, 0.2703, 0.2628, 0.2571, 0.2523, 0.2481, 0.2444, 0.2411, 0.2382,
      0.2356, 0.2309, 0.2269, 0.2235, 0.2206, 0.2181, 0.2106, 0.2089, 0.2113, 0.2167, 0.23357, 0.25504, 0.27829, 0.30274, 0.32698, 0.35056, 0.37303};
    static const double B1t[31] = {0.0287, 0.0301, 0.0319, 0.0355, 0.0384, 0.0434, 0.0532, 0.0612, 0.0681, 0.0743, 0.0800, 0.0853, 0.0902, 0.0949,
      0.0993, 0.1077, 0.1154, 0.1227, 0.1294, 0.1359, 0.1640, 0.1876, 0.2082, 0.2264, 0.25808, 0.28498, 0.30844, 0.32932, 0.34817, 0.36540, 0.38132};
    static const double B2t[31] = {0.3192, 0.3243, 0.3306, 0.3433, 0.3537, 0.3705, 0.4031, 0.4262, 0.4456, 0.4621, 0.4763, 0.4889, 0.4999, 0.5100,
      0.5191, 0.5347, 0.5477, 0.5588, 0.5682, 0.5764, 0.6042, 0.6206, 0.6324, 0.6425, 0.66172, 0.68144, 0.70086, 0.72144, 0.74178, 0.76190, 0.78155};
    const int ksol = 2 * km, nsub = 400;
    std::vector<double> ztr(ksol + 1, 0.0), Tr((size_t)(ksol + 1) * (nsub + 1));
    for (int kk = 1; kk <= km; ++kk) { ztr[2 * kk - 1] = h.zt[kk]; ztr[2 * kk] = h.zw[kk]; }
    h.sw.chlmin = cnc[0]; h.sw.chlmax = cnc[30];
    h.sw.dlogchl = (std::log10(h.sw.chlmax) - std::log10(h.sw.chlmin)) / (double)nsub;
    double logchl = std::log10(h.sw.chlmin) - h.sw.dlogchl;
    for (int n = 0; n <= nsub; ++n) {
      logchl = logchl + h.sw.dlogchl;
      const double amount = std::pow(10.0, logchl);
      int mc = -1;
      for (int q = 0; q < 30; ++q) if (cnc[q] <= amount && amount <= cnc[q + 1]) { mc = q; break; }
      if (mc < 0) mc = (amount < cnc[0]) ? 0 : 29;
      const double w2 = (amount - cnc[mc]) / (cnc[mc + 1] - cnc[mc]), w1 = 1.0 - w2;
      const double A1 = A1t[mc] * w1 + A1t[mc + 1] * w2, A2 = A2t[mc] * w1 + A2t[mc + 1] * w2;
      const double B1 = B1t[mc] * w1 + B1t[mc + 1] * w2, B2 = B2t[mc] * w1 + B2t[mc + 1] * w2;
      double *Trn = Tr.data() + (size_t)n * (ksol + 1);
      Trn[0] = 1.0;
      for (int kk = 1; kk <= ksol; ++kk) {
        double arg = std::min(B1 * ztr[kk] * 0.01, 35.0);
        Trn[kk] = A1 * std::exp(-arg);
        arg = std::min(B2 * ztr[kk] * 0.01, 35.0);
        Trn[kk] = Trn[kk] + A2 * std::exp(-arg);
      }
    }
    if (up(ztr.data(), ztr.size() * 8, &p)) { err = "sw alloc"; return 1; } h.sw.ztr = (double *)p;
    if (up(Tr.data(), Tr.size() * 8, &p)) { err = "sw alloc"; return 1; } h.sw.Tr = (double *)p;
    h.sw.ksol = ksol;
    std::vector<int> ci(a2, sw_chl_index(h.sw, 0.25));   // no chlorophyll forcing file here: 0.25 mg/m^3 until the caller sets "CHL"
    if (up(ci.data(), a2 * 4, &p)) { err = "sw alloc"; return 1; } h.sw.CHLI = (int *)p;
  }
  return 0;
}
inline int kpp_create(HostModel &h, const DevGrid &g, MixDev &m, std::vector<void *> &allocs, std::string &err) {
  const pop_config &c = h.c;
  if (c.lshort_wave && (c.sw_absorption_type < 0 || c.sw_absorption_type > 2)) { err = "KPP: sw_absorption_type: 0 top-layer, 1 jerlov, 2 chlorophyll"; return 1; }
  if (c.jerlov_water_type < 0 || c.jerlov_water_type > 5) { err = "KPP: jerlov_water_type: 1..5 (0 = 3)"; return 1; }
  if (c.num_v_smooth_Ri < 1) { err = "KPP: num_v_smooth_Ri must be >= 1 (the reference leaves FRI unset otherwise)"; return 1; }
  const int km = h.km;
  std::vector<double> zgrid(km + 3, 0.0), hwide(km + 3, 0.0), bvdc(km + 3, 0.0), bvvc(km + 3, 0.0);
  std::vector<int> kref(km + 3, 1);
  zgrid[0] = KPP_EPS; hwide[0] = KPP_EPS;
  for (int k = 1; k <= km; ++k) { zgrid[k] = -h.zt[k]; hwide[k] = h.dz[k]; }
  zgrid[km + 1] = -h.zw[km]; hwide[km + 1] = KPP_EPS;
  for (int k = 1; k <= km; ++k) {
    bvdc[k] = c.bckgrnd_vdc1 + c.bckgrnd_vdc2 * std::atan(c.bckgrnd_vdc_linv * (h.zw[k] - c.bckgrnd_vdc_dpth));
    bvvc[k] = c.Prandtl * bvdc[k];
    const double surfthick = KPP_EPSSFC * h.zt[k];
    kref[k] = k;
    for (int kt = 1; kt <= k; ++kt) if (h.zw[kt] >= surfthick) { kref[k] = kt; break; }
  }
  auto up = [&](const void *src, size_t bytes, void **dst) -> int {
    if (hipMalloc(dst, bytes) != hipSuccess) return 1;
    allocs.push_back(*dst);
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess ? 1 : 0;
  };
  KppHost *K = new KppHost();
  m.kpp = K;
  if (!tun_off(h.tun.kpp_side_stream)) {
    if (hipStreamCreateWithFlags(&K->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&K->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&K->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&K->ev_bd, hipEventDisableTiming) != hipSuccess) { err = "kpp: side stream"; return 1; }
  }
  KppDev &k = K->dev;
  void *p;
  if (up(zgrid.data(), zgrid.size() * 8, &p)) { err = "kpp alloc"; return 1; } k.zgrid = (double *)p;
  if (up(hwide.data(), hwide.size() * 8, &p)) { err = "kpp alloc"; return 1; } k.hwide = (double *)p;
  if (up(bvdc.data(), bvdc.size() * 8, &p)) { err = "kpp alloc"; return 1; } k.bckgrnd_vdc = (double *)p;
  if (up(bvvc.data(), bvvc.size() * 8, &p)) { err = "kpp alloc"; return 1; } k.bckgrnd_vvc = (double *)p;
  if (up(kref.data(), kref.size() * 4, &p)) { err = "kpp alloc"; return 1; } k.kref = (int *)p;
  {
    std::vector<double> z6(6 * (km + 3), 0.0);
    if (up(z6.data(), z6.size() * 8, &p)) { err = "kpp alloc"; return 1; }
    if (km + 1 <= 1024) hipLaunchKernelGGL(k_kpp_level_table, dim3(1), dim3(km + 1), 0, 0, g, (double *)p);
    if (hipDeviceSynchronize() != hipSuccess) { err = "kpp: level table"; return 1; }
    k.eosP = (double *)p;
  }
  const size_t a2 = h.n2 * h.nblocks;
  std::vector<double> z(a2, 0.0);
  if (up(z.data(), a2 * 8, &p)) { err = "kpp alloc"; return 1; } k.HBLT0 = (double *)p;
  if (up(z.data(), a2 * 8, &p)) { err = "kpp alloc"; return 1; } k.USTAR = (double *)p;
  if (up(z.data(), a2 * 8, &p)) { err = "kpp alloc"; return 1; } k.BFSFC = (double *)p;
  std::vector<int> zi(a2, 0);
  if (up(zi.data(), a2 * 4, &p)) { err = "kpp alloc"; return 1; } k.KBL0 = (int *)p;
  if (up(zi.data(), a2 * 4, &p)) { err = "kpp alloc"; return 1; } k.KBL = (int *)p;
  if (up(zi.data(), a2 * 4, &p)) { err = "kpp alloc"; return 1; } K->wuk = (int *)p;
  { std::vector<unsigned long long> zb(a2, 0ull); if (up(zb.data(), a2 * 8, &p)) { err = "kpp alloc"; return 1; } K->convb = (unsigned long long *)p; }
  k.CONVB = nullptr; k.src_clear_all = 0;
  k.WUK = nullptr; k.wu_margin = 3;
  k.Vtc = std::sqrt(0.2 / KPP_C_S / KPP_EPSSFC) / (KPP_VONKAR * KPP_VONKAR);
  k.cg = KPP_CSTAR * KPP_VONKAR * std::pow(KPP_C_S * KPP_VONKAR * KPP_EPSSFC, 1.0 / 3.0);
  k.rich_mix = c.kpp_rich_mix; k.lrich = c.lrich; k.ldbl_diff = c.ldbl_diff; k.nsmooth = c.num_v_smooth_Ri;
  k.lshort_wave = c.lshort_wave ? 1 : 0; k.sw_type = c.sw_absorption_type; k.jerlov = c.jerlov_water_type ? c.jerlov_water_type : 3; k.lcheckekmo = c.lcheckekmo ? 1 : 0;
  if (k.lshort_wave) {
    if (up(z.data(), a2 * 8, &p)) { err = "kpp alloc"; return 1; } k.BO = (double *)p;
    if (up(z.data(), a2 * 8, &p)) { err = "kpp alloc"; return 1; } k.BOSOL = (double *)p;
  }
  if (k.lshort_wave && k.sw_type == 2) {
    if (sw_tables_create(h, allocs, err)) return 1;
    k.ztr = h.sw.ztr; k.Tr = h.sw.Tr; k.ksol = h.sw.ksol; k.CHLI = h.sw.CHLI;
  }
  K->max_kref = 1;
  for (int kk = 1; kk <= km; ++kk) K->max_kref = std::max(K->max_kref, kref[kk]);
  // column (register) forms: bandwidth-bound grids only -- below ~2^19 columns the 3-D-parallel forms win on
  // parallelism.  Measured at tx0.1v3: ushear 11.2 -> 2.7 ms; buoydiff 16.1 -> 14.3 ms (VALU-bound either way:
  // the column form halves the instruction count by hoisting the pressure-independent half of the equation of
  // state, but its 3 x 20 register doubles leave one wave per SIMD).  POP_KPP_COL = bit mask (1 ushear,
  // 2 buoydiff) overrides.
  K->col = (K->max_kref <= 24) ? ((h.n2 * h.nblocks > (1u << 19)) ? 31 : 1) : 0;   // ushear: column form at every size (gx1v7 vmix 0.716 -> 0.692 ms)
  if (tun_set(h.tun.kpp_col)) K->col = (K->max_kref <= 24) ? h.tun.kpp_col : 0;   // bit 0 ushear column form, bit 1 buoydiff column form, bit 2 buoydiff LDS form, bit 3 buoydiff + interior fused, bit 4 as one column march
  (void)m;
  return 0;
}

// KBL of the last evaluation (the tracer kernel reads KPP_SRC down to it only)
inline const int *mix_kpp_kbl(const MixDev &m) { return m.kpp ? ((const KppHost *)m.kpp)->dev.KBL : nullptr; }

inline int kpp_vmix_coeffs(const HostModel &h, const DevGrid &g, const StepParams &sp, const MixDev &m, const MixState &s,
                           hipStream_t st, std::string &err) {
  const KppHost &KH = *(const KppHost *)m.kpp;
  KppDev g_kpp = KH.dev;
  g_kpp.SHF_QSW = s.SHF_QSW; g_kpp.FCORT = g.FCORT;
  if (s.KBL) g_kpp.KBL = s.KBL;
  g_kpp.vdc_same = (s.VDC[0] == s.VDC[1]) ? 1 : 0;
  const int g_kpp_col = KH.col;
  const dim3 GC(col_grid(g, POP_COL_THREADS), g.nblocks), BC(POP_COL_THREADS);
  const dim3 G3((g.n2 + 255) / 256, g.km, g.nblocks);
  double *DBLOC = s.S3a, *DBSFC = s.S3b, *WU = s.S3c, *VISC = s.S3d, *RIW = s.E3;
  // level-parallel LDS form (bit 2 of the mask; the default on bandwidth-bound grids, linear column order only)
  const dim3 GL(col_grid_x(g.n2, POP_COL_THREADS), g.nblocks), BL(POP_COL_THREADS, 4);
  // bit 3: buoydiff and the interior coefficients in ONE level-parallel launch
  const bool fused_bi = (g_kpp_col & 8) && KH.max_kref <= 20 && g.xcd_remap != 2 && g.km <= 64;
  // the surface-layer buoyancy difference on demand inside the boundary-layer-depth march (k_kpp_bldepth<true, .>); POP_KPP_LAZY=0 keeps
  // the full field
  // (with the level-parallel fused kernel and with the plain 3-D buoydiff kernel; the column / LDS buoydiff forms keep the full field)
  const bool plain3d = !fused_bi && !(g_kpp_col & 4 && KH.max_kref <= 28 && g.xcd_remap != 2) && !(g_kpp_col & 2);
  const bool lazy = (fused_bi || plain3d) && KH.max_kref <= 28 && !g_kpp.lcheckekmo && h.c.kpp_ml_diagnostics != 1 &&
                    !tun_off(h.tun.kpp_lazy);
  const bool lazy20 = lazy && KH.max_kref <= 20;
  // shear kernel limited by the previous evaluation's KBL (k_kpp_ushear_col): only with the on-demand march, which can form a level
  // that is missing itself (POP_KPP_USHEAR_HINT=0: every level; POP_KPP_USHEAR_MARGIN: levels beyond the hint, default 3)
  if (lazy && (g_kpp_col & 1) && !tun_off(h.tun.kpp_ushear_hint)) {
    g_kpp.WUK = KH.wuk;
    if (tun_set(h.tun.kpp_ushear_margin)) g_kpp.wu_margin = h.tun.kpp_ushear_margin;
  }
  // the shear of the velocity against its surface-layer reference needs only U and V: on the side stream it overlaps the
  // (VALU-bound) buoydiff and the interior kernel; bldepth waits for it
  const hipStream_t su = KH.side ? KH.side : st;
  // bit 4: buoydiff + interior coefficients as one column march (one smoothing pass, no double diffusion)
  const bool march = lazy && fused_bi && (g_kpp_col & 16) && g_kpp.nsmooth == 1 && !g_kpp.ldbl_diff && g.km >= 3;
  g_kpp.src_clear_all = s.src_clear_all;
  if (g.pbc && march && lazy20 && (g_kpp_col & 1) && !tun_on(h.tun.pbc_generic_kpp)) {
    // partial bottom cells on the production kernel selection (r3): the PBC instantiations of the column-march kernels
    g_kpp.CONVB = (g.km <= 64 && !tun_off(h.tun.kpp_sparse)) ? KH.convb : nullptr;
    if (KH.side) { hipEventRecord(KH.ev_fork, st); hipStreamWaitEvent(KH.side, KH.ev_fork, 0); }
    hipLaunchKernelGGL((k_kpp_ushear_col<24, true>), GC, BC, 0, su, g, g_kpp, s.UMIX, s.VMIX, WU);
    hipLaunchKernelGGL(k_kpp_buoy_interior_march<true>, GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], s.UMIX, s.VMIX, DBLOC, VISC, s.VDC[0], s.VDC[1]);
    if (KH.side) { hipEventRecord(KH.ev_bd, st); hipStreamWaitEvent(KH.side, KH.ev_bd, 0); }
    hipLaunchKernelGGL((k_kpp_bldepth<true, 20, true>), GC, BC, 0, su, g, g_kpp, s.TMIX[0], s.TMIX[1], s.STF[0], s.STF[1], (const double *)DBLOC,
                       (const double *)DBSFC, (const double *)WU, s.UMIX, s.VMIX);
    if (KH.side) { hipEventRecord(KH.ev_join, KH.side); hipStreamWaitEvent(st, KH.ev_join, 0); }
    const bool sp_ = g_kpp.CONVB != nullptr;
    if (sp_ && g_kpp.vdc_same) hipLaunchKernelGGL((k_kpp_blmix<true, true, true>), GC, BC, 0, st, g, sp, g_kpp, (const double *)DBLOC, s.STF[0], s.STF[1], VISC, s.VDC[0], s.VDC[1], s.KPP_SRC[0], s.KPP_SRC[1], s.HBLT);
    else if (sp_) hipLaunchKernelGGL((k_kpp_blmix<true, false, true>), GC, BC, 0, st, g, sp, g_kpp, (const double *)DBLOC, s.STF[0], s.STF[1], VISC, s.VDC[0], s.VDC[1], s.KPP_SRC[0], s.KPP_SRC[1], s.HBLT);
    else if (g_kpp.vdc_same) hipLaunchKernelGGL((k_kpp_blmix<true, true>), GC, BC, 0, st, g, sp, g_kpp, (const double *)DBLOC, s.STF[0], s.STF[1], VISC, s.VDC[0], s.VDC[1], s.KPP_SRC[0], s.KPP_SRC[1], s.HBLT);
    else hipLaunchKernelGGL((k_kpp_blmix<true, false>), GC, BC, 0, st, g, sp, g_kpp, (const double *)DBLOC, s.STF[0], s.STF[1], VISC, s.VDC[0], s.VDC[1], s.KPP_SRC[0], s.KPP_SRC[1], s.HBLT);
    launch_kpp_vvc(g, h, st, (const double *)VISC, s.VVC);
    if (hipGetLastError() != hipSuccess) { err = "KPP kernel launch failed"; return 1; }
    return 0;
  }
  if (g.pbc) {
    // partial bottom cells (round 3): the 3-D-parallel / scratch-staged kernel forms carry the PBC branches; every level of the
    // surface-layer buoyancy difference and of the shear is formed (no on-demand march)
    g_kpp.WUK = nullptr; g_kpp.CONVB = nullptr;
    hipLaunchKernelGGL(k_kpp_ushear<true>, G3, dim3(256), 0, st, g, g_kpp, s.UMIX, s.VMIX, WU);
    hipLaunchKernelGGL(k_kpp_buoydiff<true>, G3, dim3(256), 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], DBLOC, DBSFC);
    hipLaunchKernelGGL(k_kpp_interior<true>, GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], s.UMIX, s.VMIX, (const double *)DBLOC, RIW, VISC, s.VDC[0], s.VDC[1]);
    hipLaunchKernelGGL((k_kpp_bldepth<false, 20, true>), GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], s.STF[0], s.STF[1], (const double *)DBLOC,
                       (const double *)DBSFC, (const double *)WU, s.UMIX, s.VMIX);
    if (g_kpp.vdc_same) hipLaunchKernelGGL((k_kpp_blmix<true, true>), GC, BC, 0, st, g, sp, g_kpp, (const double *)DBLOC, s.STF[0], s.STF[1], VISC, s.VDC[0], s.VDC[1],
                                            s.KPP_SRC[0], s.KPP_SRC[1], s.HBLT);
    else hipLaunchKernelGGL((k_kpp_blmix<true, false>), GC, BC, 0, st, g, sp, g_kpp, (const double *)DBLOC, s.STF[0], s.STF[1], VISC, s.VDC[0], s.VDC[1],
                            s.KPP_SRC[0], s.KPP_SRC[1], s.HBLT);
    if (h.c.kpp_ml_diagnostics == 1 && s.HMXL && s.HMXL_DR)   // DBSFC holds every level here (the diagnostics switch the on-demand march off)
      hipLaunchKernelGGL(k_kpp_hmxl, GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], (const double *)DBSFC, s.HMXL, s.HMXL_DR);
    launch_kpp_vvc(g, h, st, (const double *)VISC, s.VVC);
    if (hipGetLastError() != hipSuccess) { err = "KPP kernel launch failed"; return 1; }
    return 0;
  }
  if (KH.side) { hipEventRecord(KH.ev_fork, st); hipStreamWaitEvent(KH.side, KH.ev_fork, 0); }
  if (g_kpp_col & 1) hipLaunchKernelGGL(k_kpp_ushear_col<24>, GC, BC, 0, su, g, g_kpp, s.UMIX, s.VMIX, WU);
  else hipLaunchKernelGGL(k_kpp_ushear<false>, G3, dim3(256), 0, su, g, g_kpp, s.UMIX, s.VMIX, WU);
  // two waves per SIMD (<= 256 VGPRs, ~80 spilled) beat one wave with everything in registers: the kernel is VALU-bound
  // and a second wave fills the division / dependency stalls of the first (POP_KPP_BUOY_WAVES=1 keeps one wave)
  const int bw = tun_or(h.tun.kpp_buoy_waves, 2);
  g_kpp.CONVB = (march && g.km <= 64 && !tun_off(h.tun.kpp_sparse)) ? KH.convb : nullptr;
  if (march) hipLaunchKernelGGL(k_kpp_buoy_interior_march<false>, GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], s.UMIX, s.VMIX, DBLOC, VISC, s.VDC[0], s.VDC[1]);
  else if (lazy && fused_bi) hipLaunchKernelGGL((k_kpp_buoy_interior_lds<20, 8, false>), GL, dim3(POP_COL_THREADS, 8), 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], s.UMIX, s.VMIX, DBLOC, DBSFC, VISC, s.VDC[0], s.VDC[1]);
  else if (fused_bi) hipLaunchKernelGGL((k_kpp_buoy_interior_lds<20, 8>), GL, dim3(POP_COL_THREADS, 8), 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], s.UMIX, s.VMIX, DBLOC, DBSFC, VISC, s.VDC[0], s.VDC[1]);
  else if ((g_kpp_col & 4) && KH.max_kref <= 20 && g.xcd_remap != 2) hipLaunchKernelGGL((k_kpp_buoydiff_lds<20, 4>), GL, BL, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], DBLOC, DBSFC);
  else if ((g_kpp_col & 4) && KH.max_kref <= 28 && g.xcd_remap != 2) hipLaunchKernelGGL((k_kpp_buoydiff_lds<28, 4>), GL, BL, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], DBLOC, DBSFC);
  else if ((g_kpp_col & 2) && KH.max_kref <= 20 && bw == 2) hipLaunchKernelGGL((k_kpp_buoydiff_col<20, 2>), GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], DBLOC, DBSFC);
  else if ((g_kpp_col & 2) && KH.max_kref <= 20) hipLaunchKernelGGL((k_kpp_buoydiff_col<20, 1>), GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], DBLOC, DBSFC);
  else if ((g_kpp_col & 2) && bw == 2) hipLaunchKernelGGL((k_kpp_buoydiff_col<24, 2>), GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], DBLOC, DBSFC);
  else if (g_kpp_col & 2) hipLaunchKernelGGL((k_kpp_buoydiff_col<24, 1>), GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], DBLOC, DBSFC);
  else if (lazy) hipLaunchKernelGGL(k_kpp_buoydiff<false>, G3, dim3(256), 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], DBLOC, DBSFC);
  else hipLaunchKernelGGL(k_kpp_buoydiff<true>, G3, dim3(256), 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], DBLOC, DBSFC);
  // the boundary-layer depth (needs buoydiff + shear, writes only the 2-D boundary-layer fields) follows the shear kernel
  // on the side stream and runs beside the interior coefficients; blmix waits for both
  if (KH.side) {
    hipEventRecord(KH.ev_bd, st); hipStreamWaitEvent(KH.side, KH.ev_bd, 0);
    if (lazy20) hipLaunchKernelGGL((k_kpp_bldepth<true, 20>), GC, BC, 0, KH.side, g, g_kpp, s.TMIX[0], s.TMIX[1], s.STF[0], s.STF[1], (const double *)DBLOC,
                                   (const double *)DBSFC, (const double *)WU, s.UMIX, s.VMIX);
    else if (lazy) hipLaunchKernelGGL((k_kpp_bldepth<true, 28>), GC, BC, 0, KH.side, g, g_kpp, s.TMIX[0], s.TMIX[1], s.STF[0], s.STF[1], (const double *)DBLOC,
                                      (const double *)DBSFC, (const double *)WU, s.UMIX, s.VMIX);
    else hipLaunchKernelGGL((k_kpp_bldepth<false, 20>), GC, BC, 0, KH.side, g, g_kpp, s.TMIX[0], s.TMIX[1], s.STF[0], s.STF[1], (const double *)DBLOC,
                            (const double *)DBSFC, (const double *)WU, s.UMIX, s.VMIX);
    hipEventRecord(KH.ev_join, KH.side);
  }
  const bool int_reg = !tun_on(h.tun.kpp_interior_generic);
  if (fused_bi) {}
  else if (int_reg && g.km == 60) hipLaunchKernelGGL(k_kpp_interior_reg<60>, GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], s.UMIX, s.VMIX, (const double *)DBLOC, VISC, s.VDC[0], s.VDC[1]);
  else if (int_reg && g.km == 62) hipLaunchKernelGGL(k_kpp_interior_reg<62>, GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], s.UMIX, s.VMIX, (const double *)DBLOC, VISC, s.VDC[0], s.VDC[1]);
  else hipLaunchKernelGGL(k_kpp_interior<false>, GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], s.UMIX, s.VMIX, (const double *)DBLOC, RIW, VISC, s.VDC[0], s.VDC[1]);
  if (KH.side) hipStreamWaitEvent(st, KH.ev_join, 0);
  else if (lazy20) hipLaunchKernelGGL((k_kpp_bldepth<true, 20>), GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], s.STF[0], s.STF[1], (const double *)DBLOC,
                                      (const double *)DBSFC, (const double *)WU, s.UMIX, s.VMIX);
  else if (lazy) hipLaunchKernelGGL((k_kpp_bldepth<true, 28>), GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], s.STF[0], s.STF[1], (const double *)DBLOC,
                                    (const double *)DBSFC, (const double *)WU, s.UMIX, s.VMIX);
  else hipLaunchKernelGGL((k_kpp_bldepth<false, 20>), GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], s.STF[0], s.STF[1], (const double *)DBLOC,
                          (const double *)DBSFC, (const double *)WU, s.UMIX, s.VMIX);
  const bool sparse = g_kpp.CONVB != nullptr;
  if (sparse && g_kpp.vdc_same) hipLaunchKernelGGL((k_kpp_blmix<false, true, true>), GC, BC, 0, st, g, sp, g_kpp, (const double *)DBLOC, s.STF[0], s.STF[1], VISC, s.VDC[0], s.VDC[1],
                                                   s.KPP_SRC[0], s.KPP_SRC[1], s.HBLT);
  else if (sparse) hipLaunchKernelGGL((k_kpp_blmix<false, false, true>), GC, BC, 0, st, g, sp, g_kpp, (const double *)DBLOC, s.STF[0], s.STF[1], VISC, s.VDC[0], s.VDC[1],
                                      s.KPP_SRC[0], s.KPP_SRC[1], s.HBLT);
  else if (g_kpp.vdc_same) hipLaunchKernelGGL((k_kpp_blmix<false, true>), GC, BC, 0, st, g, sp, g_kpp, (const double *)DBLOC, s.STF[0], s.STF[1], VISC, s.VDC[0], s.VDC[1],
                                          s.KPP_SRC[0], s.KPP_SRC[1], s.HBLT);
  else hipLaunchKernelGGL((k_kpp_blmix<false, false>), GC, BC, 0, st, g, sp, g_kpp, (const double *)DBLOC, s.STF[0], s.STF[1], VISC, s.VDC[0], s.VDC[1],
                          s.KPP_SRC[0], s.KPP_SRC[1], s.HBLT);
  // large grids: 64 x 4 patches (the row j + 1 of the four-point average is read by the same workgroup; 64 x 2 / 8 / 16 measured: vmix 7.54 / 7.65 / 7.97 ms against 7.57)
  launch_kpp_vvc(g, h, st, (const double *)VISC, s.VVC);
  if (h.c.kpp_ml_diagnostics == 1 && s.HMXL && s.HMXL_DR)
    hipLaunchKernelGGL(k_kpp_hmxl, GC, BC, 0, st, g, g_kpp, s.TMIX[0], s.TMIX[1], (const double *)DBSFC, s.HMXL, s.HMXL_DR);
  if (hipGetLastError() != hipSuccess) { err = "KPP kernel launch failed"; return 1; }
  return 0;
}

}  // namespace pop
