// kernels_thomas_reg.hpp -- implicit vertical mixing with the whole water column in registers.
//
// The generic Thomas kernels (kernels_baroclinic.hpp) stage the elimination coefficients E and the
// partial solution F through scratch fields in HBM and re-read the column three to four times.
// For the production level counts (km = 60: gx3v7 / gx1v7, km = 62: tx0.1v3) these kernels are
// compiled with KM as a template constant: every level loop is fully unrolled, E/F/U/V live in
// VGPRs (<= 512 per lane on gfx950 at one wave per SIMD), all column loads are issued up front
// (deep memory-level parallelism instead of occupancy), and each field is read once and written
// once -- the algorithmic traffic of SURVEY.md 8(d) phases C, F and G.
// Arithmetic and evaluation order are identical to the generic kernels (vertical_mix.F90:1263-1368,
// 1563-1658, 1762-1868; baroclinic.F90:1077-1129, 1418-1475).
#pragma once
#include "kernels_baroclinic.hpp"

namespace pop {

// the corrector form (MODE 1) needs ~260 VGPRs: capped at 256 it runs two waves per SIMD
// PBC (r3): partial bottom cells -- from level 2 on the column's own thicknesses (vertical_mix.F90:1279-1287, 1577-1585), which
// differ from dz(k) at level KMT only: two selects per level on top of the flat-bottom kernel, same expressions as k_impvmixt<.,.,.,true>
template <int KM, int MODE, bool PRE, bool POST, bool PBC = false>
__global__ void __launch_bounds__(POP_COL_THREADS, (MODE == 1) ? 2 : 1)
k_impvmixt_reg(DevGrid g, StepParams sp, ImpvmixtArgs a) {
  Col c;
  if (!col_setup(g, c, true)) return;
  const long long n2 = g.n2;
  const int kmt = g.KMT[c.q2];
  const double dzbc = PBC ? g.DZBC[c.q2] : 0.0;
  const double hfac1 = g.dz[1] / a.c2dtt;
  const double H1 = hfac1 + a.PSFC[c.q2] / (sp.grav * a.c2dtt);
  const long long vdcbase = ((long long)c.b * (KM + 2)) * n2 + c.p2;
  double Ea[KM], Fa[KM];   // VDC -> E and TNEW -> F, in registers
  // the value the increment is added to -- TOLD (MODE 0) or the incoming TNEW (MODE 1) -- in a third register column, loaded with
  // the others: read inside the back substitution (round 2) the compiler issued these loads one by one behind the recurrence, each
  // with its own s_waitcnt vmcnt(0): 40 exposed memory latencies per column (ISA of k_impvmixt_reg<62,0>); one wave per SIMD has
  // 512 registers, the three columns take 372.  (The corrector form, capped at 256 registers for two waves per SIMD, keeps reading
  // its base value -- the incoming TNEW -- in the back substitution.)
  double Ba[MODE == 0 ? KM : 1];
  {
    const int n = a.nfirst - 1 + blockIdx.z;            // one tracer per thread (launch z = tracer count)
    double *__restrict__ const TN = a.TNEW[n];
    const double *__restrict__ const VDC = a.VDC[n];
    const double *__restrict__ const TO = a.TOLD[n];
    // all column loads up front
#pragma unroll
    for (int k = 1; k <= KM; ++k) {
      Ea[k - 1] = VDC[vdcbase + (long long)k * n2];
      Fa[k - 1] = TN[c.base3 + (long long)(k - 1) * n2];
      if (MODE == 0) Ba[k - 1] = TO[c.base3 + (long long)(k - 1) * n2];
    }
    double rhs1 = 0.0;
    if (MODE == 1) {
      if (kmt > 0)
        rhs1 = ((2.0 * a.TCUR[n][c.base3] - TO[c.base3]) * (a.PCUR[c.q2] - a.POLD[c.q2]) -
                Fa[0] * (a.PNEW[c.q2] - a.PCUR[c.q2])) / (sp.grav * g.dz[1]);
    }
    double t1 = Fa[0];
    if (PRE) {
      if (kmt > 0) t1 = t1 - TO[c.base3] * (a.PNEW[c.q2] - a.PMIX[c.q2]) / (sp.grav * g.dz[1]);
    }
    double A = g.afac_t[1] * Ea[0];
    double D = H1 + A;
    double Ek = A / D;
    double B = H1 * Ek;
    double Fk = (MODE == 1) ? hfac1 * rhs1 / D : hfac1 * t1 / D;
    double tn1 = Fa[0];     // TNEW(1) before the update (MODE 1 adds the increment to it)
    Ea[0] = Ek; Fa[0] = Fk;
    (void)tn1;
#pragma unroll
    for (int k = 2; k <= KM; ++k) {
      const double C = A;
      double hf = g.dz[k] / a.c2dtt;
      if (PBC) {
        const double dzt = pbc_dz(g, k, kmt, dzbc);
        A = sp.aidif * Ea[k - 1] / (0.5 * (dzt + pbc_dz(g, k + 1, kmt, dzbc)));
        hf = dzt / a.c2dtt;
      } else A = g.afac_t[k] * Ea[k - 1];
      const double tn = Fa[k - 1];
      if (k > kmt) { Fk = 0.0; }
      else {
        D = (k == kmt) ? hf + B : hf + A + B;
        Ek = A / D;
        B = (hf + B) * Ek;
        Fk = (MODE == 1) ? C * Fk / D : (hf * tn + C * Fk) / D;
        Ea[k - 1] = Ek;
      }
      Fa[k - 1] = Fk;
    }
    double Fkp1 = 0.0;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = KM; k >= 1; --k) {
      double f = Fa[k - 1];
      if (k < KM && k < kmt) f = f + Ea[k - 1] * Fkp1;
      Fkp1 = f;
      double tn;
      if (MODE == 0) tn = Ba[k - 1] + f;                  // base value TOLD: in registers
      else {
        // the incoming TNEW, read here; a compiler barrier every 8 levels bounds how many of these loads are in flight
        if ((k & 7) == 0) asm volatile("" ::: "memory");
        tn = TN[c.base3 + (long long)(k - 1) * n2] + f;
      }
      if (POST && n == 0 && k == 1 && sp.reset_to_freezing) tn = fmax(tn, -2.0);
      Fa[k - 1] = tn;
    }
    // one store burst after the last load: loads and stores retire through the same in-order counter,
    // so a store between two loads would make the second load's wait cover the store's round trip
#pragma unroll
    for (int k = 1; k <= KM; ++k) TN[c.base3 + (long long)(k - 1) * n2] = Fa[k - 1];
  }
}

// Both tracers of a column in one thread, for the case that they share one diffusivity array (KPP without double diffusion,
// a.VDC[0] == a.VDC[1]): the elimination coefficients E(k) depend on VDC, dz, dt and the surface pressure only, so they are formed
// once and VDC is read once -- 7 instead of 8 fields through HBM in the predictor, 5 instead of 6 in the corrector, 3 divisions per
// level instead of 4.  Each tracer's values go through exactly the operations of k_impvmixt_reg (E is the same number in both
// threads there), so TNEW is bitwise unchanged.  Three register columns (E, F_T, F_S): one wave per SIMD.
template <int KM, int MODE, bool PRE, bool POST, bool PBC = false>
__global__ void __launch_bounds__(POP_COL_THREADS, 1)
k_impvmixt2_reg(DevGrid g, StepParams sp, ImpvmixtArgs a) {
  Col c;
  if (!col_setup(g, c, true)) return;
  const long long n2 = g.n2;
  const int kmt = g.KMT[c.q2];
  const double dzbc = PBC ? g.DZBC[c.q2] : 0.0;
  const double hfac1 = g.dz[1] / a.c2dtt;
  const double H1 = hfac1 + a.PSFC[c.q2] / (sp.grav * a.c2dtt);
  const long long vdcbase = ((long long)c.b * (KM + 2)) * n2 + c.p2;
  double Ea[KM], F0[KM], F1[KM];
  double *__restrict__ const TN0 = a.TNEW[0];
  double *__restrict__ const TN1 = a.TNEW[1];
  const double *__restrict__ const VDC = a.VDC[0];
  const double *__restrict__ const TO0 = a.TOLD[0];
  const double *__restrict__ const TO1 = a.TOLD[1];
#pragma unroll
  for (int k = 1; k <= KM; ++k) {
    Ea[k - 1] = VDC[vdcbase + (long long)k * n2];
    F0[k - 1] = TN0[c.base3 + (long long)(k - 1) * n2];
    F1[k - 1] = TN1[c.base3 + (long long)(k - 1) * n2];
  }
  double rhs0 = 0.0, rhs1 = 0.0;
  if (MODE == 1) {
    if (kmt > 0) {
      rhs0 = ((2.0 * a.TCUR[0][c.base3] - TO0[c.base3]) * (a.PCUR[c.q2] - a.POLD[c.q2]) - F0[0] * (a.PNEW[c.q2] - a.PCUR[c.q2])) / (sp.grav * g.dz[1]);
      rhs1 = ((2.0 * a.TCUR[1][c.base3] - TO1[c.base3]) * (a.PCUR[c.q2] - a.POLD[c.q2]) - F1[0] * (a.PNEW[c.q2] - a.PCUR[c.q2])) / (sp.grav * g.dz[1]);
    }
  }
  double t0 = F0[0], t1 = F1[0];
  if (PRE) {
    if (kmt > 0) {
      t0 = t0 - TO0[c.base3] * (a.PNEW[c.q2] - a.PMIX[c.q2]) / (sp.grav * g.dz[1]);
      t1 = t1 - TO1[c.base3] * (a.PNEW[c.q2] - a.PMIX[c.q2]) / (sp.grav * g.dz[1]);
    }
  }
  double A = g.afac_t[1] * Ea[0];
  double D = H1 + A;
  double Ek = A / D;
  double B = H1 * Ek;
  double Fk0 = (MODE == 1) ? hfac1 * rhs0 / D : hfac1 * t0 / D;
  double Fk1 = (MODE == 1) ? hfac1 * rhs1 / D : hfac1 * t1 / D;
  Ea[0] = Ek; F0[0] = Fk0; F1[0] = Fk1;
#pragma unroll
  for (int k = 2; k <= KM; ++k) {
    const double C = A;
    double hf = g.dz[k] / a.c2dtt;
    if (PBC) {
      const double dzt = pbc_dz(g, k, kmt, dzbc);
      A = sp.aidif * Ea[k - 1] / (0.5 * (dzt + pbc_dz(g, k + 1, kmt, dzbc)));
      hf = dzt / a.c2dtt;
    } else A = g.afac_t[k] * Ea[k - 1];
    const double tn0 = F0[k - 1], tn1 = F1[k - 1];
    if (k > kmt) { Fk0 = 0.0; Fk1 = 0.0; }
    else {
      D = (k == kmt) ? hf + B : hf + A + B;
      Ek = A / D;
      B = (hf + B) * Ek;
      Fk0 = (MODE == 1) ? C * Fk0 / D : (hf * tn0 + C * Fk0) / D;
      Fk1 = (MODE == 1) ? C * Fk1 / D : (hf * tn1 + C * Fk1) / D;
      Ea[k - 1] = Ek;
    }
    F0[k - 1] = Fk0; F1[k - 1] = Fk1;
  }
  double Fp0 = 0.0, Fp1 = 0.0;
  asm volatile("" ::: "memory");
#pragma unroll
  for (int k = KM; k >= 1; --k) {
    double f0 = F0[k - 1], f1 = F1[k - 1];
    if (k < KM && k < kmt) { f0 = f0 + Ea[k - 1] * Fp0; f1 = f1 + Ea[k - 1] * Fp1; }
    Fp0 = f0; Fp1 = f1;
    if ((k & 7) == 0) asm volatile("" ::: "memory");
    const long long ob = c.base3 + (long long)(k - 1) * n2;
    double x0 = ((MODE == 1) ? TN0[ob] : TO0[ob]) + f0;
    const double x1 = ((MODE == 1) ? TN1[ob] : TO1[ob]) + f1;
    if (POST && k == 1 && sp.reset_to_freezing) x0 = fmax(x0, -2.0);
    F0[k - 1] = x0; F1[k - 1] = x1;
  }
#pragma unroll
  for (int k = 1; k <= KM; ++k) { TN0[c.base3 + (long long)(k - 1) * n2] = F0[k - 1]; TN1[c.base3 + (long long)(k - 1) * n2] = F1[k - 1]; }
}

// one velocity component per thread (blockIdx.z = 0: U, 1: V): the two solves share only the
// elimination coefficients, which each thread recomputes, so the column fits two register arrays
// and the launch has twice the waves
// WAVES = 2 caps the kernel at 256 VGPRs (38 - 64 spilled): faster on small, latency-bound grids (gx1v7: 0.111 -> 0.099 ms),
// slower on bandwidth-bound ones (tx0.1v3: 6.5 -> 6.9 ms), so the launcher picks by grid size
// ADD: the step tail's "add the barotropic velocity where k <= KMU" (step_mod.F90:572-592, k_add_barotropic) applied to the value
// on its way out -- the same sum (X - mean) + UBTROP, so U, V(new) are bitwise what the two launches leave, and one read and one
// write of both 3-D fields are gone.  Only valid once the barotropic solve of the step has finished (pop_amd.hip: deferred form).
// PBC (r3): the U cells' own thicknesses, DZU = dz except DZUB at level KMU (vertical_mix.F90:1777-1785; baroclinic.F90:1097-1106)
template <int KM, int WAVES, bool ADD = false, bool PBC = false>
__global__ void __launch_bounds__(POP_COL_THREADS, WAVES)
k_impvmixu_reg(DevGrid g, StepParams sp, ImpvmixuArgs a) {
  Col c;
  if (!col_setup(g, c, true)) return;
  const long long n2 = g.n2;
  const int kmu = g.KMU[c.q2];
  const double hur = g.HUR[c.q2];
  const double dzub = PBC ? g.DZUB[c.q2] : 0.0;
  const bool isv = (blockIdx.z == 1);
  double *__restrict__ const XN = isv ? a.VNEW : a.UNEW;
  const double *__restrict__ const XO = isv ? a.VOLD : a.UOLD;
  const double *__restrict__ const VVC = a.VVC;
  double Xa[KM], Ea[KM];
#pragma unroll
  for (int k = 1; k <= KM; ++k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    Xa[k - 1] = XN[o]; Ea[k - 1] = VVC[o];
  }
  const double hf1 = g.dz[1] / sp.c2dtu;
  double A = g.afac_u[1] * Ea[0];
  double D = hf1 + A;
  double Ek = A / D;
  double B = hf1 * Ek;
  double F1 = hf1 * Xa[0] / D;
  Ea[0] = Ek; Xa[0] = F1;
#pragma unroll
  for (int k = 2; k <= KM; ++k) {
    const double C = A;
    double hf = g.dz[k] / sp.c2dtu;
    if (PBC) {
      // the thicknesses depend on (kmu, dzub) only, so the scheduler would form all 62 selections at the top of the unrolled
      // column and keep them alive (460 B of scratch): tied to the recurrence, each is formed where it is used
      double dzub_k = dzub;
      asm volatile("" : "+v"(dzub_k) : "v"(B));
      const double dzu = pbc_dz(g, k, kmu, dzub_k);
      hf = dzu / sp.c2dtu;
      A = sp.aidif * Ea[k - 1] / (0.5 * (dzu + pbc_dz(g, k + 1, kmu, dzub_k)));
    } else A = g.afac_u[k] * Ea[k - 1];
    if (k <= kmu) {
      D = (k < kmu) ? hf + A + B : hf + B;
      Ek = A / D;
      B = (hf + B) * Ek;
      F1 = (hf * Xa[k - 1] + C * F1) / D;
      Ea[k - 1] = Ek;
    } else { F1 = 0.0; }
    Xa[k - 1] = F1;
  }
  double F1p = 0.0;
  asm volatile("" ::: "memory");
#pragma unroll
  for (int k = KM; k >= 1; --k) {
    double f1 = Xa[k - 1];
    if (k < KM && k < kmu) f1 = f1 + Ea[k - 1] * F1p;
    F1p = f1;
    if ((k & 7) == 0) asm volatile("" ::: "memory");     // bound the old-velocity loads in flight
    Xa[k - 1] = XO[c.base3 + (long long)(k - 1) * n2] + f1;
  }
  double w1 = 0.0;
#pragma unroll
  for (int k = 1; k <= KM; ++k) {
    if (PBC) {
      double dzub_k = dzub;
      asm volatile("" : "+v"(dzub_k) : "v"(w1));       // as above: the selection formed where it is used
      w1 = w1 + Xa[k - 1] * pbc_dz(g, k, kmu, dzub_k);
    } else w1 = w1 + Xa[k - 1] * g.dz[k];
  }
  w1 = w1 * hur;
#pragma unroll
  for (int k = 1; k <= KM; ++k) Xa[k - 1] = (k <= kmu) ? Xa[k - 1] - w1 : 0.0;
  if (ADD) {
    const double xb = (isv ? a.VB : a.UB)[c.q2];
#pragma unroll
    for (int k = 1; k <= KM; ++k) if (k <= kmu) Xa[k - 1] = Xa[k - 1] + xb;
  }
#pragma unroll
  for (int k = 1; k <= KM; ++k) XN[c.base3 + (long long)(k - 1) * n2] = Xa[k - 1];
}

// dispatch on the level count: register kernels for the production grids, generic otherwise
template <int MODE, bool PRE, bool POST>
inline void launch_impvmixt(const DevGrid &g, const StepParams &sp, const ImpvmixtArgs &a, dim3 G, hipStream_t st, bool allow_reg, int pair_tuning) {
  const dim3 B(POP_COL_THREADS);
  const dim3 G2(G.x, G.y, a.nlast - a.nfirst + 1);
  // both tracers in one thread when they share the diffusivity array (pop_tuning.thomas_pair = 0 | 1 overrides the size rule)
  const int pair_env = tun_or(pair_tuning, -1);
  // (corrector form only: the predictor's three full register columns + its up-front loads spill ~ 900 B per lane)
  if (g.pbc) {   // partial bottom cells: the register kernels' PBC instantiations at km = 60 / 62, else the scratch-staged kernel
    const bool reg = allow_reg && (g.km == 60 || g.km == 62);
    const bool pairp = MODE == 1 && reg && a.nfirst == 1 && a.nlast == 2 && a.VDC[0] == a.VDC[1] &&
                       (pair_env >= 0 ? pair_env != 0 : (long long)g.n2 * g.nblocks > (1 << 19));
    if (!reg) { hipLaunchKernelGGL((k_impvmixt<MODE, PRE, POST, true>), G, B, 0, st, g, sp, a); return; }
    if (pairp) {
      if constexpr (MODE == 1) {
        if (g.km == 60) hipLaunchKernelGGL((k_impvmixt2_reg<60, MODE, PRE, POST, true>), G, B, 0, st, g, sp, a);
        else hipLaunchKernelGGL((k_impvmixt2_reg<62, MODE, PRE, POST, true>), G, B, 0, st, g, sp, a);
      }
    } else if (g.km == 60) hipLaunchKernelGGL((k_impvmixt_reg<60, MODE, PRE, POST, true>), G2, B, 0, st, g, sp, a);
    else hipLaunchKernelGGL((k_impvmixt_reg<62, MODE, PRE, POST, true>), G2, B, 0, st, g, sp, a);
    if (POST) launch_state3d(g, a.TNEW[0], a.TNEW[1], a.RHO, st);
    return;
  }
  const bool pair = MODE == 1 && allow_reg && (g.km == 60 || g.km == 62) && a.nfirst == 1 && a.nlast == 2 && a.VDC[0] == a.VDC[1] &&
                    (pair_env >= 0 ? pair_env != 0 : (long long)g.n2 * g.nblocks > (1 << 19));
  if (pair) {
    if constexpr (MODE == 1) {
      if (g.km == 60) hipLaunchKernelGGL((k_impvmixt2_reg<60, MODE, PRE, POST>), G, B, 0, st, g, sp, a);
      else hipLaunchKernelGGL((k_impvmixt2_reg<62, MODE, PRE, POST>), G, B, 0, st, g, sp, a);
    }
    if (POST) launch_state3d(g, a.TNEW[0], a.TNEW[1], a.RHO, st);
  } else if (allow_reg && (g.km == 60 || g.km == 62)) {
    if (g.km == 60) hipLaunchKernelGGL((k_impvmixt_reg<60, MODE, PRE, POST>), G2, B, 0, st, g, sp, a);
    else hipLaunchKernelGGL((k_impvmixt_reg<62, MODE, PRE, POST>), G2, B, 0, st, g, sp, a);
    // the density of the finished tracers (baroclinic.F90:1468-1475) as its own 3-D-parallel pass
    if (POST) launch_state3d(g, a.TNEW[0], a.TNEW[1], a.RHO, st);
  } else hipLaunchKernelGGL((k_impvmixt<MODE, PRE, POST>), G, B, 0, st, g, sp, a);
}
inline void launch_impvmixu(const DevGrid &g, const StepParams &sp, const ImpvmixuArgs &a, dim3 G, hipStream_t st, bool allow_reg) {
  const dim3 B(POP_COL_THREADS);
  const dim3 G2(G.x, G.y, 2);
  const bool small = (long long)g.n2 * g.nblocks <= (1 << 19);
  if (g.pbc) {
    if (allow_reg && g.km == 60) hipLaunchKernelGGL((k_impvmixu_reg<60, 1, false, true>), G2, B, 0, st, g, sp, a);
    else if (allow_reg && g.km == 62) hipLaunchKernelGGL((k_impvmixu_reg<62, 1, false, true>), G2, B, 0, st, g, sp, a);
    else hipLaunchKernelGGL(k_impvmixu_norm<true>, G, B, 0, st, g, sp, a);
    return;
  }
  if (allow_reg && g.km == 60 && small) hipLaunchKernelGGL((k_impvmixu_reg<60, 2>), G2, B, 0, st, g, sp, a);
  else if (allow_reg && g.km == 60) hipLaunchKernelGGL((k_impvmixu_reg<60, 1>), G2, B, 0, st, g, sp, a);
  else if (allow_reg && g.km == 62 && small) hipLaunchKernelGGL((k_impvmixu_reg<62, 2>), G2, B, 0, st, g, sp, a);
  else if (allow_reg && g.km == 62) hipLaunchKernelGGL((k_impvmixu_reg<62, 1>), G2, B, 0, st, g, sp, a);
  else hipLaunchKernelGGL(k_impvmixu_norm<false>, G, B, 0, st, g, sp, a);
}
// the register kernel with the barotropic velocity added on the way out (a.UB, a.VB set); km = 60 / 62 only
inline bool impvmixu_add_available(const DevGrid &g, bool allow_reg) { return allow_reg && (g.km == 60 || g.km == 62); }
inline void launch_impvmixu_add(const DevGrid &g, const StepParams &sp, const ImpvmixuArgs &a, dim3 G, hipStream_t st) {
  const dim3 B(POP_COL_THREADS);
  const dim3 G2(G.x, G.y, 2);
  const bool small = (long long)g.n2 * g.nblocks <= (1 << 19);
  if (g.pbc) {
    if (g.km == 60) hipLaunchKernelGGL((k_impvmixu_reg<60, 1, true, true>), G2, B, 0, st, g, sp, a);
    else hipLaunchKernelGGL((k_impvmixu_reg<62, 1, true, true>), G2, B, 0, st, g, sp, a);
    return;
  }
  if (g.km == 60 && small) hipLaunchKernelGGL((k_impvmixu_reg<60, 2, true>), G2, B, 0, st, g, sp, a);
  else if (g.km == 60) hipLaunchKernelGGL((k_impvmixu_reg<60, 1, true>), G2, B, 0, st, g, sp, a);
  else if (small) hipLaunchKernelGGL((k_impvmixu_reg<62, 2, true>), G2, B, 0, st, g, sp, a);
  else hipLaunchKernelGGL((k_impvmixu_reg<62, 1, true>), G2, B, 0, st, g, sp, a);
}

}  // namespace pop
