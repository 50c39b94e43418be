// kernels_baroclinic.hpp -- HIP kernels of the baroclinic step (gfx950).
//
// Mapping: one thread owns one water column (i,j,block) and marches k with all k-carried
// quantities (vertical velocity, vertical fluxes, hydrostatic pressure sums) in registers --
// the reference keeps these in module-level save arrays (advection.F90:96-97,
// vertical_mix.F90:131-135, pressure_grad.F90:56-58).  Lanes of a wavefront are 64 consecutive
// i, so every level read is one fully coalesced 512-byte request; horizontal neighbours are
// re-read through L1/L2.  All kernels are HBM-bound (no dense contraction -> no MFMA).
// Arithmetic is written in the reference's evaluation order and compiled with
// -ffp-contract=off so results match the CPU restatement to rounding of the library
// functions only.
#pragma once
#include "kernels_common.hpp"

namespace pop {

// ------------------------------------------------------------------------------------------
// dhdt  (surface_hgt.F90:208-286) + tgrid_to_ugrid (grid.F90:3399-3413)
// ------------------------------------------------------------------------------------------
__global__ void k_dhdt(DevGrid g, StepParams sp, const double *__restrict__ PC, const double *__restrict__ PO,
                       const double *__restrict__ FW_OLD, double *__restrict__ DH, double *__restrict__ DHU) {
  Col c;
  if (!col_setup(g, c, false)) return;
  auto dh = [&](long long q) { return (PC[q] - PO[q]) / (sp.grav * sp.dtp) - FW_OLD[q]; };
  DH[c.q2] = dh(c.q2);
  double u = 0.0;
  if (c.i < g.nxb - 1 && c.j < g.nyb - 1) {
    u = g.AU0[c.q2] * dh(c.q2) + g.AUN[c.q2] * dh(c.q2 + g.nxb) + g.AUE[c.q2] * dh(c.q2 + 1) + g.AUNE[c.q2] * dh(c.q2 + g.nxb + 1);
    if (!(g.KMU[c.q2] >= 1)) u = 0.0;
  }
  DHU[c.q2] = u;
}

// ------------------------------------------------------------------------------------------
// state over whole 3-D arrays (state_mod.F90:258-498): RHO = rho(T,S,pressz(k)).  3-D parallel.
// ------------------------------------------------------------------------------------------
// the same on the cells p2_first <= p2 < p2_end of every level (a band of rows: rim / interior split around a halo exchange)
__global__ void k_state3d_rows(DevGrid g, const double *__restrict__ T, const double *__restrict__ S, double *__restrict__ RHO,
                               int p2_first, int p2_end) {
  const int p2 = p2_first + blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y + 1, b = blockIdx.z;
  if (p2 >= p2_end) return;
  const long long o = (long long)b * g.n3 + (long long)(k - 1) * g.n2 + p2;
  const MwjfP P = mwjf_level(g.pressz[k]);
  RHO[o] = mwjf_rho<false>(P, T[o], S[o], nullptr, nullptr);
}
__global__ void k_state3d(DevGrid g, const double *__restrict__ T, const double *__restrict__ S, double *__restrict__ RHO) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y + 1, b = blockIdx.z;
  if (p2 >= g.n2) return;
  if (land_run(g, b, (long long)blockIdx.x * blockDim.x, blockDim.x)) return;
  const long long o = (long long)b * g.n3 + (long long)(k - 1) * g.n2 + p2;
  const MwjfP P = mwjf_level(g.pressz[k]);
  RHO[o] = mwjf_rho<false>(P, T[o], S[o], nullptr, nullptr);
}

// The pressure-dependent coefficients of every level as a table (6 doubles per level, slot 0 unused), formed by mwjf_level on
// the device so that a kernel reading them (scalar loads) holds the bits a kernel forming them in place holds.
__global__ void k_eos_level_table(DevGrid g, double *__restrict__ tab) {
  const int k = threadIdx.x;
  if (k < 1 || k > g.km) return;
  const MwjfP P = mwjf_level(g.pressz[k]);
  double *t = tab + 6 * k;
  t[0] = P.n0; t[1] = P.n2; t[2] = P.ns1t0; t[3] = P.d0; t[4] = P.d1; t[5] = P.d3;
}
// LK levels of one (i,j) per thread (round 3): k_state3d above spends a quarter of its vector instructions on the level's
// coefficients (one cell per lane to amortise them over) and as many scalar instructions as vector ones on its addressing; here
// the coefficients come from the table through scalar loads, the land test and the addressing are shared by LK cells and the
// 2 LK operand loads are in flight together.  Same operations per cell: bitwise k_state3d (tests/test_gpu_parity.py).
template <int LK>
__global__ void __launch_bounds__(256)
k_state3d_lv(DevGrid g, const double *__restrict__ T, const double *__restrict__ S, double *__restrict__ RHO) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int k0 = blockIdx.y * LK + 1, b = blockIdx.z;
  if (p2 >= g.n2) return;
  if (land_run(g, b, (long long)blockIdx.x * blockDim.x, blockDim.x)) return;
  const long long n2 = g.n2, o = (long long)b * g.n3 + (long long)(k0 - 1) * n2 + p2;
  double t[LK], s[LK];
#pragma unroll
  for (int u = 0; u < LK; ++u) {
    const bool in = k0 + u <= g.km;                                   // block-uniform
    t[u] = in ? T[o + u * n2] : 0.0; s[u] = in ? S[o + u * n2] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < LK; ++u) {
    if (k0 + u > g.km) break;
    const int e = 6 * (k0 + u);
    MwjfP P;
    P.n0 = g.eosP[e]; P.n2 = g.eosP[e + 1]; P.ns1t0 = g.eosP[e + 2]; P.d0 = g.eosP[e + 3]; P.d1 = g.eosP[e + 4]; P.d3 = g.eosP[e + 5];
    RHO[o + u * n2] = mwjf_rho<false>(P, t[u], s[u], nullptr, nullptr);
  }
}
inline void launch_state3d(const DevGrid &g, const double *T, const double *S, double *RHO, hipStream_t st) {
  const unsigned gx = (g.n2 + 255) / 256;
  if (g.state_lv == 4 && g.eosP) hipLaunchKernelGGL(k_state3d_lv<4>, dim3(gx, (g.km + 3) / 4, g.nblocks), dim3(256), 0, st, g, T, S, RHO);
  else if (g.state_lv == 2 && g.eosP) hipLaunchKernelGGL(k_state3d_lv<2>, dim3(gx, (g.km + 1) / 2, g.nblocks), dim3(256), 0, st, g, T, S, RHO);
  else if (g.state_lv == 8 && g.eosP) hipLaunchKernelGGL(k_state3d_lv<8>, dim3(gx, (g.km + 7) / 8, g.nblocks), dim3(256), 0, st, g, T, S, RHO);
  else hipLaunchKernelGGL(k_state3d, dim3(gx, g.km, g.nblocks), dim3(256), 0, st, g, T, S, RHO);
}

// ------------------------------------------------------------------------------------------
// vmix_coeffs_const with convection_type='diffusion' (vmix_const.F90:205-228).  3-D parallel.
// VDC is stored (nxb,nyb,0:km+1,block); VVC (nxb,nyb,km,block).
// ------------------------------------------------------------------------------------------
__global__ void k_vmix_const(DevGrid g, StepParams sp, const double *__restrict__ TM, const double *__restrict__ SM,
                             double *__restrict__ VDC, double *__restrict__ VVC) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y + 1, b = blockIdx.z;
  if (p2 >= g.n2) return;
  if (land_run(g, b, (long long)blockIdx.x * blockDim.x, blockDim.x)) return;
  const int kp1 = min(k + 1, g.km);
  const long long o = (long long)b * g.n3 + p2;
  const MwjfP P = mwjf_level(g.pressz[kp1]);
  const double rhok = mwjf_rho<false>(P, TM[o + (long long)(k - 1) * g.n2], SM[o + (long long)(k - 1) * g.n2], nullptr, nullptr);
  const double rhokp = mwjf_rho<false>(P, TM[o + (long long)(kp1 - 1) * g.n2], SM[o + (long long)(kp1 - 1) * g.n2], nullptr, nullptr);
  const double vvconv = (sp.convect_visc != 0.0) ? sp.convect_visc : sp.const_vvc;
  double vdc = sp.const_vdc, vvc = sp.const_vvc;
  if (rhok > rhokp && k < g.KMT[(long long)b * g.n2 + p2]) { vdc = sp.convect_diff; vvc = vvconv; }
  VDC[((long long)b * (g.km + 2) + k) * g.n2 + p2] = vdc;
  VVC[o + (long long)(k - 1) * g.n2] = vvc;
}

// ------------------------------------------------------------------------------------------
// Tracer right-hand side: the k loop of baroclinic_driver's first block loop
// (baroclinic.F90:700-870) = tracer_update (:1981-2300) with
//   hdifft_del2 (hmix_del2.F90:1030-1095), comp_flux_vel (advection.F90:2068-2127),
//   advt_centered (:2243-2301), vdifft (vertical_mix.F90:770-840), fresh-water and KPP
//   non-local sources, and the RHS/predictor store (:2212-2237).
// Algorithmic traffic per cell (SURVEY.md 8d phase B): 12 words with KPP, 9 without.
// ------------------------------------------------------------------------------------------
// third-order upwind weights (advection.F90:420-562): six per direction {alf+,bet+,gam+,alf-,bet-,del-}
struct Upw3Dev {
  const double *cx[6], *cy[6];     // (nx,ny,block)
  const double *cz[6];             // 1..km
};
// one face value of hupw3 (advection.F90:2560-2590 / :2617-2648): `flux` is the face flux weight whose
// sign selects the upwind side; kA/kB/kC are the KMT of the cells the stencil may reach
// (east|north, west|south, east-east|north-north); x(-1), x(0), x(+1), x(+2) along the direction
__device__ __forceinline__ double upw3_face(int k, double flux, int kA, int kB, int kC, const double (&w)[6],
                                            double xm1, double x0, double xp1, double xp2) {
  double work, ap, bp, gp, am, dm;
  if (k <= kA) { work = w[1]; ap = w[0]; } else { work = w[1] + w[0]; ap = 0.0; }
  if (k <= kB) { bp = work; gp = w[2]; } else { bp = work + w[2]; gp = 0.0; }
  if (k <= kC) { am = w[3]; dm = w[5]; } else { am = w[3] + w[5]; dm = 0.0; }
  const double bm = w[4];
  return (flux > 0.0) ? ap * xp1 + bp * x0 + gp * xm1 : am * xp1 + bm * x0 + dm * xp2;
}

struct TracerRhsArgs;
// the source of level k for a column with surface flux qsw = max(SHF_QSW, 0); trans_km1 carries the transmission to the top
// of the level from level to level (chlorophyll: TRANSKM1)
// dzt_pbc > 0: partial bottom cells -- the ground term divides by the cell's own thickness (sw_absorption.F90:880-889, 913-921)
__device__ __forceinline__ double sw_source(const TracerRhsArgs &a, double qsw, int k, int kmt, double dzrk, int chli, double &trans_km1, double dzt_pbc = 0.0);
struct TracerRhsArgs {
  const double *TCUR[2], *TOLD[2], *TMIX[2];
  double *TNEW[2];
  const double *UCUR, *VCUR, *VDC[2], *KPP_SRC[2], *STF[2], *TFW[2];
  const double *HDT[2];     // del4 only: precomputed biharmonic term per tracer (else null)
  const double *DH, *PCUR, *POLD;
  double c2dtt;
  int use_kpp_src;
  const int *KBL = nullptr;   // LDS kernel: KPP_SRC is read down to this level only (it is +-0 below)
  // LDS kernel, del4: the first Laplacian of the CURRENT tracers (k_del4_d2t's formula, from the tile already in LDS) for the next
  // step, whose mix-time field they are; nullptr: not formed
  double *D2N[2] = {nullptr, nullptr};
  const double *AHF = nullptr;
  // add_sw_absorb (sw_absorption.F90:818-947; tracer_update, baroclinic.F90:2176): penetrating short wave as a source of
  // potential temperature.  sw_on 0: off; type 0 / 1: the per-level table swabs(0:km); 2: the chlorophyll transmission table
  int sw_on, sw_type, sw_ksol;
  const double *QSW, *swabs, *swTr;
  const int *swCHLI;
  Upw3Dev up;               // UPW3 only
  const double *LTK[2];     // tadvect = 3: L(T) formed beforehand by the lw_lim kernels (kernels_lwlim.hpp); else null
  // forward elimination of the implicit vertical mixing fused into the right-hand side (k_tracer_rhs_lds<R, true>): the
  // elimination coefficients and the reduced right-hand side of tracer n go to E[n], F[n] instead of the RHS to TNEW
  double *E[2], *F[2];
};
__device__ __forceinline__ double sw_source(const TracerRhsArgs &a, double qsw, int k, int kmt, double dzrk, int chli, double &trans_km1, double dzt_pbc) {
  double top, bot;
  if (a.sw_type == 2) {
    if (k == 1) trans_km1 = 1.0;
    top = trans_km1;
    bot = a.swTr[(long long)chli * (a.sw_ksol + 1) + 2 * k];
    trans_km1 = bot;
  } else { top = a.swabs[k - 1]; bot = a.swabs[k]; }
  if (dzt_pbc > 0.0) return (k < kmt) ? qsw * (top - bot) * dzrk : qsw * top / dzt_pbc;
  return (k < kmt) ? qsw * (top - bot) * dzrk : qsw * top * dzrk;
}

// UPW3: advt_upwind3 + hupw3 (advection.F90:2313-2481, 2488-2676) in place of advt_centered.  The east
// face value of the west neighbour and the north face value of the south neighbour are recomputed by
// this thread (same operands, same order as the neighbour's own evaluation), the flux through the
// top face (AUX) is carried in a register, and four tracer levels k-1..k+2 are kept in registers.
// PBC: partial bottom cells (the branches of advection.F90:2040-2062, 2110, 2223-2294, 2380-2385, 2461-2465; hmix_del2.F90:1034-1051;
// vertical_mix.F90:790-807; sw_absorption.F90:880-921) with DZT / DZU formed from DZBC / DZUB (pbc_dz)
template <bool DEL4, bool UPW3, bool PBC = false>
__global__ void __launch_bounds__(POP_COL_THREADS, POP_TRC_WAVES)
k_tracer_rhs(DevGrid g, StepParams sp, TracerRhsArgs a) {
  Col c;
  if (!col_setup(g, c, true)) return;
  const int nxb = g.nxb, km = g.km;
  const long long n2 = g.n2;
  const int kmt = g.KMT[c.q2], kmtn = g.KMTN[c.q2], kmts = g.KMTS[c.q2], kmte = g.KMTE[c.q2], kmtw = g.KMTW[c.q2];
  const double dtn = g.DTN[c.q2], dts = g.DTS[c.q2], dte = g.DTE[c.q2], dtw = g.DTW[c.q2];
  const double dyu00 = g.DYU[c.q2], dyu0m = g.DYU[c.q2 - nxb], dyum0 = g.DYU[c.q2 - 1], dyumm = g.DYU[c.q2 - 1 - nxb];
  const double dxu00 = g.DXU[c.q2], dxu0m = g.DXU[c.q2 - nxb], dxum0 = g.DXU[c.q2 - 1], dxumm = g.DXU[c.q2 - 1 - nxb];
  const double tarear = g.TAREA_R[c.q2];
  const double psfac = (a.PCUR[c.q2] - a.POLD[c.q2]);
  double wtk = a.DH[c.q2];
  // partial bottom cells: bottom level / thickness of the four U cells around the T cell and of the T cell and its neighbours
  int kmu00 = 0, kmu0m = 0, kmum0 = 0, kmumm = 0;
  double dzub00 = 0, dzub0m = 0, dzubm0 = 0, dzubmm = 0, dzbc = 0, dzbcn = 0, dzbcs = 0, dzbce = 0, dzbcw = 0;
  if (PBC) {
    kmu00 = g.KMU[c.q2]; kmu0m = g.KMU[c.q2 - nxb]; kmum0 = g.KMU[c.q2 - 1]; kmumm = g.KMU[c.q2 - 1 - nxb];
    dzub00 = g.DZUB[c.q2]; dzub0m = g.DZUB[c.q2 - nxb]; dzubm0 = g.DZUB[c.q2 - 1]; dzubmm = g.DZUB[c.q2 - 1 - nxb];
    dzbc = g.DZBC[c.q2]; dzbcn = g.DZBC[c.q2 + nxb]; dzbcs = g.DZBC[c.q2 - nxb]; dzbce = g.DZBC[c.q2 + 1]; dzbcw = g.DZBC[c.q2 - 1];
  }
  double vtf[2];
  double tc_km1[2] = {0.0, 0.0}, tc_k[2], tc_kp1[2], to_k[2], to_kp1[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) { tc_k[n] = a.TCUR[n][c.base3]; to_k[n] = a.TOLD[n][c.base3]; }
  const double sw_q = a.sw_on ? fmax(a.QSW[c.q2], 0.0) : 0.0;
  const int sw_chli = (a.sw_on && a.sw_type == 2) ? a.swCHLI[c.q2] : 0;
  double sw_tkm1 = 1.0;
  // upwind3: weights at this cell, its west neighbour (x) and its south neighbour (y)
  double wx0[6], wxw[6], wy0[6], wys[6], aux[2] = {0.0, 0.0};
  int kmtee = 0, kmtnn = 0, kEw = 0, kWw = 0, kEEw = 0, kNs = 0, kSs = 0, kNNs = 0;
  double tarear_w = 0.0, tarear_s = 0.0;
  if (UPW3) {
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      wx0[t] = a.up.cx[t][c.q2]; wxw[t] = a.up.cx[t][c.q2 - 1];
      wy0[t] = a.up.cy[t][c.q2]; wys[t] = a.up.cy[t][c.q2 - nxb];
    }
    kmtee = g.KMTEE[c.q2]; kmtnn = g.KMTNN[c.q2];
    kEw = g.KMTE[c.q2 - 1]; kWw = g.KMTW[c.q2 - 1]; kEEw = g.KMTEE[c.q2 - 1];
    kNs = g.KMTN[c.q2 - nxb]; kSs = g.KMTS[c.q2 - nxb]; kNNs = g.KMTNN[c.q2 - nxb];
    tarear_w = g.TAREA_R[c.q2 - 1]; tarear_s = g.TAREA_R[c.q2 - nxb];
  }
  const long long vdcbase = ((long long)c.b * (km + 2)) * n2 + c.p2;
  double *__restrict__ const TNp[2] = {a.TNEW[0], a.TNEW[1]};   // outputs alias no input
  for (int k = 1; k <= km; ++k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    const int kp1 = (k < km) ? k + 1 : km;
    const long long okp = c.base3 + (long long)(kp1 - 1) * n2;
    // face flux velocities
    const double u00 = a.UCUR[o], u0m = a.UCUR[o - nxb], um0 = a.UCUR[o - 1], umm = a.UCUR[o - 1 - nxb];
    const double v00 = a.VCUR[o], v0m = a.VCUR[o - nxb], vm0 = a.VCUR[o - 1], vmm = a.VCUR[o - 1 - nxb];
    double UTE = 0.5 * (u00 * dyu00 + u0m * dyu0m);
    double UTW = 0.5 * (um0 * dyum0 + umm * dyumm);
    double VTN = 0.5 * (v00 * dxu00 + vm0 * dxum0);
    double VTS = 0.5 * (v0m * dxu0m + vmm * dxumm);
    double dzt = 0.0, dzt_kp1 = 0.0;
    if (PBC) {
      const double z00 = pbc_dz(g, k, kmu00, dzub00), z0m = pbc_dz(g, k, kmu0m, dzub0m), zm0 = pbc_dz(g, k, kmum0, dzubm0), zmm = pbc_dz(g, k, kmumm, dzubmm);
      UTE = 0.5 * (u00 * dyu00 * z00 + u0m * dyu0m * z0m);
      UTW = 0.5 * (um0 * dyum0 * zm0 + umm * dyumm * zmm);
      VTN = 0.5 * (v00 * dxu00 * z00 + vm0 * dxum0 * zm0);
      VTS = 0.5 * (v0m * dxu0m * z0m + vmm * dxumm * zmm);
      dzt = pbc_dz(g, k, kmt, dzbc); dzt_kp1 = pbc_dz(g, kp1, kmt, dzbc);
    }
    const double hdiv = VTN - VTS + UTE - UTW;
    double wtkb = 0.0;
    if (k < km) { const double FC = hdiv * tarear; wtkb = (k < kmt) ? (PBC ? wtk + FC : wtk + g.dz[k] * FC) : 0.0; }
    double CN = dtn, CS = dts, CE = dte, CW = dtw;
    if (PBC) {
      CN = dtn * fmin(dzt, pbc_dz(g, k, kmtn, dzbcn)) / dzt; CS = dts * fmin(dzt, pbc_dz(g, k, kmts, dzbcs)) / dzt;
      CE = dte * fmin(dzt, pbc_dz(g, k, kmte, dzbce)) / dzt; CW = dtw * fmin(dzt, pbc_dz(g, k, kmtw, dzbcw)) / dzt;
    }
    if (!(k <= kmtn && k <= kmt)) CN = 0.0;
    if (!(k <= kmts && k <= kmt)) CS = 0.0;
    if (!(k <= kmte && k <= kmt)) CE = 0.0;
    if (!(k <= kmtw && k <= kmt)) CW = 0.0;
    const double CC = -(CN + CS + CE + CW);
    const double dz2rk = g.dz2r[k], dzrk = g.dzr[k], dzwrk = g.dzwr[k];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      tc_kp1[n] = a.TCUR[n][okp];
      to_kp1[n] = a.TOLD[n][okp];
      double FT;
      if (DEL4) FT = a.HDT[n][o];
      else {
        const double *TM = a.TMIX[n];
        FT = sp.ah * (CC * TM[o] + CN * TM[o + nxb] + CS * TM[o - nxb] + CE * TM[o + 1] + CW * TM[o - 1]);
      }
      const double *TC = a.TCUR[n];
      double L;
      if (UPW3) {
        const double w1 = PBC ? tarear / dzt : tarear;   // WORK1 = TAREA_R / DZT (advection.F90:2380-2385)
        const double FVN = VTN * w1, FVS = -VTS * w1, FUE = UTE * w1, FUW = -UTW * w1;
        const double xw2 = TC[o - 2], xw1 = TC[o - 1], xe1 = TC[o + 1], xe2 = TC[o + 2];
        const double xs2 = TC[o - 2 * (long long)nxb], xs1 = TC[o - nxb], xn1 = TC[o + nxb], xn2 = TC[o + 2 * (long long)nxb];
        const double te = upw3_face(k, FUE, kmte, kmtw, kmtee, wx0, xw1, tc_k[n], xe1, xe2);
        const double tew = upw3_face(k, PBC ? UTW * (tarear_w / pbc_dz(g, k, kmtw, dzbcw)) : UTW * tarear_w, kEw, kWw, kEEw, wxw, xw2, xw1, tc_k[n], xe1);
        L = FUE * te + FUW * tew;
        const double tn = upw3_face(k, FVN, kmtn, kmts, kmtnn, wy0, xs1, tc_k[n], xn1, xn2);
        const double tns = upw3_face(k, PBC ? VTS * (tarear_s / pbc_dz(g, k, kmts, dzbcs)) : VTS * tarear_s, kNs, kSs, kNNs, wys, xs2, xs1, tc_k[n], xn1);
        L = L + FVN * tn + FVS * tns;
        // vertical: flux through the bottom face (:2397-2432)
        double azminus, dzminus;
        if (k < kmt - 1) { azminus = a.up.cz[3][k]; dzminus = a.up.cz[5][k]; } else { azminus = a.up.cz[3][k] + a.up.cz[5][k]; dzminus = 0.0; }
        double auxb = 0.0;
        if (k <= km - 1) {
          double tplus = a.up.cz[0][k] * tc_kp1[n] + a.up.cz[1][k] * tc_k[n];
          if (k > 1) tplus = tplus + a.up.cz[2][k] * tc_km1[n];
          double tminus = azminus * tc_kp1[n] + a.up.cz[4][k] * tc_k[n];
          if (k < km - 1) tminus = tminus + dzminus * TC[o + 2 * n2];
          auxb = (wtkb - fabs(wtkb)) * tplus + (wtkb + fabs(wtkb)) * tminus;
        }
        if (k == 1) L = L - dz2rk * auxb;
        else if (PBC) L = L + 0.5 / dzt * (aux[n] - auxb);
        else L = L + dz2rk * (aux[n] - auxb);
        aux[n] = auxb;
      } else if (a.LTK[0]) {
        L = a.LTK[n][o];                                   // lw_lim advection (advection.F90:2684-3280), formed beforehand
      } else {
        L = 0.5 * (hdiv * tc_k[n] + VTN * TC[o + nxb] - VTS * TC[o - nxb] + UTE * TC[o + 1] - UTW * TC[o - 1]) * tarear;
        if (PBC) {
          L = L / dzt;
          if (k != 1) L = L + 0.5 / dzt * wtk * (tc_km1[n] + tc_k[n]);
          if (k < km) L = L - 0.5 / dzt * wtkb * (tc_k[n] + tc_kp1[n]);
        } else {
        if (k != 1) L = L + dz2rk * wtk * (tc_km1[n] + tc_k[n]);
        if (k < km) L = L - dz2rk * wtkb * (tc_k[n] + tc_kp1[n]);
        }
      }
      FT = FT - L;
      if (k == 1) vtf[n] = (kmt >= 1) ? a.STF[n][c.q2] : 0.0;
      double vtfb, vd;
      if (PBC) {
        vtfb = (kmt > k) ? a.VDC[n][vdcbase + (long long)k * n2] * (to_k[n] - to_kp1[n]) / (0.5 * (dzt + dzt_kp1)) : 0.0;
        vd = (k <= kmt) ? (vtf[n] - vtfb) / dzt : 0.0;
      } else {
        vtfb = (kmt > k) ? a.VDC[n][vdcbase + (long long)k * n2] * (to_k[n] - to_kp1[n]) * dzwrk : 0.0;
        vd = (k <= kmt) ? (vtf[n] - vtfb) * dzrk : 0.0;
      }
      vtf[n] = vtfb;
      FT = FT + vd;
      if (k == 1) FT = FT + g.dzr[1] * a.TFW[n][c.q2];
      double src = 0.0;
      if (a.use_kpp_src) src = src + a.KPP_SRC[n][o];
      if (a.sw_on && n == 0) src = src + sw_source(a, sw_q, k, kmt, dzrk, sw_chli, sw_tkm1, PBC ? dzt : 0.0);
      FT = FT + src;
      if (k == 1 && sp.pavg) {
        if (kmt > 0) TNp[n][o] = a.c2dtt * FT - 2.0 * tc_k[n] * psfac / (sp.grav * g.dz[1]);
      } else {
        TNp[n][o] = (k <= kmt) ? a.c2dtt * FT : 0.0;
      }
      tc_km1[n] = tc_k[n]; tc_k[n] = tc_kp1[n]; to_k[n] = to_kp1[n];
    }
    wtk = wtkb;
  }
}

// ------------------------------------------------------------------------------------------
// Implicit vertical mixing of tracers: impvmixt (vertical_mix.F90:1263-1368) and
// impvmixt_correct (:1563-1658).  Thomas algorithm, one thread per column; E is kept in a
// scratch field, F in a scratch field, both only touched by the owning thread.
//   MODE 0: predictor/standard  TNEW = TOLD + F,  rhs = TNEW(k)
//   MODE 1: corrector           TNEW = TNEW + F,  rhs = RHS1 at k=1 only
// PRE (MODE 0 only): the k=1 update of baroclinic_correct_adjust for the no-pressure-averaging
//   branch (baroclinic.F90:1330-1338) applied to TNEW(1) before the solve.
// POST: freeze clamp (baroclinic.F90:1418-1421) + state -> RHO(new) (:1468-1475)
// ------------------------------------------------------------------------------------------
struct ImpvmixtArgs {
  double *TNEW[2];
  const double *TOLD[2], *TCUR[2], *VDC[2];
  const double *PSFC;                 // surface pressure on the LHS
  const double *POLD, *PCUR, *PNEW, *PMIX;
  double *E, *F, *RHO;
  double c2dtt;
  int nfirst, nlast;                  // 1-based tracer range
};

// PBC: partial bottom cells (vertical_mix.F90:1279-1287, 1577-1585): from level 2 on the column's own thicknesses
template <int MODE, bool PRE, bool POST, bool PBC = false>
__global__ void __launch_bounds__(POP_COL_THREADS)
k_impvmixt(DevGrid g, StepParams sp, ImpvmixtArgs a) {
  Col c;
  if (!col_setup(g, c, true)) return;
  const int km = g.km;
  const long long n2 = g.n2;
  const int kmt = g.KMT[c.q2];
  const double dzbc = PBC ? g.DZBC[c.q2] : 0.0;
  const double hfac1 = g.dz[1] / a.c2dtt;
  const double H1 = hfac1 + a.PSFC[c.q2] / (sp.grav * a.c2dtt);
  const long long vdcbase = ((long long)c.b * (km + 2)) * n2 + c.p2;
  double *__restrict__ const E = a.E;
  double *__restrict__ const F = a.F;
  for (int n = a.nfirst - 1; n <= a.nlast - 1; ++n) {
    double *__restrict__ const TN = a.TNEW[n];
    const double *__restrict__ const VDC = a.VDC[n];
    const double *__restrict__ const TO = a.TOLD[n];
    double rhs1 = 0.0;
    if (MODE == 1) {
      if (kmt > 0)
        rhs1 = ((2.0 * a.TCUR[n][c.base3] - a.TOLD[n][c.base3]) * (a.PCUR[c.q2] - a.POLD[c.q2]) -
                TN[c.base3] * (a.PNEW[c.q2] - a.PCUR[c.q2])) / (sp.grav * g.dz[1]);
    }
    double t1 = TN[c.base3];
    if (PRE) {
      if (kmt > 0) t1 = t1 - a.TOLD[n][c.base3] * (a.PNEW[c.q2] - a.PMIX[c.q2]) / (sp.grav * g.dz[1]);
    }
    double A = g.afac_t[1] * VDC[vdcbase + n2];
    double D = H1 + A;
    double Ek = A / D;
    double B = H1 * Ek;
    double Fk = (MODE == 1) ? hfac1 * rhs1 / D : hfac1 * t1 / D;
    E[c.base3] = Ek;
    F[c.base3] = Fk;
#pragma unroll 4
    for (int k = 2; k <= km; ++k) {
      const long long o = c.base3 + (long long)(k - 1) * n2;
      const double C = A;
      double hf = g.dz[k] / a.c2dtt;
      A = g.afac_t[k] * VDC[vdcbase + (long long)k * n2];
      if (PBC) {
        const double dzt = pbc_dz(g, k, kmt, dzbc);
        A = sp.aidif * VDC[vdcbase + (long long)k * n2] / (0.5 * (dzt + pbc_dz(g, k + 1, kmt, dzbc)));
        hf = dzt / a.c2dtt;
      }
      const double tnk = (MODE == 1) ? 0.0 : TN[o];
      if (k > kmt) { Fk = 0.0; }
      else {
        D = (k == kmt) ? hf + B : hf + A + B;
        Ek = A / D;
        B = (hf + B) * Ek;
        Fk = (MODE == 1) ? C * Fk / D : (hf * tnk + C * Fk) / D;
        E[o] = Ek;
      }
      F[o] = Fk;
    }
    // back substitution and update, bottom to top
    double Fkp1 = 0.0;
#pragma unroll 4
    for (int k = km; k >= 1; --k) {
      const long long o = c.base3 + (long long)(k - 1) * n2;
      double f = F[o];
      if (k < km && k < kmt) f = f + E[o] * Fkp1;
      Fkp1 = f;
      double tn = (MODE == 1) ? TN[o] + f : TO[o] + f;
      if (POST && n == 0 && k == 1 && sp.reset_to_freezing) tn = fmax(tn, -2.0);
      TN[o] = tn;
    }
  }
  if (POST) {
    for (int k = 1; k <= km; ++k) {
      const long long o = c.base3 + (long long)(k - 1) * n2;
      const MwjfP P = mwjf_level(g.pressz[k]);
      a.RHO[o] = mwjf_rho<false>(P, a.TNEW[0][o], a.TNEW[1][o], nullptr, nullptr);
    }
  }
}

// back substitution of impvmixt (predictor) from elimination coefficients a fused right-hand-side kernel left in
// E[n], F[n] (k_tracer_rhs_lds<R, true>): T(new) = T(old) + x.  Same operations as the second half of k_impvmixt<0>.
struct ImpvmixtBackArgs { double *TNEW[2]; const double *TOLD[2], *E[2], *F[2]; };
__global__ void __launch_bounds__(POP_COL_THREADS)
k_impvmixt_back(DevGrid g, ImpvmixtBackArgs a) {
  Col c;
  if (!col_setup(g, c, true)) return;
  const int km = g.km, n = blockIdx.z;
  const long long n2 = g.n2;
  const int kmt = g.KMT[c.q2];
  double *__restrict__ const TN = a.TNEW[n];
  const double *__restrict__ const TO = a.TOLD[n];
  const double *__restrict__ const E = a.E[n];
  const double *__restrict__ const F = a.F[n];
  double Fkp1 = 0.0;
#pragma unroll 4
  for (int k = km; k >= 1; --k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    double f = F[o];
    if (k < km && k < kmt) f = f + E[o] * Fkp1;
    Fkp1 = f;
    TN[o] = TO[o] + f;
  }
}

// ------------------------------------------------------------------------------------------
// Momentum right-hand side: the k loop of baroclinic_driver's second block loop
// (baroclinic.F90:966-1048) = clinic (:1724-1890) with advu (advection.F90:1307-1491),
// Coriolis, gradp (pressure_grad.F90:258-301) + grad (operators.F90:178-187),
// hdiffu_del2 (hmix_del2.F90:892-927), vdiffu (vertical_mix.F90:935-1010), the implicit
// Coriolis solve and the vertical integrals ZX, ZY (:1013-1057).
// Algorithmic traffic per cell (SURVEY.md 8d phase E): 10 words.
// ------------------------------------------------------------------------------------------
struct MomentumRhsArgs {
  const double *UCUR, *VCUR, *UOLD, *VOLD, *UMIX, *VMIX;
  const double *RHOOLD, *RHOCUR, *RHONEW, *VVC, *DHU;
  const double *HDU, *HDV;   // del4 only: precomputed biharmonic friction
  double *UNEW, *VNEW, *ZX, *ZY;
  // LDS kernel, del4: the first Laplacian of the CURRENT velocity (k_del4_d2u's formula, from the tile in LDS) for the next step
  double *D2N[2] = {nullptr, nullptr};
  const double *AMF = nullptr;
};

// PBC: partial bottom cells (advection.F90:1245-1300, 1352, 1381-1467; hmix_del2.F90:852-886; vertical_mix.F90:946-995;
// baroclinic.F90:1037-1039) with DZU formed from KMU / DZUB (pbc_dz)
template <bool DEL4, bool PBC = false>
__global__ void __launch_bounds__(POP_COL_THREADS, POP_MOM_WAVES)
k_momentum_rhs(DevGrid g, StepParams sp, MomentumRhsArgs a) {
  Col c;
  if (!col_setup(g, c, true)) return;
  const int nxb = g.nxb, km = g.km;
  const long long n2 = g.n2;
  const int kmu = g.KMU[c.q2];
  double dyu[3][3], dxu[3][3];
  int kmu9[3][3] = {}; double dzub9[3][3] = {};   // PBC: bottom level / thickness of the nine U cells
#pragma unroll
  for (int dj = -1; dj <= 1; ++dj)
#pragma unroll
    for (int di = -1; di <= 1; ++di) {
      dyu[dj + 1][di + 1] = g.DYU[c.q2 + dj * nxb + di];
      dxu[dj + 1][di + 1] = g.DXU[c.q2 + dj * nxb + di];
      if (PBC) { kmu9[dj + 1][di + 1] = g.KMU[c.q2 + dj * nxb + di]; dzub9[dj + 1][di + 1] = g.DZUB[c.q2 + dj * nxb + di]; }
    }
  const double uar = g.UAREA_R[c.q2], fcor = g.FCOR[c.q2], kxu = g.KXU[c.q2], kyu = g.KYU[c.q2];
  const double dxur = g.DXUR[c.q2], dyur = g.DYUR[c.q2], hur = g.HUR[c.q2];
  const double cc_h = g.DUC[c.q2] + g.DUM[c.q2];
  const double dun = g.DUN[c.q2], dus = g.DUS[c.q2], due = g.DUE[c.q2], duw = g.DUW[c.q2];
  const double dmc = g.DMC[c.q2], dmn = g.DMN[c.q2], dms = g.DMS[c.q2], dme = g.DME[c.q2], dmw = g.DMW[c.q2];
  const double smfx = (kmu >= 1) ? g.SMF1[c.q2] : 0.0, smfy = (kmu >= 1) ? g.SMF2[c.q2] : 0.0;
  double wuk = a.DHU[c.q2];
  double vuf = smfx, vvf = smfy;
  double rhokmx = 0.0, rhokmy = 0.0, sumx = 0.0, sumy = 0.0, zx = 0.0, zy = 0.0;
  double uc_km1 = 0.0, vc_km1 = 0.0, uc_k = a.UCUR[c.base3], vc_k = a.VCUR[c.base3];
  double uo_k = a.UOLD[c.base3], vo_k = a.VOLD[c.base3];
  double *__restrict__ const UNp = a.UNEW;   // outputs alias no input
  double *__restrict__ const VNp = a.VNEW;
  for (int k = 1; k <= km; ++k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    const int kp1 = (k < km) ? k + 1 : km;
    const long long okp = c.base3 + (long long)(kp1 - 1) * n2;
    const double uc_kp1 = a.UCUR[okp], vc_kp1 = a.VCUR[okp], uo_kp1 = a.UOLD[okp], vo_kp1 = a.VOLD[okp];
    double u[3][3], v[3][3], ud[3][3], vd[3][3];
#pragma unroll
    for (int dj = -1; dj <= 1; ++dj)
#pragma unroll
      for (int di = -1; di <= 1; ++di) {
        const double uu = (di == 0 && dj == 0) ? uc_k : a.UCUR[o + dj * nxb + di];
        const double vv = (di == 0 && dj == 0) ? vc_k : a.VCUR[o + dj * nxb + di];
        u[dj + 1][di + 1] = uu; v[dj + 1][di + 1] = vv;
        ud[dj + 1][di + 1] = uu * dyu[dj + 1][di + 1];
        vd[dj + 1][di + 1] = vv * dxu[dj + 1][di + 1];
        if (PBC) {
          const double z = pbc_dz(g, k, kmu9[dj + 1][di + 1], dzub9[dj + 1][di + 1]);
          ud[dj + 1][di + 1] = ud[dj + 1][di + 1] * z; vd[dj + 1][di + 1] = vd[dj + 1][di + 1] * z;
        }
      }
    // PBC: thickness of this U cell at level k and k + 1, of its four neighbours at level k
    const double dzu = PBC ? pbc_dz(g, k, kmu, dzub9[1][1]) : 0.0, dzu_kp1 = PBC ? pbc_dz(g, kp1, kmu, dzub9[1][1]) : 0.0;
#define UD(di, dj) ud[(dj) + 1][(di) + 1]
#define VD(di, dj) vd[(dj) + 1][(di) + 1]
#define UU(di, dj) u[(dj) + 1][(di) + 1]
#define VV(di, dj) v[(dj) + 1][(di) + 1]
    const double UUW = 0.25 * (UD(0, 0) + UD(-1, 0)) + 0.125 * (UD(0, -1) + UD(-1, -1) + UD(0, 1) + UD(-1, 1));
    const double UUE = 0.25 * (UD(1, 0) + UD(0, 0)) + 0.125 * (UD(1, -1) + UD(0, -1) + UD(1, 1) + UD(0, 1));   // = UUW(i+1,j)
    const double VUS = 0.25 * (VD(0, 0) + VD(0, -1)) + 0.125 * (VD(-1, 0) + VD(-1, -1) + VD(1, 0) + VD(1, -1));
    const double VUN = 0.25 * (VD(0, 1) + VD(0, 0)) + 0.125 * (VD(-1, 1) + VD(-1, 0) + VD(1, 1) + VD(1, 0));   // = VUS(i,j+1)
    const double wukb = PBC ? wuk + (VUN - VUS + UUE - UUW) * uar : wuk + g.c2dz[k] * 0.5 * (VUN - VUS + UUE - UUW) * uar;
    const double cc = VUN - VUS + UUE - UUW;
    double LU = 0.5 * (cc * UU(0, 0) + VUN * UU(0, 1) - VUS * UU(0, -1) + UUE * UU(1, 0) - UUW * UU(-1, 0)) * uar;
    double LV = 0.5 * (cc * VV(0, 0) + VUN * VV(0, 1) - VUS * VV(0, -1) + UUE * VV(1, 0) - UUW * VV(-1, 0)) * uar;
    if (PBC) { LU = LU / dzu; LV = LV / dzu; }
    if (k == 1) { LU = LU + g.dzr[k] * wuk * uc_k; LV = LV + g.dzr[k] * wuk * vc_k; }
    else if (PBC) { LU = LU + 0.5 / dzu * wuk * (uc_km1 + uc_k); LV = LV + 0.5 / dzu * wuk * (vc_km1 + vc_k); }
    else { LU = LU + g.dz2r[k] * wuk * (uc_km1 + uc_k); LV = LV + g.dz2r[k] * wuk * (vc_km1 + vc_k); }
    if (k < km) {
      if (PBC) { LU = LU - 0.5 / dzu * wukb * (uc_k + uc_kp1); LV = LV - 0.5 / dzu * wukb * (vc_k + vc_kp1); }
      else { LU = LU - g.dz2r[k] * wukb * (uc_k + uc_kp1); LV = LV - g.dz2r[k] * wukb * (vc_k + vc_kp1); }
    }
    if (k <= kmu) {
      LU = LU + uc_k * vc_k * kyu - vc_k * vc_k * kxu;
      LV = LV + uc_k * vc_k * kxu - uc_k * uc_k * kyu;
    } else { LU = 0.0; LV = 0.0; }
    double FX = -LU, FY = -LV;
    // Coriolis (baroclinic.F90:1764-1781)
    if (sp.impcor && sp.leapfrogts) {
      FX = FX + fcor * (sp.gamma * vc_k + (1.0 - sp.gamma) * vo_k);
      FY = FY - fcor * (sp.gamma * uc_k + (1.0 - sp.gamma) * uo_k);
    } else if (!sp.impcor && sp.leapfrogts) {
      FX = FX + fcor * vc_k; FY = FY - fcor * uc_k;
    } else {
      FX = FX + fcor * vo_k; FY = FY - fcor * uo_k;
    }
    // hydrostatic pressure gradient
    {
      const double bk = g.bouss[k];
      double f00, f10, f01, f11;
      if (sp.pavg) {
        f00 = 0.25 * (a.RHONEW[o] + 2.0 * a.RHOCUR[o] + a.RHOOLD[o]) * bk;
        f10 = 0.25 * (a.RHONEW[o + 1] + 2.0 * a.RHOCUR[o + 1] + a.RHOOLD[o + 1]) * bk;
        f01 = 0.25 * (a.RHONEW[o + nxb] + 2.0 * a.RHOCUR[o + nxb] + a.RHOOLD[o + nxb]) * bk;
        f11 = 0.25 * (a.RHONEW[o + nxb + 1] + 2.0 * a.RHOCUR[o + nxb + 1] + a.RHOOLD[o + nxb + 1]) * bk;
      } else {
        f00 = a.RHOCUR[o] * bk; f10 = a.RHOCUR[o + 1] * bk; f01 = a.RHOCUR[o + nxb] * bk; f11 = a.RHOCUR[o + nxb + 1] * bk;
      }
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
    // horizontal friction
    {
      double hdu, hdv;
      if (DEL4) { hdu = a.HDU[o]; hdv = a.HDV[o]; }
      else {
        const double *UM = a.UMIX, *VM = a.VMIX;
        const double um0 = UM[o], umn = UM[o + nxb], ums = UM[o - nxb], ume = UM[o + 1], umw = UM[o - 1];
        const double vm0 = VM[o], vmn = VM[o + nxb], vms = VM[o - nxb], vme = VM[o + 1], vmw = VM[o - 1];
        double cn = dun, cs = dus, ce = due, cw = duw;
        if (PBC) {   // the four neighbour weights scaled by min(DZU) / DZU; the central one unchanged
          cn = dun * fmin(pbc_dz(g, k, kmu9[2][1], dzub9[2][1]), dzu) / dzu; cs = dus * fmin(pbc_dz(g, k, kmu9[0][1], dzub9[0][1]), dzu) / dzu;
          ce = due * fmin(pbc_dz(g, k, kmu9[1][2], dzub9[1][2]), dzu) / dzu; cw = duw * fmin(pbc_dz(g, k, kmu9[1][0], dzub9[1][0]), dzu) / dzu;
        }
        hdu = sp.am * ((cc_h * um0 + cn * umn + cs * ums + ce * ume + cw * umw) +
                       (dmc * vm0 + dmn * vmn + dms * vms + dme * vme + dmw * vmw));
        hdv = sp.am * ((cc_h * vm0 + cn * vmn + cs * vms + ce * vme + cw * vmw) -
                       (dmc * um0 + dmn * umn + dms * ums + dme * ume + dmw * umw));
        if (k > kmu) { hdu = 0.0; hdv = 0.0; }
      }
      FX = FX + hdu; FY = FY + hdv;
    }
    // vertical friction (explicit part) with quadratic bottom drag
    {
      const double vvc = a.VVC[o];
      double vufb = vvc * (uo_k - uo_kp1) * g.dzwr[k];
      double vvfb = vvc * (vo_k - vo_kp1) * g.dzwr[k];
      if (PBC) {
        const double W = (k < km) ? 0.5 * (dzu + dzu_kp1) : 0.5 * dzu_kp1;
        vufb = vvc * (uo_k - uo_kp1) / W; vvfb = vvc * (vo_k - vo_kp1) / W;
      }
      if (k == kmu) {
        const double vmag = sp.bottom_drag * sqrt(uo_k * uo_k + vo_k * vo_k);
        vufb = vmag * uo_k; vvfb = vmag * vo_k;
      }
      double vdu = (k <= kmu) ? (vuf - vufb) * g.dzr[k] : 0.0;
      double vdv = (k <= kmu) ? (vvf - vvfb) * g.dzr[k] : 0.0;
      if (PBC) { vdu = (k <= kmu) ? (vuf - vufb) / dzu : 0.0; vdv = (k <= kmu) ? (vvf - vvfb) / dzu : 0.0; }
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
    if (PBC) { zx = zx + FX * dzu; zy = zy + FY * dzu; }
    else { zx = zx + FX * g.dz[k]; zy = zy + FY * g.dz[k]; }
    wuk = wukb;
    uc_km1 = uc_k; vc_km1 = vc_k; uc_k = uc_kp1; vc_k = vc_kp1; uo_k = uo_kp1; vo_k = vo_kp1;
#undef UD
#undef VD
#undef UU
#undef VV
  }
  a.ZX[c.q2] = zx * hur;
  a.ZY[c.q2] = zy * hur;
}

// ------------------------------------------------------------------------------------------
// impvmixu (vertical_mix.F90:1762-1868) + add old velocity + removal of the vertical mean
// and land zeroing (baroclinic.F90:1077-1129).  One thread per column.
// ------------------------------------------------------------------------------------------
struct ImpvmixuArgs {
  double *UNEW, *VNEW, *E;
  const double *UOLD, *VOLD, *VVC;
  const double *UB = nullptr, *VB = nullptr;   // register kernel with ADD: the new barotropic velocity, added in the final store
};
// PBC: partial bottom cells (vertical_mix.F90:1777-1785; baroclinic.F90:1097-1106)
template <bool PBC = false>
__global__ void __launch_bounds__(POP_COL_THREADS)
k_impvmixu_norm(DevGrid g, StepParams sp, ImpvmixuArgs a) {
  Col c;
  if (!col_setup(g, c, true)) return;
  const int km = g.km;
  const long long n2 = g.n2;
  const int kmu = g.KMU[c.q2];
  const double hur = g.HUR[c.q2];
  const double dzub = PBC ? g.DZUB[c.q2] : 0.0;
  const double hf1 = g.dz[1] / sp.c2dtu;
  double *__restrict__ const UN = a.UNEW;
  double *__restrict__ const VN = a.VNEW;
  double *__restrict__ const E = a.E;
  const double *__restrict__ const VVC = a.VVC;
  const double *__restrict__ const UO = a.UOLD;
  const double *__restrict__ const VO = a.VOLD;
  double A = g.afac_u[1] * VVC[c.base3];
  double D = hf1 + A;
  double Ek = A / D;
  double B = hf1 * Ek;
  double F1 = hf1 * UN[c.base3] / D, F2 = hf1 * VN[c.base3] / D;
  E[c.base3] = Ek; UN[c.base3] = F1; VN[c.base3] = F2;
#pragma unroll 4
  for (int k = 2; k <= km; ++k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    const double C = A;
    double hf = g.dz[k] / sp.c2dtu;
    A = g.afac_u[k] * VVC[o];
    if (PBC) {
      const double dzu = pbc_dz(g, k, kmu, dzub);
      hf = dzu / sp.c2dtu;
      A = sp.aidif * VVC[o] / (0.5 * (dzu + pbc_dz(g, k + 1, kmu, dzub)));
    }
    const double un = UN[o], vn = VN[o];
    if (k <= kmu) {
      D = (k < kmu) ? hf + A + B : hf + B;
      Ek = A / D;
      B = (hf + B) * Ek;
      F1 = (hf * un + C * F1) / D;
      F2 = (hf * vn + C * F2) / D;
      E[o] = Ek;
    } else { F1 = 0.0; F2 = 0.0; }
    UN[o] = F1; VN[o] = F2;
  }
  double F1p = 0.0, F2p = 0.0;
#pragma unroll 4
  for (int k = km; k >= 1; --k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    double f1 = UN[o], f2 = VN[o];
    if (k < km && k < kmu) { const double e = E[o]; f1 = f1 + e * F1p; f2 = f2 + e * F2p; }
    F1p = f1; F2p = f2;
    UN[o] = UO[o] + f1;
    VN[o] = VO[o] + f2;
  }
  double w1 = 0.0, w2 = 0.0;
#pragma unroll 4
  for (int k = 1; k <= km; ++k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    const double dzk = PBC ? pbc_dz(g, k, kmu, dzub) : g.dz[k];
    w1 = w1 + UN[o] * dzk;
    w2 = w2 + VN[o] * dzk;
  }
  w1 = w1 * hur; w2 = w2 * hur;
#pragma unroll 4
  for (int k = 1; k <= km; ++k) {
    const long long o = c.base3 + (long long)(k - 1) * n2;
    if (k <= kmu) { UN[o] = UN[o] - w1; VN[o] = VN[o] - w2; }
    else { UN[o] = 0.0; VN[o] = 0.0; }
  }
}

// ------------------------------------------------------------------------------------------
// step tail (step_mod.F90:572-592): add the barotropic velocity where k <= KMU.  3-D parallel.
// ------------------------------------------------------------------------------------------
__global__ void k_add_barotropic(DevGrid g, double *__restrict__ UNEW, double *__restrict__ VNEW,
                                 const double *__restrict__ UB, const double *__restrict__ VB) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y + 1, b = blockIdx.z;
  if (p2 >= g.n2) return;
  if (land_run(g, b, (long long)blockIdx.x * blockDim.x, blockDim.x)) return;
  const long long q2 = (long long)b * g.n2 + p2;
  if (k <= g.KMU[q2]) {
    const long long o = (long long)b * g.n3 + (long long)(k - 1) * g.n2 + p2;
    UNEW[o] = UNEW[o] + UB[q2];
    VNEW[o] = VNEW[o] + VB[q2];
  }
}

// PGUESS (step_mod.F90:634-640); on averaging steps the :793-794 correction is applied by k_avg2d
__global__ void k_pguess(long long n, double *__restrict__ PG, const double *__restrict__ PN, const double *__restrict__ PC,
                         const double *__restrict__ PO) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) PG[p] = 3.0 * (PN[p] - PC[p]) + PO[p];
}

// ------------------------------------------------------------------------------------------
// averaging step (step_mod.F90:663-796)
// ------------------------------------------------------------------------------------------
struct Avg2dArgs {
  double *UBO, *UBC, *VBO, *VBC, *GXO, *GXC, *GYO, *GYC, *PO, *PC, *PG, *FW_OLD;
  const double *UBN, *VBN, *GXN, *GYN, *PN, *FW;
  double *T1O[2], *T1C[2];         // level-1 slices of TRACER(old), TRACER(cur) base pointers (3-D arrays)
  const double *T1N[2];
  double dz1, grav;
};
__global__ void k_avg2d(DevGrid g, Avg2dArgs a) {
  Col c;
  if (!col_setup(g, c, false)) return;
  const long long q = c.q2;
  a.UBO[q] = 0.5 * (a.UBO[q] + a.UBC[q]); a.VBO[q] = 0.5 * (a.VBO[q] + a.VBC[q]);
  a.UBC[q] = 0.5 * (a.UBC[q] + a.UBN[q]); a.VBC[q] = 0.5 * (a.VBC[q] + a.VBN[q]);
  a.GXO[q] = 0.5 * (a.GXO[q] + a.GXC[q]); a.GYO[q] = 0.5 * (a.GYO[q] + a.GYC[q]);
  a.GXC[q] = 0.5 * (a.GXC[q] + a.GXN[q]); a.GYC[q] = 0.5 * (a.GYC[q] + a.GYN[q]);
  a.FW_OLD[q] = 0.5 * (a.FW[q] + a.FW_OLD[q]);
  const double po = a.PO[q], pc = a.PC[q], pn = a.PN[q];
  const double pfo = 0.5 * (po + pc), pfc = 0.5 * (pc + pn);
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const double to = a.T1O[n][c.base3], tc = a.T1C[n][c.base3], tn = a.T1N[n][c.base3];
    double t = 0.5 * ((a.dz1 + po / a.grav) * to + (a.dz1 + pc / a.grav) * tc);
    t = t / (a.dz1 + pfo / a.grav);
    const double mn = fmin(to, tc), mx = fmax(to, tc);
    if (t < mn) t = mn;
    if (t > mx) t = mx;
    a.T1O[n][c.base3] = t;
    double t2 = 0.5 * ((a.dz1 + pc / a.grav) * tc + (a.dz1 + pn / a.grav) * tn);
    t2 = t2 / (a.dz1 + pfc / a.grav);
    const double mn2 = fmin(tc, tn), mx2 = fmax(tc, tn);
    if (t2 < mn2) t2 = mn2;
    if (t2 > mx2) t2 = mx2;
    a.T1C[n][c.base3] = t2;
  }
  a.PO[q] = pfo; a.PC[q] = pfc;
  a.PG[q] = 0.5 * (a.PG[q] + pn);
}
struct Avg3dArgs {
  double *UO, *UC, *VO, *VC, *TO[2], *TC[2], *RO, *RC;
  const double *UN, *VN, *TN[2];
};
// 3-D fields: U,V all levels; tracers k >= 2 here, k == 1 in k_avg2d (run first); then RHO(old,cur)
__global__ void k_avg3d(DevGrid g, Avg3dArgs a) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y + 1, b = blockIdx.z;
  if (p2 >= g.n2) return;
  if (land_run(g, b, (long long)blockIdx.x * blockDim.x, blockDim.x)) return;
  const long long o = (long long)b * g.n3 + (long long)(k - 1) * g.n2 + p2;
  const double uc = a.UC[o], vc = a.VC[o];
  a.UO[o] = 0.5 * (a.UO[o] + uc); a.VO[o] = 0.5 * (a.VO[o] + vc);
  a.UC[o] = 0.5 * (uc + a.UN[o]); a.VC[o] = 0.5 * (vc + a.VN[o]);
  double to[2], tc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    if (k >= 2) {
      const double c0 = a.TC[n][o];
      to[n] = 0.5 * (a.TO[n][o] + c0);
      tc[n] = 0.5 * (c0 + a.TN[n][o]);
      a.TO[n][o] = to[n]; a.TC[n][o] = tc[n];
    } else { to[n] = a.TO[n][o]; tc[n] = a.TC[n][o]; }
  }
  const MwjfP P = mwjf_level(g.pressz[k]);
  a.RO[o] = mwjf_rho<false>(P, to[0], to[1], nullptr, nullptr);
  a.RC[o] = mwjf_rho<false>(P, tc[0], tc[1], nullptr, nullptr);
}

}  // namespace pop
