// kernels_mix.hpp -- vertical-mixing coefficient kernels (Richardson; KPP in kernels_kpp.hpp),
// biharmonic horizontal mixing (kernels_del4.hpp) and the host-array state() entry point.
#pragma once
#include "kernels_common.hpp"

namespace pop {

struct MixDev {          // device-resident constants of the mixing schemes
  const double *bckgrnd_vdc = nullptr, *bckgrnd_vvc = nullptr, *zgrid = nullptr, *hwide = nullptr, *Ricr = nullptr;
  const double *D4AMF = nullptr, *D4AHF = nullptr;
  const double *d4DTN = nullptr, *d4DTS = nullptr, *d4DTE = nullptr, *d4DTW = nullptr;
  const double *d4DUC = nullptr, *d4DUN = nullptr, *d4DUS = nullptr, *d4DUE = nullptr, *d4DUW = nullptr;
  const double *d4DMC = nullptr, *d4DMN = nullptr, *d4DMS = nullptr, *d4DME = nullptr, *d4DMW = nullptr, *d4DUM = nullptr;
  void *kpp = nullptr;     // KppHost (kernels_kpp.hpp), owned by the context
};
struct MixState {        // per-step field pointers handed to the mixing kernels
  const double *TMIX[2], *UMIX, *VMIX, *UCUR, *VCUR, *RHOMIX, *STF[2], *SHF_QSW;
  double *VDC[2], *VVC, *KPP_SRC[2], *HBLT, *HMXL, *HMXL_DR;
  int *KBL = nullptr;        // KPP: level of the boundary-layer depth, paired with KPP_SRC (nullptr: the scheme's own array)
  int src_clear_all = 0;     // KPP: this set's KPP_SRC may hold non-zeros below its KBL (a caller wrote it): clear every level
  double *S3a, *S3b, *S3c, *S3d, *E3, *F3;   // 3-D scratch
};

// ---- vmix_coeffs_rich (vmix_rich.F90:224-400), convection by diffusion
// pass 1: Richardson number at T points (whole array) and tracer diffusivity; 3-D parallel.  PBC (:266-275, :308-312): the
// vertical shear is formed at the U points from the U-cell thicknesses and averaged to the T point
template <bool PBC>
__global__ void k_rich_t(DevGrid g, StepParams sp, MixState s, double *__restrict__ RICH) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y + 1, b = blockIdx.z;
  if (p2 >= g.n2) return;
  const int i = p2 % g.nxb, j = p2 / g.nxb, nxb = g.nxb;
  const int kp1 = min(k + 1, g.km);
  const long long q2 = (long long)b * g.n2 + p2, o = (long long)b * g.n3 + p2;
  const long long ok = o + (long long)(k - 1) * g.n2, okp = o + (long long)(kp1 - 1) * g.n2;
  auto u2t = [&](const double *A, long long q) {   // ugrid_to_tgrid, AT0=ATS=ATW=ATSW=p25 (grid.F90:2905-2908)
    return (i >= 1 && j >= 1) ? 0.25 * A[q] + 0.25 * A[q - nxb] + 0.25 * A[q - 1] + 0.25 * A[q - 1 - nxb] : 0.0;
  };
  double rich = 0.0, vdc = 0.0;
  const double critnu = sp.convect_diff;
  const int kmt = g.KMT[q2];
  if (k < kmt) {
    const MwjfP P = mwjf_level(g.pressz[kp1]);
    const double rhok = mwjf_rho<false>(P, s.TMIX[0][ok], s.TMIX[1][ok], nullptr, nullptr);
    if constexpr (PBC) {
      auto shear = [&](const double *A, long long d) {   // (A(k) - A(kp1)) / (p5*(DZU(k) + DZU(kp1))) at the U point q2 + d
        const int kmu = g.KMU[q2 + d];
        const double dzub = g.DZUB[q2 + d];
        return (A[ok + d] - A[okp + d]) / (0.5 * (pbc_dz(g, k, kmu, dzub) + pbc_dz(g, kp1, kmu, dzub)));
      };
      double ut = 0.0, vt = 0.0;
      if (i >= 1 && j >= 1) {
        ut = 0.25 * shear(s.UMIX, 0) + 0.25 * shear(s.UMIX, -nxb) + 0.25 * shear(s.UMIX, -1) + 0.25 * shear(s.UMIX, -1 - nxb);
        vt = 0.25 * shear(s.VMIX, 0) + 0.25 * shear(s.VMIX, -nxb) + 0.25 * shear(s.VMIX, -1) + 0.25 * shear(s.VMIX, -1 - nxb);
      }
      const double dzb = g.DZBC[q2];
      const double h = 0.5 * (pbc_dz(g, k, kmt, dzb) + pbc_dz(g, kp1, kmt, dzb));
      rich = -sp.grav * (rhok - s.RHOMIX[okp]) / h / (ut * ut + vt * vt + 1.0e-10 / (h * h));
    } else {
      const double du = u2t(s.UMIX, ok) - u2t(s.UMIX, okp), dv = u2t(s.VMIX, ok) - u2t(s.VMIX, okp);
      rich = -sp.grav * g.dzw[k] * (rhok - s.RHOMIX[okp]) / (du * du + dv * dv + 1.0e-10);
    }
    const double f = 1.0 + 5.0 * rich;
    vdc = fmin(critnu, sp.rich_bckgrnd_vdc + (sp.rich_bckgrnd_vvc + sp.rich_mix / (f * f)) / f);
  }
  if (rich < 0.0) vdc = critnu;
  RICH[ok] = rich;
  s.VDC[0][((long long)b * (g.km + 2) + k) * g.n2 + p2] = vdc;
}
// pass 2: Richardson number at U points (tgrid_to_ugrid, grid.F90:3399-3413) and viscosity
__global__ void k_rich_u(DevGrid g, StepParams sp, MixState s, const double *__restrict__ RICH) {
  const int p2 = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y + 1, b = blockIdx.z;
  if (p2 >= g.n2) return;
  const int i = p2 % g.nxb, j = p2 / g.nxb, nxb = g.nxb;
  const long long q2 = (long long)b * g.n2 + p2, ok = (long long)b * g.n3 + (long long)(k - 1) * g.n2 + p2;
  double richu = 0.0;
  if (i < g.nxb - 1 && j < g.nyb - 1)
    richu = g.AU0[q2] * RICH[ok] + g.AUN[q2] * RICH[ok + nxb] + g.AUE[q2] * RICH[ok + 1] + g.AUNE[q2] * RICH[ok + nxb + 1];
  const double critnu = sp.convect_visc;
  double vvc;
  if (k < g.KMU[q2]) { const double f = 1.0 + 5.0 * richu; vvc = fmin(critnu, sp.rich_bckgrnd_vvc + sp.rich_mix / (f * f)); }
  else { richu = 0.0; vvc = 0.0; }
  if (richu < 0.0) vvc = critnu;
  s.VVC[ok] = vvc;
}

// state() on host arrays staged through the GPU (state_mod.F90:258): rho and optional derivatives
__global__ void k_state_points(DevGrid g, int kk, const double *__restrict__ T, const double *__restrict__ S, double *__restrict__ rho,
                               double *__restrict__ drdt, double *__restrict__ drds, long long n) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const MwjfP P = mwjf_level(g.pressz[kk]);
  double a, b;
  rho[p] = mwjf_rho<true>(P, T[p], S[p], &a, &b);
  if (drdt) drdt[p] = a;
  if (drds) drds[p] = b;
}
inline int mix_state_host(const HostModel &h, const DevGrid &g, int kk, const double *T, const double *S, double *rho, double *drdt,
                          double *drds, long long n, hipStream_t st, std::string &err) {
  if (kk < 1 || kk > h.km) { err = "state: kk out of range"; return 1; }
  double *d = nullptr;
  if (hipMalloc((void **)&d, sizeof(double) * 5 * n) != hipSuccess) { err = "state: hipMalloc failed"; return 1; }
  hipMemcpyAsync(d, T, sizeof(double) * n, hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d + n, S, sizeof(double) * n, hipMemcpyHostToDevice, st);
  hipLaunchKernelGGL(k_state_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, kk, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, n);
  hipMemcpyAsync(rho, d + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost, st);
  if (drdt) hipMemcpyAsync(drdt, d + 3 * n, sizeof(double) * n, hipMemcpyDeviceToHost, st);
  if (drds) hipMemcpyAsync(drds, d + 4 * n, sizeof(double) * n, hipMemcpyDeviceToHost, st);
  const hipError_t e = hipStreamSynchronize(st);
  hipFree(d);
  if (e != hipSuccess) { err = std::string("state: ") + hipGetErrorString(e); return 1; }
  return 0;
}

}  // namespace pop

#include "kernels_del4.hpp"
#include "kernels_kpp.hpp"

namespace pop {

inline int mix_create(HostModel &h, const DevGrid &g, MixDev &m, std::vector<void *> &allocs, std::string &err) {
  if ((h.c.hmix_momentum == 4 || h.c.hmix_tracer == 4) && del4_create(h, g, m, allocs, err)) return 1;
  if (h.c.vmix_choice == 3 && kpp_create(h, g, m, allocs, err)) return 1;
  return 0;
}
inline int mix_vmix_coeffs(const HostModel &h, const DevGrid &g, const StepParams &sp, const MixDev &m, const MixState &s,
                           hipStream_t st, std::string &err) {
  const dim3 G3((g.n2 + 255) / 256, g.km, g.nblocks);
  if (h.c.vmix_choice == 2) {
    if (g.pbc) hipLaunchKernelGGL(k_rich_t<true>, G3, dim3(256), 0, st, g, sp, s, s.S3c);
    else hipLaunchKernelGGL(k_rich_t<false>, G3, dim3(256), 0, st, g, sp, s, s.S3c);
    hipLaunchKernelGGL(k_rich_u, G3, dim3(256), 0, st, g, sp, s, (const double *)s.S3c);
    return 0;
  }
  if (h.c.vmix_choice == 3) return kpp_vmix_coeffs(h, g, sp, m, s, st, err);
  err = "unknown vmix_choice";
  return 1;
}

}  // namespace pop
