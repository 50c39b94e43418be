// device_types.hpp -- device-side view of the model passed by value to kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "pop_internal.hpp"

namespace pop {

// Init-time constant arrays that kernels index with WAVE-UNIFORM indices (the level k of a column march): dz(k), dzr(k), afac_t(k),
// zgrid(k) ...  Read through a plain `const double *` inside a kernel that also stores to global memory, the compiler cannot prove
// the array is not clobbered, emits a VECTOR load with a uniform address and -- the load sitting in the middle of the level's
// arithmetic -- an `s_waitcnt vmcnt(0)` behind it, which also drains every prefetch load in flight (seen in the ISA of the
// LDS-tiled stencil kernels: three such round trips per level).  Loads through the constant address space are selected as scalar
// (SMEM) loads into SGPRs: no VGPRs, no vmcnt.  Same bytes, same values; the arrays are written once at pop_create.
// A divergent index still works (the load is then an ordinary vector load).
template <class T>
struct ConstArr {
  const T *p = nullptr;
  __device__ __forceinline__ T operator[](int k) const { return ((const __attribute__((address_space(4))) T *)p)[k]; }
  // the caller asserts that k is the same in every lane: pointer and index pass through v_readfirstlane, so the load is a scalar
  // load wherever it stands (inside a loop whose induction variable the compiler moved into vector registers, operator[]
  // becomes a vector load followed by a full wait)
  __device__ __forceinline__ T u(int k) const {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    const unsigned ku = __builtin_amdgcn_readfirstlane((unsigned)k);
    return ((const __attribute__((address_space(4))) T *)(((unsigned long long)hi << 32) | lo))[(int)ku];
  }
  __host__ __device__ ConstArr &operator=(const T *q) { p = q; return *this; }
  __host__ __device__ operator const T *() const { return p; }
};
using CArr = ConstArr<double>;
using CArrI = ConstArr<int>;

// Time-invariant data resident in HBM.  2-D fields are (nxb,nyb,nblocks) i-fastest; vertical
// arrays are 1-based with slot 0 (index k is wave-uniform, so they are read by scalar loads).
struct DevGrid {
  int nxb, nyb, km, nt, nblocks, ib, ie, jb, je;   // ib..je are 1-based like the reference
  // padded blocks (blocks.F90:174-265): when the block size does not divide the domain the last column / row of blocks ends
  // early; ieb[b], jeb[b] = last physical i, j of local block b.  nullptr: every block ends at ie, je (kernels_common.hpp blk_ie / blk_je)
  const int *ieb, *jeb;
  int xcd_remap;                                   // column kernels: workgroup order (kernels_common.hpp col_setup)
  int red_band;                                    // 2-D reduction kernels: XCD-banded chunk order
  int red_tiles;                                   // 2-D reduction kernels: 64x4 tiles in XCD-strided columns
  int lds_order;                                   // LDS-tiled stencil kernels: 1 = XCD patch order (kernels_common.hpp lds_tile)
  int n2;                                          // nxb*nyb
  long long n3;                                    // n2*km
  CArr dz, dzw, zt, zw, c2dz, dzr, dz2r, dzwr, pressz, bouss, afac_t, afac_u;   // vertical arrays: scalar loads (ConstArr)
  CArr eosP;                                       // 6 (km + 2): the pressure-dependent MWJF coefficients of every level, formed on the device by mwjf_level itself (k_eos_level_table)
  int state_lv;                                    // k_state3d: levels per thread (0 / 1: one cell per thread with the coefficients formed in place)
  const double *DXU, *DYU, *DXUR, *DYUR, *UAREA_R, *TAREA_R, *TAREA, *FCOR, *FCORT, *HU, *HUR;
  const double *AU0, *AUN, *AUE, *AUNE, *RCALCT;
  const int *KMT, *KMU, *KMTN, *KMTS, *KMTE, *KMTW, *KMTEE, *KMTNN;
  const double *DTN, *DTS, *DTE, *DTW;
  // partial bottom cells (grid.F90:916-1020): thickness of the bottom T cell (DZBC) and of the bottom U cell (DZUB = DZU at level
  // KMU) of every column; every other level has dz(k).  pbc = 0: both null
  const double *DZBC, *DZUB;
  int pbc;
  double *dump;                                    // scratch words that inactive lanes of branch-free kernels store to
  const double *zero;                              // a few words of +0.0 that no kernel writes (loads that must yield zero without a branch)
  const double *DUC, *DUN, *DUS, *DUE, *DUW, *DMC, *DMN, *DMS, *DME, *DMW, *DUM, *KXU, *KYU;
  const double *WNE, *WEa, *WNo, *WC0, *mMask, *CHECKER, *CONSTNT;
  const double *XW, *YW;                           // U-point terms of the barotropic operator: WNE = XW + YW, WEa = XW + XW(j-1) - YW - YW(j-1), WNo = YW + YW(i-1) - XW - XW(i-1)
  const double *SMF1, *SMF2, *SMFT1, *SMFT2;
  const unsigned char *mMask8;                     // mMask (exactly 0 or 1) as bytes: the fused solver kernels read 1 B instead of 8
  // land elimination at workgroup-tile granularity (the reference drops land BLOCKS from the distribution,
  // distribution.F90 / domain.F90 'Eliminating land blocks'; here a block is as large as a GPU's share, so the unit is the tile
  // a workgroup works on).  opre[b*(n2+1) + p] = number of cells with KMT > 0 among the cells < p of block b (ghosts included).
  const int *opre;
  int skip;                                        // 1: workgroups whose tile holds no ocean cell return at once
  // fused solver kernels only (their launches are ~100 us, so even workgroups that return at once cost a quarter of it): the
  // launch covers red_nact workgroups per block and workgroup w of block b takes chunk red_act[b*red_nact + w] -- the chunks
  // with an ocean cell first, in increasing order, padded with land chunks of the same block
  // LDS-tiled stencil kernels (64 x R column tiles): launch order of the tiles with an ocean cell, tj << 16 | ti, -1 = padding.
  // Without the list the XCDs that draw land tiles idle while the in-order dispatcher waits for the others (measured: the
  // momentum kernel gained 22% from skipping 35% of its tiles)
  const int *lds_act4, *lds_act8;
  int lds_n4, lds_n8;
  const int *red_act, *red_cnt;                    // red_cnt[b]: how many entries of block b's list are chunks with ocean
  int red_nact;
};

// scalar parameters of the current step (step_mod.F90:302-320)
struct StepParams {
  double c2dtu, c2dtp, beta, gamma, dtp, grav;
  double am, ah, bottom_drag, const_vvc, const_vdc, convect_diff, convect_visc;
  double rich_bckgrnd_vvc, rich_bckgrnd_vdc, rich_mix, aidif;
  int leapfrogts, pavg, impcor, reset_to_freezing, nvdc;
};

}  // namespace pop
