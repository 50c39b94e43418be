// kernels_evp.hpp -- EVP block preconditioner on the device (POP_SolversMod.F90:2268-2369 preconditioner,
// :2618-2696 ExplicitEvp; preprocessing in host_evp.cpp).
//
// One thread owns one sub-block (at most 8x8 cells + rim): the solve is two sequential marching sweeps with a
// small dense correction in between, in exactly the reference's operation order, so the only parallelism is
// across sub-blocks (n2/64 of them per block).  The marching array lives in LDS as y[cell][thread] (bank-conflict
// free, 50 KB per 64-thread workgroup); the per-sub-block coefficients are stored [cell][sub-block] so that a
// wave reads them coalesced.  Ghost cells of PX are never written here: the buffer starts zeroed and the halo
// update that follows every application (:1352, :1719, :2023, :2130) fills them, as in the reference where PX is
// zeroed first.
#pragma once

namespace pop {

#define POP_EVP_THREADS 64

struct EvpDev {
  long long S;                    // local sub-blocks (stride of the coefficient arrays)
  const int4 *meta;               // x: cell index of (1,1) of the sub-block with rim; y: n | m << 8; z: land (diagonal scaling); w: not a single ocean cell
  const double *cc, *ne, *icc, *ine;   // [EVP_LD*EVP_LD][S]
  const double *rinv;             // [EVP_LE*EVP_LE][S]
  const double *C0, *WNE;         // r3: the 2-D fields cc and ne are copies of (centre weight at set-up, NE weight), for k_evp_apply_wave2
};

__global__ void __launch_bounds__(POP_EVP_THREADS)
k_evp_apply(EvpDev e, int nxb, const double *__restrict__ X, double *__restrict__ PX) {
  __shared__ double ys[EVP_LD * EVP_LD * POP_EVP_THREADS];
  const int t = threadIdx.x;
  const long long s = (long long)blockIdx.x * POP_EVP_THREADS + t;
  if (s >= e.S) return;
  const int4 mt = e.meta[s];
  const int n = mt.y & 255, m = mt.y >> 8;
  auto cell = [&](int a, int c) { return (long long)mt.x + (long long)(c - 1) * nxb + (a - 1); };
  auto co = [&](const double *A, int a, int c) { return A[(long long)((a - 1) + EVP_LD * (c - 1)) * e.S + s]; };
  if (mt.z) {   // sub-block with land: diagonal scaling (:2344-2348)
    for (int c = 2; c <= m - 1; ++c) for (int a = 2; a <= n - 1; ++a) PX[cell(a, c)] = X[cell(a, c)] * co(e.icc, a, c);
    return;
  }
  auto y = [&](int a, int c) -> double & { return ys[((a - 1) + EVP_LD * (c - 1)) * POP_EVP_THREADS + t]; };
  for (int c = 1; c <= m; ++c) for (int a = 1; a <= n; ++a) y(a, c) = 0.0;
  auto sweep = [&](int imax, int jmax) {
    for (int j = 2; j <= jmax; ++j)
      for (int i = 2; i <= imax; ++i)
        y(i + 1, j + 1) = (X[cell(i, j)] - co(e.cc, i, j) * y(i, j) - co(e.ne, i, j - 1) * y(i + 1, j - 1) -
                           co(e.ne, i - 1, j) * y(i - 1, j + 1) - co(e.ne, i - 1, j - 1) * y(i - 1, j - 1)) * co(e.ine, i, j);
  };
  sweep(n - 1, m - 1);
  // what reached the north / east rim, then the corrected values on the west column and the south row (:2667-2680)
  const int nm = n + m - 5;
  auto r = [&](int k) { return k <= n - 2 ? y(k + 2, m) : y(n, m - (k - (n - 2))); };
  auto rinv = [&](int k, int j) { return e.rinv[(long long)((k - 1) + EVP_LE * (j - 1)) * e.S + s]; };
  for (int jj = 1; jj <= m - 2; ++jj) {
    double acc = y(2, m - jj);
    for (int k = 1; k <= nm; ++k) acc = acc + rinv(k, jj) * r(k);
    y(2, m - jj) = acc;
  }
  for (int ii = 1; ii <= n - 3; ++ii) {
    double acc = y(ii + 2, 2);
    for (int k = 1; k <= nm; ++k) acc = acc + rinv(k, m - 2 + ii) * r(k);
    y(ii + 2, 2) = acc;
  }
  sweep(n - 2, m - 2);
  for (int c = 2; c <= m - 1; ++c) for (int a = 2; a <= n - 1; ++a) PX[cell(a, c)] = y(a, c);
}


// ---- the same solve with the marching sweeps as anti-diagonal wavefronts (round 3) ------------------------------------------
// y(i+1,j+1) is formed from y(i,j), y(i+1,j-1), y(i-1,j+1) (all on the anti-diagonal i + j) and y(i-1,j-1) (two diagonals
// back): every cell of an anti-diagonal can be formed at once, 2 (n - 2) - 1 <= 15 steps per sweep instead of (n - 2)^2 <= 64.
// Eight lanes own a sub-block (one per cell of the diagonal; a diagonal holds at most eight), eight sub-blocks share a
// wave; the coefficients (cc, ne, 1/ne), X and y of the sub-block sit in LDS, loaded with the eight sub-blocks side by side
// (64-byte segments), and the 13 x 13 correction between the sweeps is spread over the lanes row by row, each row summed in
// the reference's order.  Every value is formed by the expression of k_evp_apply with the same operands: bitwise equal
// (tests/test_gpu_parity.py).  A workgroup is exactly one wavefront; its barriers only order the LDS traffic.
#define POP_EVP_SB 8
constexpr int EVP_CELLS = EVP_LD * EVP_LD, EVP_PAD = EVP_CELLS + 5;
__global__ void __launch_bounds__(64)
k_evp_apply_wave(EvpDev e, int nxb, const double *__restrict__ X, double *__restrict__ PX) {
  __shared__ double ys[POP_EVP_SB][EVP_PAD], xs[POP_EVP_SB][EVP_PAD];
  __shared__ double ccs[POP_EVP_SB][EVP_PAD], nes[POP_EVP_SB][EVP_PAD], ins[POP_EVP_SB][EVP_PAD];
  __shared__ double rs[POP_EVP_SB][EVP_LE + 1];
  const int t = threadIdx.x, w = t >> 3, l = t & 7;
  const long long s0 = (long long)blockIdx.x * POP_EVP_SB, s = s0 + w;
  const bool live = s < e.S;
  int4 mt = make_int4(0, 3 | (3 << 8), 1, 0);
  if (live) mt = e.meta[s];
  const int n = mt.y & 255, m = mt.y >> 8;
  const bool solve = live && !mt.z;
  auto cell = [&](int a, int c) { return (long long)mt.x + (long long)(c - 1) * nxb + (a - 1); };
  auto at = [](int a, int c) { return (a - 1) + EVP_LD * (c - 1); };
  // coefficients: cell q of the eight sub-blocks of this workgroup lies at q * S + s0 .. + 7: lane (q8, sb) = (t >> 3, t & 7)
  {
    const int sb = t & 7, q8 = t >> 3;
    const bool ok = s0 + sb < e.S;
    for (int q = q8; q < EVP_CELLS; q += 8) {
      double c0 = 0.0, c1 = 0.0, c2 = 0.0;
      if (ok) { const long long o = (long long)q * e.S + s0 + sb; c0 = e.cc[o]; c1 = e.ne[o]; c2 = e.ine[o]; }
      ccs[sb][q] = c0; nes[sb][q] = c1; ins[sb][q] = c2; ys[sb][q] = 0.0;
    }
  }
  // X of the interior, row by row (lane = column); sub-blocks with land are scaled by the diagonal at once (:2344-2348)
  const int a0 = 2 + l;
  for (int c = 2; c <= EVP_LD - 1; ++c) {
    if (live && c <= m - 1 && a0 <= n - 1) {
      const double x = X[cell(a0, c)];
      xs[w][at(a0, c)] = x;
      if (mt.z) PX[cell(a0, c)] = x * e.icc[(long long)at(a0, c) * e.S + s];
    }
  }
  // rows of the correction this lane sums (row index jj = 1 .. n + m - 5; see below), requested before the first sweep
  const int nm = n + m - 5;
  double rv0[EVP_LE], rv1[EVP_LE];
  const int row0 = 1 + l, row1 = 9 + l;
#pragma unroll
  for (int k = 1; k <= EVP_LE; ++k) {
    rv0[k - 1] = (solve && row0 <= nm && k <= nm) ? e.rinv[(long long)((k - 1) + EVP_LE * (row0 - 1)) * e.S + s] : 0.0;
    rv1[k - 1] = (solve && row1 <= nm && k <= nm) ? e.rinv[(long long)((k - 1) + EVP_LE * (row1 - 1)) * e.S + s] : 0.0;
  }
  __syncthreads();
  auto sweep = [&](int imax, int jmax) {
    for (int d = 4; d <= 2 * (EVP_LD - 1); ++d) {
      const int ilo = (d - jmax > 2) ? d - jmax : 2, ihi = (d - 2 < imax) ? d - 2 : imax;
      const int i = ilo + l, j = d - i;
      const bool on = solve && d <= imax + jmax && i <= ihi;
      double v = 0.0;
      if (on)
        v = (xs[w][at(i, j)] - ccs[w][at(i, j)] * ys[w][at(i, j)] - nes[w][at(i, j - 1)] * ys[w][at(i + 1, j - 1)] -
             nes[w][at(i - 1, j)] * ys[w][at(i - 1, j + 1)] - nes[w][at(i - 1, j - 1)] * ys[w][at(i - 1, j - 1)]) * ins[w][at(i, j)];
      if (on) ys[w][at(i + 1, j + 1)] = v;
      __syncthreads();
    }
  };
  sweep(n - 1, m - 1);
  // what reached the north / east rim (:2667-2680)
  for (int k = 1 + l; k <= EVP_LE; k += 8)
    if (solve && k <= nm) rs[w][k] = (k <= n - 2) ? ys[w][at(k + 2, m)] : ys[w][at(n, m - (k - (n - 2)))];
  __syncthreads();
  // corrected values on the west column (rows jj = 1 .. m-2 -> y(2, m-jj)) and the south row (jj = m-2+ii -> y(ii+2, 2))
  auto target = [&](int jj) { return (jj <= m - 2) ? at(2, m - jj) : at(jj - (m - 2) + 2, 2); };
  if (solve && row0 <= nm) {
    double acc = ys[w][target(row0)];
#pragma unroll
    for (int k = 1; k <= EVP_LE; ++k) if (k <= nm) acc = acc + rv0[k - 1] * rs[w][k];
    ys[w][target(row0)] = acc;
  }
  if (solve && row1 <= nm) {
    double acc = ys[w][target(row1)];
#pragma unroll
    for (int k = 1; k <= EVP_LE; ++k) if (k <= nm) acc = acc + rv1[k - 1] * rs[w][k];
    ys[w][target(row1)] = acc;
  }
  __syncthreads();
  sweep(n - 2, m - 2);
  for (int c = 2; c <= EVP_LD - 1; ++c)
    if (solve && c <= m - 1 && a0 <= n - 1) PX[cell(a0, c)] = ys[w][at(a0, c)];
}


// ---- wavefront form, second version (r3): more waves per CU, fewer bytes ---------------------------------------------------
// k_evp_apply_wave keeps five coefficient / operand arrays of the eight sub-blocks in LDS (34.6 KB): four one-wave workgroups per
// CU, i.e. one wave per SIMD on a kernel that is a chain of ~80 dependent LDS steps -- 352 us per application at tx0.1v3 for
// ~130 us of bytes.  Here a lane owns one COLUMN of the marching index (i = 2 + lane) and meets the anti-diagonal d = i + j at
// step d: what it needs at that step -- cc(i,j), X(i,j), 1/ne(i,j) -- depends on the lane only through d, so it sits in
// registers indexed by the step (15 slots each), loaded up front; only y and ne (read at three neighbours) stay in LDS (14 KB: eleven
// workgroups per CU).  cc and ne are read from the 2-D fields they were copied from (C0, WNE: 64 cells + rim through the cache
// instead of 2 x 100 duplicated words per sub-block) and 1/ne is formed here (IEEE division: the bits of the host's table), so a
// sub-block costs 169 (correction matrix) + ~130 coefficient words instead of 469.  Same expressions, same operands: bitwise
// k_evp_apply (tests/test_gpu_parity.py).
constexpr int EVP_STEPS = 2 * (EVP_LD - 1) - 3;   // anti-diagonals d = 4 .. 2 (EVP_LD - 1)
__global__ void __launch_bounds__(64)
k_evp_apply_wave2(EvpDev e, int nxb, const double *__restrict__ X, double *__restrict__ PX) {
  __shared__ double ys[POP_EVP_SB][EVP_PAD], nes[POP_EVP_SB][EVP_PAD];
  __shared__ double rs[POP_EVP_SB][EVP_LE + 1];
  const int t = threadIdx.x, w = t >> 3, l = t & 7;
  const long long s = (long long)blockIdx.x * POP_EVP_SB + w;
  const bool live = s < e.S;
  int4 mt = make_int4(0, 3 | (3 << 8), 1, 0);
  if (live) mt = e.meta[s];
  const int n = mt.y & 255, m = mt.y >> 8;
  const bool solve = live && !mt.z;
  auto cell = [&](int a, int c) { return (long long)mt.x + (long long)(c - 1) * nxb + (a - 1); };
  auto at = [](int a, int c) { return (a - 1) + EVP_LD * (c - 1); };
  // ne of the sub-block with its rim (rows c = 1 .. m; lane = column, lanes 0 and 1 also take columns 9 and 10), y = 0
  for (int c = 1; c <= EVP_LD; ++c) {
    const int a1 = 1 + l, a2 = 9 + l;
    double v1 = 0.0, v2 = 0.0;
    if (live && c <= m && a1 <= n) v1 = e.WNE[cell(a1, c)];
    if (live && c <= m && l < 2 && a2 <= n) v2 = e.WNE[cell(a2, c)];
    nes[w][at(a1, c)] = v1; ys[w][at(a1, c)] = 0.0;
    if (l < 2) { nes[w][at(a2, c)] = v2; ys[w][at(a2, c)] = 0.0; }
  }
  // sub-blocks with land: diagonal scaling (:2344-2348), row by row
  const int a0 = 2 + l;
  if (live && mt.z) {
    for (int c = 2; c <= EVP_LD - 1; ++c)
      if (c <= m - 1 && a0 <= n - 1) {
        const double cc = e.C0[cell(a0, c)];
        PX[cell(a0, c)] = X[cell(a0, c)] * ((cc != 0.0) ? 1.0 / cc : 0.0);
      }
  }
  // the operands this lane (column i = 2 + l) meets at step d = i + j: X(i,j), cc(i,j) in registers indexed by the step
  const int i = 2 + l;
  double xx[EVP_STEPS], cs[EVP_STEPS];
#pragma unroll
  for (int q = 0; q < EVP_STEPS; ++q) {
    const int j = q + 4 - i;
    const bool ok = solve && j >= 2 && j <= m - 1 && i <= n - 1;
    xx[q] = ok ? X[cell(i, j)] : 0.0;
    cs[q] = ok ? e.C0[cell(i, j)] : 0.0;
  }
  const int nm = n + m - 5;
  double rv0[EVP_LE], rv1[EVP_LE];
  const int row0 = 1 + l, row1 = 9 + l;
#pragma unroll
  for (int k = 1; k <= EVP_LE; ++k) {
    rv0[k - 1] = (solve && row0 <= nm && k <= nm) ? e.rinv[(long long)((k - 1) + EVP_LE * (row0 - 1)) * e.S + s] : 0.0;
    rv1[k - 1] = (solve && row1 <= nm && k <= nm) ? e.rinv[(long long)((k - 1) + EVP_LE * (row1 - 1)) * e.S + s] : 0.0;
  }
  __syncthreads();
  double in[EVP_STEPS];         // 1 / ne(i, j) of the step (host: ine = 1 / ne where ne != 0, else 0)
#pragma unroll
  for (int q = 0; q < EVP_STEPS; ++q) {
    const int j = q + 4 - i;
    const bool ok = solve && j >= 2 && j <= m - 1 && i <= n - 1;
    const double nv = ok ? nes[w][at(i, j)] : 0.0;
    in[q] = (nv != 0.0) ? 1.0 / nv : 0.0;
  }
  auto sweep = [&](int imax, int jmax) {
#pragma unroll
    for (int q = 0; q < EVP_STEPS; ++q) {
      const int j = q + 4 - i;
      const bool on = solve && j >= 2 && j <= jmax && i <= imax;
      double v = 0.0;
      if (on)
        v = (xx[q] - cs[q] * ys[w][at(i, j)] - nes[w][at(i, j - 1)] * ys[w][at(i + 1, j - 1)] -
             nes[w][at(i - 1, j)] * ys[w][at(i - 1, j + 1)] - nes[w][at(i - 1, j - 1)] * ys[w][at(i - 1, j - 1)]) * in[q];
      if (on) ys[w][at(i + 1, j + 1)] = v;
      __syncthreads();
    }
  };
  sweep(n - 1, m - 1);
  for (int k = 1 + l; k <= EVP_LE; k += 8)
    if (solve && k <= nm) rs[w][k] = (k <= n - 2) ? ys[w][at(k + 2, m)] : ys[w][at(n, m - (k - (n - 2)))];
  __syncthreads();
  auto target = [&](int jj) { return (jj <= m - 2) ? at(2, m - jj) : at(jj - (m - 2) + 2, 2); };
  if (solve && row0 <= nm) {
    double acc = ys[w][target(row0)];
#pragma unroll
    for (int k = 1; k <= EVP_LE; ++k) if (k <= nm) acc = acc + rv0[k - 1] * rs[w][k];
    ys[w][target(row0)] = acc;
  }
  if (solve && row1 <= nm) {
    double acc = ys[w][target(row1)];
#pragma unroll
    for (int k = 1; k <= EVP_LE; ++k) if (k <= nm) acc = acc + rv1[k - 1] * rs[w][k];
    ys[w][target(row1)] = acc;
  }
  __syncthreads();
  sweep(n - 2, m - 2);
  for (int c = 2; c <= EVP_LD - 1; ++c)
    if (solve && c <= m - 1 && a0 <= n - 1) PX[cell(a0, c)] = ys[w][at(a0, c)];
}

// ---- wavefront form, third version (r4): the same solve with every load of the front end issued up front ------------------------------------
// k_evp_apply_wave2 requests its operands behind their conditions: one row of ne, one step's X and cc, one entry of the correction matrix
// at a time, each `if (...) v = load` a branch with its own s_waitcnt -- ~50 dependent round trips before the first marching step
// (90 s_waitcnt vmcnt in the ISA; gx1v7: 11.6 us per application for ~3 us of marching).  Here every load is unconditional at a clamped
// address and the conditions select values (the rewriting of the solver kernels' rim paths, 3d); X and cc of the interior are loaded once for
// both uses (the solve and the diagonal scaling of sub-blocks with land).  Same expressions on the same operands: bitwise.
// RESIDUAL: X is a residual of the solvers -- exactly zero on land, so a sub-block without a single ocean cell (meta.w) yields zeros (x * icc
// of the diagonal scaling): its loads are not issued, and a wave whose eight sub-blocks are all of that kind stores its zeros and leaves.
template <bool RESIDUAL>
__global__ void __launch_bounds__(64)
k_evp_apply_wave3(EvpDev e, int nxb, const double *__restrict__ X, double *__restrict__ PX) {
  __shared__ double ys[POP_EVP_SB][EVP_PAD], nes[POP_EVP_SB][EVP_PAD];
  __shared__ double rs[POP_EVP_SB][EVP_LE + 1];
  const int t = threadIdx.x, w = t >> 3, l = t & 7;
  const long long s = (long long)blockIdx.x * POP_EVP_SB + w;
  const bool live = s < e.S;
  const long long sl = live ? s : 0;
  const int4 mt0 = e.meta[sl];
  const int4 mt = live ? mt0 : make_int4(0, 3 | (3 << 8), 1, 0);
  const int n = mt.y & 255, m = mt.y >> 8;
  const bool solve = live && !mt.z;
  auto cell = [&](int a, int c) { return (long long)mt.x + (long long)(c - 1) * nxb + (a - 1); };
  auto at = [](int a, int c) { return (a - 1) + EVP_LD * (c - 1); };
  const long long qs = mt.x;   // a cell that exists (0 for a lane without a sub-block)
  const bool allland = RESIDUAL && live && mt.w != 0;
  if (RESIDUAL && __all(allland || !live)) {
    const int a0 = 2 + (t & 7);
    for (int c = 2; c <= EVP_LD - 1; ++c)
      if (allland && c <= m - 1 && a0 <= n - 1) PX[cell(a0, c)] = 0.0;
    return;
  }
  // ---- requests: ne with its rim, X and cc of the lane's column, the lane's two rows of the correction matrix
  const int i = 2 + l;
  // ne with its rim: the 100 cells of the sub-block dealt to its eight lanes in linear order (cell t = l + 8 k: 13 loads, not 2 x 10 by row)
  constexpr int NEK = (EVP_CELLS + 7) / 8;
  double nv[NEK];
#pragma unroll
  for (int k = 0; k < NEK; ++k) {
    const int tt = l + 8 * k, a = tt % EVP_LD + 1, c = tt / EVP_LD + 1;
    const bool ok = live && !allland && tt < EVP_CELLS && c <= m && a <= n;
    nv[k] = e.WNE[ok ? cell(a, c) : qs];
  }
  // X and cc of the lane's column by ROW (8 loads each, not one per step: the kernel is bound by its vector-memory instructions), moved to
  // the slot of the step that meets the row (row j at step q = j + i - 4) by selects below
  double xr[EVP_LD - 2], cr[EVP_LD - 2];
#pragma unroll
  for (int jr = 0; jr < EVP_LD - 2; ++jr) {
    const int j = 2 + jr;
    const bool ok = live && !allland && j <= m - 1 && i <= n - 1;
    const long long qa = ok ? cell(i, j) : qs;
    xr[jr] = X[qa]; cr[jr] = e.C0[qa];
  }
  const int nm = n + m - 5;
  double rv0[EVP_LE], rv1[EVP_LE];
  const int row0 = 1 + l, row1 = 9 + l;
#pragma unroll
  for (int k = 1; k <= EVP_LE; ++k) {
    const bool ok0 = solve && row0 <= nm && k <= nm, ok1 = solve && row1 <= nm && k <= nm;
    rv0[k - 1] = e.rinv[(ok0 ? (long long)((k - 1) + EVP_LE * (row0 - 1)) * e.S : 0) + sl];
    rv1[k - 1] = e.rinv[(ok1 ? (long long)((k - 1) + EVP_LE * (row1 - 1)) * e.S : 0) + sl];
  }
  // ---- values
#pragma unroll
  for (int k = 0; k < NEK; ++k) {
    const int tt = l + 8 * k, a = tt % EVP_LD + 1, c = tt / EVP_LD + 1;
    const bool ok = live && tt < EVP_CELLS && c <= m && a <= n;
    if (tt < EVP_CELLS) { nes[w][tt] = ok ? nv[k] : 0.0; ys[w][tt] = 0.0; }   // (at(a, c) = tt)
  }
#pragma unroll
  for (int k = 1; k <= EVP_LE; ++k) {
    if (!(solve && row0 <= nm && k <= nm)) rv0[k - 1] = 0.0;
    if (!(solve && row1 <= nm && k <= nm)) rv1[k - 1] = 0.0;
  }
#pragma unroll
  for (int jr = 0; jr < EVP_LD - 2; ++jr) {
    const int j = 2 + jr;
    const bool ok = live && j <= m - 1 && i <= n - 1;
    if (ok && mt.z) PX[cell(i, j)] = allland ? 0.0 : xr[jr] * ((cr[jr] != 0.0) ? 1.0 / cr[jr] : 0.0);   // sub-blocks with land: diagonal scaling (:2344-2348)
    if (!(ok && solve)) { xr[jr] = 0.0; cr[jr] = 0.0; }
  }
  double xx[EVP_STEPS], cs[EVP_STEPS];
#pragma unroll
  for (int q = 0; q < EVP_STEPS; ++q) {
    double xv = 0.0, cv = 0.0;
#pragma unroll
    for (int jr = 0; jr < EVP_LD - 2; ++jr)
      if (q - jr >= 0 && q - jr < 8) { const bool hit = (l == q - jr); xv = hit ? xr[jr] : xv; cv = hit ? cr[jr] : cv; }   // j = q + 4 - i  <=>  jr = q - l
    xx[q] = xv; cs[q] = cv;
  }
  __syncthreads();
  double in[EVP_STEPS];         // 1 / ne(i, j) of the step (host: ine = 1 / ne where ne != 0, else 0)
#pragma unroll
  for (int q = 0; q < EVP_STEPS; ++q) {
    const int j = q + 4 - i;
    const bool ok = solve && j >= 2 && j <= m - 1 && i <= n - 1;
    const double nv = ok ? nes[w][at(i, j)] : 0.0;
    in[q] = (nv != 0.0) ? 1.0 / nv : 0.0;
  }
  auto sweep = [&](int imax, int jmax) {
#pragma unroll
    for (int q = 0; q < EVP_STEPS; ++q) {
      const int j = q + 4 - i;
      const bool on = solve && j >= 2 && j <= jmax && i <= imax;
      double v = 0.0;
      if (on)
        v = (xx[q] - cs[q] * ys[w][at(i, j)] - nes[w][at(i, j - 1)] * ys[w][at(i + 1, j - 1)] -
             nes[w][at(i - 1, j)] * ys[w][at(i - 1, j + 1)] - nes[w][at(i - 1, j - 1)] * ys[w][at(i - 1, j - 1)]) * in[q];
      if (on) ys[w][at(i + 1, j + 1)] = v;
      __syncthreads();
    }
  };
  sweep(n - 1, m - 1);
  for (int k = 1 + l; k <= EVP_LE; k += 8)
    if (solve && k <= nm) rs[w][k] = (k <= n - 2) ? ys[w][at(k + 2, m)] : ys[w][at(n, m - (k - (n - 2)))];
  __syncthreads();
  auto target = [&](int jj) { return (jj <= m - 2) ? at(2, m - jj) : at(jj - (m - 2) + 2, 2); };
  if (solve && row0 <= nm) {
    double acc = ys[w][target(row0)];
#pragma unroll
    for (int k = 1; k <= EVP_LE; ++k) if (k <= nm) acc = acc + rv0[k - 1] * rs[w][k];
    ys[w][target(row0)] = acc;
  }
  if (solve && row1 <= nm) {
    double acc = ys[w][target(row1)];
#pragma unroll
    for (int k = 1; k <= EVP_LE; ++k) if (k <= nm) acc = acc + rv1[k - 1] * rs[w][k];
    ys[w][target(row1)] = acc;
  }
  __syncthreads();
  sweep(n - 2, m - 2);
  const int a0 = 2 + l;
  for (int c = 2; c <= EVP_LD - 1; ++c)
    if (solve && c <= m - 1 && a0 <= n - 1) PX[cell(a0, c)] = ys[w][at(a0, c)];
}

// ---- P-CSI iteration + sub-block solves in ONE launch (round 4) ---------------------------------------------------------------------------
// With the EVP preconditioner a P-CSI iteration was two launches: k_pcsi_step2 (dx, x, r = b - A x; the residual itself to memory) and
// k_evp_apply_wave2 (r' = M^-1 r, sub-block by sub-block).  Here the wave that solves eight sub-blocks first forms what the step kernel forms
// for them: x_new = x + (omega r' + (csy omega - 1) dx) on every sub-block WITH its rim (100 cells; a rim cell that is a ghost cell of the block
// is formed at its source cell, the fill value where there is none) into the LDS array the marching sweeps use afterwards, then r = b - A x_new on
// the 8 x 8 interior -- lane = column, so r(i, j) lands in the register slot of the anti-diagonal step that consumes it -- and then the solve of
// k_evp_apply_wave2.  The expressions and their order are those of k_pcsi_step2 / k_evp_apply_wave2: bitwise the two launches
// (tests/test_gpu_parity.py).  r' ping-pongs like x and dx (a neighbouring sub-block still reads the old one).  RAW: the residual itself also
// goes to `raw`, from which k_pcsi_rr_chunks forms the chunk partials of (r, r) for the check.
// MEASURED SLOWER than the two launches (gx1v7 28.9 against 22.7 us per iteration, tx0.1v3 515 against 283): the step's loads -- issued by
// 4 waves per SIMD in k_pcsi_step2 -- come here from the one wave per eight sub-blocks of the solve, row by row behind their conditions, and
// the launch loses the land elimination of the step kernel.  Kept behind pop_tuning.pcsi_evp_fused = 1 (default off) as the bitwise-checked
// starting point of a version with every load issued up front.
template <bool RAW>
__global__ void __launch_bounds__(64)
k_pcsi_evp_step(EvpDev e, DevGrid g, PcsiArgs a, double *__restrict__ raw) {
  if (a.sc->stop) return;   // an earlier check has converged: this launch belongs to the look-ahead interval
  __shared__ double ys[POP_EVP_SB][EVP_PAD], nes[POP_EVP_SB][EVP_PAD];
  __shared__ double rs[POP_EVP_SB][EVP_LE + 1];
  const int nxb = g.nxb;
  const int t = threadIdx.x, w = t >> 3, l = t & 7;
  const long long s = (long long)blockIdx.x * POP_EVP_SB + w;
  const bool live = s < e.S;
  int4 mt = make_int4(0, 3 | (3 << 8), 1, 0);
  if (live) mt = e.meta[s];
  const int n = mt.y & 255, m = mt.y >> 8;
  const bool solve = live && !mt.z;
  auto cell = [&](int a_, int c) { return (long long)mt.x + (long long)(c - 1) * nxb + (a_ - 1); };
  auto at = [](int a_, int c) { return (a_ - 1) + EVP_LD * (c - 1); };
  const double om = a.omega[*a.base + a.j], cq = a.csy * om - 1.0;
  const long long qsafe = cell(2, 2);
  // position of (1,1) in its block: the sub-blocks tile the NOMINAL block, so on padded blocks (blocks.F90:174-265) a sub-block can hold
  // cells beyond the block's own extent (ghost cells and cells of no block at all): only physical cells are advanced in place and stored
  const int bk = (int)(mt.x / g.n2), p11 = (int)(mt.x - (long long)bk * g.n2), i11 = p11 % nxb, j11 = p11 / nxb;
  auto phys = [&](int a_, int c) { return interior(g, bk, i11 + a_ - 1, j11 + c - 1); };
  // ---- x_new of the sub-block with its rim -> ys; dx_new, x_new of the interior -> memory; ne with its rim -> nes
  auto advance = [&](int a_, int c) {
    double xn = 0.0, nev = 0.0;
    if (live && c <= m && a_ <= n) {
      const long long q = cell(a_, c);
      const bool rim = a_ == 1 || a_ == n || c == 1 || c == m, own = phys(a_, c);
      const long long src = own ? q : (long long)a.srcmap[q];
      const long long qq = (src >= 0) ? src : qsafe;
      const double dx = om * a.Ri[qq] + cq * a.Qi[qq];
      const double x = a.Xi[qq] + dx;
      xn = (src >= 0) ? x : 0.0;
      nev = e.WNE[q];
      if (!rim && own) { a.Qo[q] = dx; a.Xo[q] = x; }
    }
    ys[w][at(a_, c)] = xn; nes[w][at(a_, c)] = nev;
  };
  for (int c = 1; c <= EVP_LD; ++c) {
    advance(1 + l, c);
    if (l < 2) advance(9 + l, c);
  }
  // the correction rows of this lane (as in k_evp_apply_wave2), requested before the first barrier
  const int i = 2 + l;
  const int nm = n + m - 5;
  double rv0[EVP_LE], rv1[EVP_LE];
  const int row0 = 1 + l, row1 = 9 + l;
#pragma unroll
  for (int k = 1; k <= EVP_LE; ++k) {
    rv0[k - 1] = (solve && row0 <= nm && k <= nm) ? e.rinv[(long long)((k - 1) + EVP_LE * (row0 - 1)) * e.S + s] : 0.0;
    rv1[k - 1] = (solve && row1 <= nm && k <= nm) ? e.rinv[(long long)((k - 1) + EVP_LE * (row1 - 1)) * e.S + s] : 0.0;
  }
  __syncthreads();
  // ---- r = b - A x_new on the interior: lane = column i, row j = step + 4 - i; the operator of k_pcsi_step2 (weights at the cell, its
  // southern and western neighbours; centre, N, S, E, W, NE, SE, NW, SW added in this order)
  double xx[EVP_STEPS], cs[EVP_STEPS];
#pragma unroll
  for (int q = 0; q < EVP_STEPS; ++q) {
    const int j = q + 4 - i;
    const bool in_sb = live && j >= 2 && j <= m - 1 && i <= n - 1, ok = in_sb && phys(i, j);
    double r = 0.0, c0 = 0.0;
    if (in_sb) c0 = e.C0[cell(i, j)];
    if (ok) {
      const long long qc = cell(i, j);
      const double wv[9] = {a.C[qc], g.WNo[qc], g.WNo[qc - nxb], g.WEa[qc], g.WEa[qc - 1], g.WNE[qc], g.WNE[qc - nxb], g.WNE[qc - 1], g.WNE[qc - 1 - nxb]};
      const double *X = ys[w];
      const double ax = wv[0] * X[at(i, j)] + wv[1] * X[at(i, j + 1)] + wv[2] * X[at(i, j - 1)] + wv[3] * X[at(i + 1, j)] + wv[4] * X[at(i - 1, j)] +
                        wv[5] * X[at(i + 1, j + 1)] + wv[6] * X[at(i + 1, j - 1)] + wv[7] * X[at(i - 1, j + 1)] + wv[8] * X[at(i - 1, j - 1)];
      r = a.Bv[qc] - ax;
      if (RAW) raw[qc] = r;
      if (mt.z) a.Ro[qc] = r * ((c0 != 0.0) ? 1.0 / c0 : 0.0);   // sub-blocks with land: diagonal scaling (:2344-2348)
    }
    xx[q] = solve ? r : 0.0; cs[q] = solve ? c0 : 0.0;
  }
  __syncthreads();
  // ---- the solve of k_evp_apply_wave2 from here on (y = 0 first: the array held x_new)
  for (int c = 1; c <= EVP_LD; ++c) {
    ys[w][at(1 + l, c)] = 0.0;
    if (l < 2) ys[w][at(9 + l, c)] = 0.0;
  }
  double in[EVP_STEPS];
#pragma unroll
  for (int q = 0; q < EVP_STEPS; ++q) {
    const int j = q + 4 - i;
    const bool ok = solve && j >= 2 && j <= m - 1 && i <= n - 1;
    const double nv = ok ? nes[w][at(i, j)] : 0.0;
    in[q] = (nv != 0.0) ? 1.0 / nv : 0.0;
  }
  __syncthreads();
  auto sweep = [&](int imax, int jmax) {
#pragma unroll
    for (int q = 0; q < EVP_STEPS; ++q) {
      const int j = q + 4 - i;
      const bool on = solve && j >= 2 && j <= jmax && i <= imax;
      double v = 0.0;
      if (on)
        v = (xx[q] - cs[q] * ys[w][at(i, j)] - nes[w][at(i, j - 1)] * ys[w][at(i + 1, j - 1)] -
             nes[w][at(i - 1, j)] * ys[w][at(i - 1, j + 1)] - nes[w][at(i - 1, j - 1)] * ys[w][at(i - 1, j - 1)]) * in[q];
      if (on) ys[w][at(i + 1, j + 1)] = v;
      __syncthreads();
    }
  };
  sweep(n - 1, m - 1);
  for (int k = 1 + l; k <= EVP_LE; k += 8)
    if (solve && k <= nm) rs[w][k] = (k <= n - 2) ? ys[w][at(k + 2, m)] : ys[w][at(n, m - (k - (n - 2)))];
  __syncthreads();
  auto target = [&](int jj) { return (jj <= m - 2) ? at(2, m - jj) : at(jj - (m - 2) + 2, 2); };
  if (solve && row0 <= nm) {
    double acc = ys[w][target(row0)];
#pragma unroll
    for (int k = 1; k <= EVP_LE; ++k) if (k <= nm) acc = acc + rv0[k - 1] * rs[w][k];
    ys[w][target(row0)] = acc;
  }
  if (solve && row1 <= nm) {
    double acc = ys[w][target(row1)];
#pragma unroll
    for (int k = 1; k <= EVP_LE; ++k) if (k <= nm) acc = acc + rv1[k - 1] * rs[w][k];
    ys[w][target(row1)] = acc;
  }
  __syncthreads();
  sweep(n - 2, m - 2);
  const int a0 = 2 + l;
  for (int c = 2; c <= EVP_LD - 1; ++c)
    if (solve && c <= m - 1 && a0 <= n - 1) a.Ro[cell(a0, c)] = ys[w][at(a0, c)];
}

}  // namespace pop
