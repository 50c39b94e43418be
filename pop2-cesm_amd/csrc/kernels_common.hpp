// kernels_common.hpp -- shared device helpers: MWJF equation of state, column indexing.
#pragma once
#include "device_types.hpp"

namespace pop {

#define POP_COL_THREADS 64   // one wavefront per workgroup for column-march kernels
// waves per SIMD the stencil column kernels are compiled for (second __launch_bounds__ argument)
#ifndef POP_TRC_WAVES
#define POP_TRC_WAVES 2
#endif
#ifndef POP_MOM_WAVES
#define POP_MOM_WAVES 2
#endif

// Partial bottom cells: DZT(i,j,k) / DZU(i,j,k) of the reference (grid.F90:926-1016) from the column's bottom level and bottom
// thickness -- the bottom cell's own thickness at k = kbot, dz(k) at every other level 1..km, 0 at the levels 0 and km+1 the
// reference's arrays carry.
__device__ __forceinline__ double pbc_dz(const DevGrid &g, int k, int kbot, double dzbot) {
  return (k < 1 || k > g.km) ? 0.0 : ((k == kbot) ? dzbot : g.dz[k]);
}

// last physical column / row of local block b (1-based): block-uniform unless the decomposition is padded
__device__ __forceinline__ int blk_ie(const DevGrid &g, int b) { return g.ieb ? g.ieb[b] : g.ie; }
__device__ __forceinline__ int blk_je(const DevGrid &g, int b) { return g.jeb ? g.jeb[b] : g.je; }

// Column-kernel prologue: one thread per (i,j) of local block b; returns false for threads
// outside the physical domain ib..ie, jb..je (or outside the block).
struct Col {
  int i, j, b, p2;        // 0-based i,j; p2 = j*nxb+i
  long long q2;           // b*n2 + p2
  long long base3;        // b*n3 + p2   (+ (k-1)*n2 for level k)
};
// XCD-aware tile order.  The hardware deals workgroups round-robin over the 8 XCDs (workgroup w runs on
// XCD w % 8, each with its own 4 MB L2; MI355X_MICROARCH.md), so with the natural order the rows j+-1 a
// stencil re-reads were fetched by workgroups of OTHER XCDs and every L2 fetches them again.
//   g.xcd_remap == 1 (small grids, everything co-resident): XCD x owns one contiguous band of the
//     linear tile order.
//   g.xcd_remap == 2 (large grids): tiles are blockDim.x wide, blockDim.y rows high; XCD x owns the
//     tile COLUMNS ti = x (mod 8) and walks them row by row, so all XCDs stream the same rows at the
//     same time (DRAM page locality of the natural order is kept) and the vertical neighbour of a
//     tile is the tile the same XCD touched a few workgroups earlier.  The ntx % 8 left-over columns
//     are dealt out tile by tile so every XCD gets the same number of workgroups.
// The launch grid's x size comes from tile_grid_x(); surplus workgroups idle.
struct TileId { int ti, tj; bool valid; };
__host__ __device__ inline int tile_grid_x(int nxb, int nyb, int tw, int th) {
  const int ntx = (nxb + tw - 1) / tw, nty = (nyb + th - 1) / th;
  const int full = ntx >> 3, rem = ntx & 7;
  return 8 * (full * nty + (rem * nty + 7) / 8);
}
__device__ __forceinline__ TileId tile_of_block(int nxb, int nyb, int tw, int th) {
  const int ntx = (nxb + tw - 1) / tw, nty = (nyb + th - 1) / th;
  const int full = ntx >> 3, rem = ntx & 7;
  const int x = blockIdx.x & 7, seq = blockIdx.x >> 3;
  TileId t;
  if (seq < full * nty) { t.tj = seq / full; t.ti = x + 8 * (seq % full); t.valid = true; }
  else {
    const int r = (seq - full * nty) * 8 + x;
    t.valid = r < rem * nty;
    t.tj = t.valid ? r / rem : 0; t.ti = t.valid ? 8 * full + r % rem : 0;
  }
  return t;
}
// Workgroup -> tile of the LDS-tiled stencil kernels (64 x R column tiles, ntx x nty of them per block).
// lds_order 1: the tiles are numbered patch by patch (patches of 4 x 8 tiles, row-major inside a patch, patches
// row-major), and runs of 32 consecutive numbers are dealt to the XCDs in turn (workgroup w runs on XCD w % 8, 32
// CUs each): the workgroups an XCD runs at the same time form a compact patch, so the halo rows / columns one tile
// shares with its neighbours are fetched into that XCD's L2 once instead of once per XCD.
__host__ __device__ inline int lds_grid_x(int lds_order, int ntx, int nty) {
  const int nt = ntx * nty;
  return lds_order ? 256 * ((nt + 255) / 256) : nt;
}
// position in the launch (blockIdx.x) of the a-th tile of the patch-major numbering: runs of 32 consecutive numbers go to one
// XCD (workgroup w runs on XCD w % 8)
__host__ __device__ inline int lds_slot_of(int a) {
  const int r = a >> 5, t = a & 31;
  return (((r >> 3) << 5) + t) * 8 + (r & 7);
}
// tile gi of the patch-major numbering
__host__ __device__ inline bool lds_tile_from_gi(int gi, int ntx, int nty, int &ti, int &tj);
__host__ __device__ inline bool lds_tile_of(int lds_order, int bx, int ntx, int nty, int &ti, int &tj) {
  if (!lds_order) { ti = bx % ntx; tj = bx / ntx; return tj < nty; }
  const int e = bx & 7, s = bx >> 3;
  const int gi = ((s >> 5) * 8 + e) * 32 + (s & 31);
  return lds_tile_from_gi(gi, ntx, nty, ti, tj);
}
__host__ __device__ inline bool lds_tile_from_gi(int gi, int ntx, int nty, int &ti, int &tj) {
  constexpr int PW = 4, PH = 8;
  if (gi >= ntx * nty) return false;
  const int row_tiles = PH * ntx, full_rows = nty / PH;
  int Pj, r, h;
  if (gi >= full_rows * row_tiles) { Pj = full_rows; r = gi - full_rows * row_tiles; h = nty - full_rows * PH; }
  else { Pj = gi / row_tiles; r = gi % row_tiles; h = PH; }
  const int patch_tiles = h * PW, nfullp = ntx / PW;
  int Pi, w, rr;
  if (r < nfullp * patch_tiles) { Pi = r / patch_tiles; w = PW; rr = r % patch_tiles; }
  else { Pi = nfullp; w = ntx - nfullp * PW; rr = r - nfullp * patch_tiles; }
  ti = Pi * PW + rr % w; tj = Pj * PH + rr / w;
  return true;
}
__device__ __forceinline__ bool lds_tile(int lds_order, int ntx, int nty, int &ti, int &tj) {
  return lds_tile_of(lds_order, (int)blockIdx.x, ntx, nty, ti, tj);
}
// the same through the list of tiles with an ocean cell (DevGrid::lds_act4/8) when land elimination is on
template <int R>
__device__ __forceinline__ bool lds_tile_active(const DevGrid &g, int b, int ntx, int nty, int &ti, int &tj, bool &listed) {
  const int *L = (R == 8) ? g.lds_act8 : g.lds_act4;
  const int nL = (R == 8) ? g.lds_n8 : g.lds_n4;
  listed = g.skip && L != nullptr;
  if (!listed) return lds_tile(g.lds_order, ntx, nty, ti, tj);
  const int e = L[(long long)b * nL + blockIdx.x];
  if (e < 0) return false;
  ti = e & 0xffff; tj = e >> 16;
  return true;
}
template <int R>
__host__ inline int lds_launch_x(const DevGrid &g, int ntx, int nty) {
  const int *L = (R == 8) ? g.lds_act8 : g.lds_act4;
  if (g.skip && L) return (R == 8) ? g.lds_n8 : g.lds_n4;
  return lds_grid_x(g.lds_order, ntx, nty);
}
// ---- land elimination: does the tile of this workgroup hold an ocean cell? -------------------------------------------
// A tile without one (KMT = 0 everywhere; KMU = min of four KMT is 0 there too) has nothing to compute: what the full
// kernels write there does not depend on the state (zeros, rho(0,0,p(k)), the vertical-mixing floor values), so it is
// written during the first steps after set-up (DevGrid::skip = 0) and left alone afterwards.  Arguments are uniform over
// the workgroup, so the look-ups are scalar loads.
__device__ __forceinline__ int ocean_cells(const DevGrid &g, int b, int p0, int p1) {   // cells p0 <= p < p1 of block b
  const int *S = g.opre + (long long)b * (g.n2 + 1);
  return S[p1] - S[p0];
}
__device__ __forceinline__ bool land_tile(const DevGrid &g, int b, int i0, int w, int j0, int h) {   // cells [i0, i0+w) x [j0, j0+h)
  if (!g.skip) return false;
  int i1 = i0 + w, j1 = j0 + h;
  if (i0 < 0) i0 = 0;
  if (j0 < 0) j0 = 0;
  if (i1 > g.nxb) i1 = g.nxb;
  if (j1 > g.nyb) j1 = g.nyb;
  if (i0 >= i1) return false;
  int n = 0;
  for (int j = j0; j < j1; ++j) n += ocean_cells(g, b, j * g.nxb + i0, j * g.nxb + i1);
  return n == 0;
}
__device__ __forceinline__ bool land_run(const DevGrid &g, int b, long long p0, int n) {   // n consecutive cells from p0
  if (!g.skip || p0 >= g.n2) return false;
  const int p1 = (p0 + n < g.n2) ? (int)(p0 + n) : g.n2;
  return ocean_cells(g, b, (int)p0, p1) == 0;
}
__host__ __device__ inline int col_grid_x(int n2, int threads) { const int nt = (n2 + threads - 1) / threads; return 8 * ((nt + 7) / 8); }
// launch grid x for a col_setup kernel with `threads` x 1 workgroups
__host__ inline int col_grid(const DevGrid &g, int threads) {
  return g.xcd_remap == 2 ? tile_grid_x(g.nxb, g.nyb, threads, 1) : col_grid_x(g.n2, threads);
}
__device__ __forceinline__ bool col_setup(const DevGrid &g, Col &c, bool interior_only) {
  c.b = blockIdx.y;
  if (g.xcd_remap == 2) {
    const TileId t = tile_of_block(g.nxb, g.nyb, blockDim.x, blockDim.y);
    c.i = t.ti * blockDim.x + threadIdx.x;
    c.j = t.tj * blockDim.y + threadIdx.y;
    if (!t.valid || c.i >= g.nxb || c.j >= g.nyb) return false;
    if (land_tile(g, c.b, t.ti * blockDim.x, blockDim.x, t.tj * blockDim.y, blockDim.y)) return false;
    c.p2 = c.j * g.nxb + c.i;
  } else {
    const int tile = g.xcd_remap ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    c.p2 = tile * blockDim.x + threadIdx.x;
    if (c.p2 >= g.n2) return false;
    if (land_run(g, c.b, (long long)tile * blockDim.x, blockDim.x)) return false;
    c.i = c.p2 % g.nxb;
    c.j = c.p2 / g.nxb;
  }
  if (interior_only && (c.i + 1 < g.ib || c.i + 1 > blk_ie(g, c.b) || c.j + 1 < g.jb || c.j + 1 > blk_je(g, c.b))) return false;
  c.q2 = (long long)c.b * g.n2 + c.p2;
  c.base3 = (long long)c.b * g.n3 + c.p2;
  return true;
}
// Element-wise 3-D kernels that read rows j +- 1 (del4 first Laplacians, k_kpp_vvc): cell of this thread.  tile = 0: 256 consecutive cells of the flattened (i,j) index; tile = R > 0: a 64 x R patch (64 R threads), whose
// rows j+-1 are mostly read by the same workgroup (R + 2 rows fetched for R written instead of 3 for 1)
__device__ __forceinline__ bool patch_cell(const DevGrid &g, int tile, int b, int &p2) {
  if (!tile) {
    p2 = blockIdx.x * blockDim.x + threadIdx.x;
    if (p2 >= g.n2) return false;
    return !land_run(g, b, (long long)blockIdx.x * blockDim.x, blockDim.x);
  }
  const int tiles_i = (g.nxb + 63) / 64;
  const int ti = blockIdx.x % tiles_i, tj = blockIdx.x / tiles_i;
  const int i = ti * 64 + (threadIdx.x & 63), j = tj * tile + (threadIdx.x >> 6);
  if (land_tile(g, b, ti * 64, 64, tj * tile, tile)) return false;
  if (i >= g.nxb || j >= g.nyb) return false;
  p2 = j * g.nxb + i;
  return true;
}
// bandwidth-bound grids: 64 x 4 patches (tx0.1v3: -0.5 ms per step); pop_tuning.del4_tile = 0 | 2 | 4 | 8 | 16 overrides
__host__ inline int patch_rows(const DevGrid &g, int tile_tuning) {
  if (tun_set(tile_tuning)) { const int r = tile_tuning; return (r == 2 || r == 4 || r == 8 || r == 16) ? r : 0; }
  return ((long long)g.n2 * g.nblocks > (1 << 19)) ? 4 : 0;
}
__host__ inline unsigned patch_grid_x(const DevGrid &g, int tile) {
  return tile ? (unsigned)(((g.nxb + 63) / 64) * ((g.nyb + tile - 1) / tile)) : (unsigned)((g.n2 + 255) / 256);
}
// 2-D reduction kernels (POP_RED_THREADS = 256 threads, one partial per workgroup): cell of this thread,
// or g.n2 (not a cell) for surplus threads.  Large grids use 64 x 4 tiles in the XCD-strided column order.
__host__ inline int red_grid_x(const DevGrid &g) {
  if (g.red_act) return g.red_nact;
  if (g.red_tiles) return tile_grid_x(g.nxb, g.nyb, 64, 4);
  const int nc = (g.n2 + 255) / 256;
  return g.red_band ? 8 * ((nc + 7) / 8) : nc;
}
// chunk (= partial slot) of this workgroup.  red_band: XCD x takes one contiguous band of chunks, so the rows
// j+-1 of a 9-point stencil were touched by the same XCD a few workgroups earlier; the chunk -> cells map and the
// order of the partials are unchanged, only which workgroup computes which chunk.
__device__ __forceinline__ int red_chunk(const DevGrid &g) {
  if (g.red_act) return g.red_act[(long long)blockIdx.y * g.red_nact + blockIdx.x];
  return g.red_band ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
}
__device__ __forceinline__ int red_cell(const DevGrid &g) {
  if (g.red_tiles) {
    const TileId t = tile_of_block(g.nxb, g.nyb, 64, 4);
    const int i = t.ti * 64 + (threadIdx.x & 63), j = t.tj * 4 + (threadIdx.x >> 6);
    return (t.valid && i < g.nxb && j < g.nyb) ? j * g.nxb + i : g.n2;
  }
  const long long p2 = (long long)red_chunk(g) * blockDim.x + threadIdx.x;
  return p2 < g.n2 ? (int)p2 : g.n2;
}

// 2-D reduction / fused solver kernels: does the chunk (or 64 x 4 tile) of this workgroup hold no ocean cell?  Every solver
// vector stays exactly 0 there (right-hand side, weights and the first guess are), so the workgroup only writes its zero
// partial(s).  Workgroup (0,0) never says so: it publishes the scalars of the iteration.
// deep: the tile must also lie at least NGHOST cells inside the physical domain (blocks spread over ranks: the cells
// nearer the edge are packed for / advanced on behalf of other ranks by the same kernels, which a skipped workgroup would not do)
__device__ __forceinline__ bool red_land(const DevGrid &g, bool deep = false) {
  if (!g.skip || (blockIdx.x == 0 && blockIdx.y == 0)) return false;
  if (g.red_act)   // compacted launch: entry w = (x % 8) * (red_nact / 8) + x / 8 of the sorted sequence, which land chunks only pad
    return (int)((blockIdx.x & 7) * (g.red_nact >> 3) + (blockIdx.x >> 3)) >= g.red_cnt[blockIdx.y];
  int i0, i1, j0, j1;   // inclusive, 0-based
  if (g.red_tiles) {
    const TileId t = tile_of_block(g.nxb, g.nyb, 64, 4);
    if (!t.valid) return false;
    i0 = t.ti * 64; i1 = i0 + 63; j0 = t.tj * 4; j1 = j0 + 3;
    if (!land_tile(g, blockIdx.y, i0, 64, j0, 4)) return false;
  } else {
    const long long p0 = (long long)red_chunk(g) * 256;
    if (!land_run(g, blockIdx.y, p0, 256)) return false;
    const long long p1 = (p0 + 255 < g.n2) ? p0 + 255 : g.n2 - 1;
    j0 = (int)(p0 / g.nxb); j1 = (int)(p1 / g.nxb);
    i0 = (j0 == j1) ? (int)(p0 % g.nxb) : 0; i1 = (j0 == j1) ? (int)(p1 % g.nxb) : g.nxb - 1;
  }
  if (deep && !(i0 >= g.ib - 1 + NGHOST && i1 <= blk_ie(g, blockIdx.y) - 1 - NGHOST && j0 >= g.jb - 1 + NGHOST && j1 <= blk_je(g, blockIdx.y) - 1 - NGHOST)) return false;
  return true;
}
template <int NF>
__device__ __forceinline__ bool red_land_out(const DevGrid &g, double *__restrict__ partial, int nchunk, bool deep = false) {
  if (!red_land(g, deep)) return false;
  if (threadIdx.x == 0) {
    const long long slot = (long long)blockIdx.y * nchunk + red_chunk(g);
#pragma unroll
    for (int f = 0; f < NF; ++f) partial[NF * slot + f] = 0.0;
  }
  return true;
}

// ---- McDougall, Wright, Jackett & Feistel (2003) equation of state as used by the
// reference (state_mod.F90:418-498, state_range_opt='enforce' :383-389, ranges :1054-1057).
// Coefficients are the published MWJF values scaled to g/cm^3 (numerator * 0.001).
struct MwjfP {   // pressure-dependent coefficients, wave-uniform per level
  double n0, n2, ns1t0, d0, d1, d3;
};
__device__ __forceinline__ MwjfP mwjf_level(double pbar) {
  const double p = 10.0 * pbar;
  MwjfP c;
  c.n0 = (9.99843699e+2 * 0.001) + p * ((1.04004591e-2 * 0.001) + p * (-3.24041825e-8 * 0.001));
  c.n2 = (-5.45928211e-2 * 0.001) + p * ((1.03970529e-7 * 0.001) + p * (-1.23869360e-11 * 0.001));
  c.ns1t0 = (2.96938239e+0 * 0.001) + p * (5.18761880e-6 * 0.001);
  c.d0 = 1.0e+0 + p * 5.30848875e-6;
  c.d1 = 7.28606739e-3 + (p * p * p) * -1.27934137e-17;
  c.d3 = 3.68390573e-7 + (p * p) * -3.03175128e-16;
  return c;
}
// the pressure-independent part of an evaluation (clamped T, S*1000 and its square root) and the rest;
// mwjf_eval(c, prep) == mwjf_rho<false>(c, T, S) bit for bit (same operations in the same order)
struct MwjfTS { double TQ, SQ, SQR; };
__device__ __forceinline__ MwjfTS mwjf_prep(double TK, double SK) {
  MwjfTS r;
  r.TQ = fmin(TK, 999.0); r.TQ = fmax(r.TQ, -2.0);
  double SQ = fmin(SK, 0.999); SQ = fmax(SQ, 0.0);
  r.SQ = 1000.0 * SQ;
  r.SQR = sqrt(r.SQ);
  return r;
}
__device__ __forceinline__ double mwjf_eval(const MwjfP &c, const MwjfTS &x) {
  const double n1 = 7.35212840e+0 * 0.001, n3 = 3.98476704e-4 * 0.001;
  const double ns1t1 = -7.23268813e-3 * 0.001, ns2t0 = 2.12382341e-3 * 0.001;
  const double d2 = -4.60835542e-5, d4 = 1.80809186e-10, ds1t0 = 2.14691708e-3, ds1t1 = -9.27062484e-6;
  const double ds1t3 = -1.78343643e-10, dsqt0 = 4.76534122e-6, dsqt2 = 1.63410736e-9;
  const double TQ = x.TQ, SQ = x.SQ, SQR = x.SQR;
  const double W1 = c.n0 + TQ * (n1 + TQ * (c.n2 + n3 * TQ)) + SQ * (c.ns1t0 + ns1t1 * TQ + ns2t0 * SQ);
  const double W2 = c.d0 + TQ * (c.d1 + TQ * (d2 + TQ * (c.d3 + d4 * TQ))) +
                    SQ * (ds1t0 + TQ * (ds1t1 + TQ * TQ * ds1t3) + SQR * (dsqt0 + TQ * TQ * dsqt2));
  const double DEN = 1.0 / W2;
  return W1 * DEN;
}
// Variant for many evaluations of the same (T,S) at different pressures (KPP buoydiff): the whole second
// term of the denominator, SQ*(ds1t0 + TQ*(ds1t1 + TQ*TQ*ds1t3) + SQR*(dsqt0 + TQ*TQ*dsqt2)), does not depend on
// pressure, nor does the square root inside it; it is formed once (same operations, same order) and reused.
struct MwjfTS2 { double TQ, SQ, A2; };
__device__ __forceinline__ MwjfTS2 mwjf_prep2(double TK, double SK) {
  const double ds1t0 = 2.14691708e-3, ds1t1 = -9.27062484e-6, ds1t3 = -1.78343643e-10, dsqt0 = 4.76534122e-6, dsqt2 = 1.63410736e-9;
  MwjfTS2 r;
  r.TQ = fmin(TK, 999.0); r.TQ = fmax(r.TQ, -2.0);
  double SQ = fmin(SK, 0.999); SQ = fmax(SQ, 0.0);
  r.SQ = 1000.0 * SQ;
  const double SQR = sqrt(r.SQ), TQ = r.TQ;
  r.A2 = r.SQ * (ds1t0 + TQ * (ds1t1 + TQ * TQ * ds1t3) + SQR * (dsqt0 + TQ * TQ * dsqt2));
  return r;
}
__device__ __forceinline__ double mwjf_eval2(const MwjfP &c, const MwjfTS2 &x) {
  const double n1 = 7.35212840e+0 * 0.001, n3 = 3.98476704e-4 * 0.001;
  const double ns1t1 = -7.23268813e-3 * 0.001, ns2t0 = 2.12382341e-3 * 0.001;
  const double d2 = -4.60835542e-5, d4 = 1.80809186e-10;
  const double TQ = x.TQ, SQ = x.SQ;
  const double W1 = c.n0 + TQ * (n1 + TQ * (c.n2 + n3 * TQ)) + SQ * (c.ns1t0 + ns1t1 * TQ + ns2t0 * SQ);
  const double W2 = c.d0 + TQ * (c.d1 + TQ * (d2 + TQ * (c.d3 + d4 * TQ))) + x.A2;
  const double DEN = 1.0 / W2;
  return W1 * DEN;
}
template <bool DERIV>
__device__ __forceinline__ double mwjf_rho(const MwjfP &c, double TK, double SK, double *drdt, double *drds) {
  const double n1 = 7.35212840e+0 * 0.001, n3 = 3.98476704e-4 * 0.001;
  const double ns1t1 = -7.23268813e-3 * 0.001, ns2t0 = 2.12382341e-3 * 0.001;
  const double d2 = -4.60835542e-5, d4 = 1.80809186e-10, ds1t0 = 2.14691708e-3, ds1t1 = -9.27062484e-6;
  const double ds1t3 = -1.78343643e-10, dsqt0 = 4.76534122e-6, dsqt2 = 1.63410736e-9;
  double TQ = fmin(TK, 999.0); TQ = fmax(TQ, -2.0);
  double SQ = fmin(SK, 0.999); SQ = fmax(SQ, 0.0);
  SQ = 1000.0 * SQ;
  const double SQR = sqrt(SQ);
  const double W1 = c.n0 + TQ * (n1 + TQ * (c.n2 + n3 * TQ)) + SQ * (c.ns1t0 + ns1t1 * TQ + ns2t0 * SQ);
  const double W2 = c.d0 + TQ * (c.d1 + TQ * (d2 + TQ * (c.d3 + d4 * TQ))) +
                    SQ * (ds1t0 + TQ * (ds1t1 + TQ * TQ * ds1t3) + SQR * (dsqt0 + TQ * TQ * dsqt2));
  const double DEN = 1.0 / W2;
  if (DERIV) {
    double W3 = n1 + TQ * (2.0 * c.n2 + 3.0 * n3 * TQ) + ns1t1 * SQ;
    double W4 = c.d1 + SQ * ds1t1 + TQ * (2.0 * (d2 + SQ * SQR * dsqt2) + TQ * (3.0 * (c.d3 + SQ * ds1t3) + TQ * 4.0 * d4));
    *drdt = (W3 - W1 * DEN * W4) * DEN;
    W3 = c.ns1t0 + ns1t1 * TQ + 2.0 * ns2t0 * SQ;
    W4 = ds1t0 + TQ * (ds1t1 + TQ * TQ * ds1t3) + 1.5 * SQR * (dsqt0 + TQ * TQ * dsqt2);
    *drds = (W3 - W1 * DEN * W4) * DEN * 1000.0;
  }
  return W1 * DEN;
}

}  // namespace pop
