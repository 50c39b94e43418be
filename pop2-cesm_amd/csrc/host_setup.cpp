// host_setup.cpp -- init-time host logic of libpop_amd: block decomposition, internal grid,
// masks, operator coefficients and the synthetic initial state.  Pure host code (no HIP), so
// it is testable without a GPU.  Every rank builds the 2-D fields of ALL blocks of the
// decomposition (they are functions of the global index only), which makes init-time halo
// updates and global sums local and identical on every rank.
//
// Reference behaviour restated (file:line under /root/reference/source unless noted):
//   blocks.F90:90-275, grid.F90:587-647,786-803,978-1043,1158-1159,1226-1297,1549-1709,
//   1957-2016,2555-2591,2908-2928,2984-3062, hmix_del2.F90:223-404,560-634,
//   advection.F90:387-396, POP_SolversMod.F90:771-822,895-906, barotropic.F90:150-205,
//   time_management.F90:753-858, initial.F90:964-1005,1389-1427, forcing_ws.F90:266-307.
#include "pop_internal.hpp"

namespace pop {

namespace {

struct Shift {   // zero-filled neighbour access inside one block (Fortran eoshift semantics)
  int nxb, nyb;
  size_t n2;
  double operator()(const std::vector<double> &A, int b, int i, int j) const {
    if (i < 0 || i >= nxb || j < 0 || j >= nyb) return 0.0;
    return A[b * n2 + (size_t)j * nxb + i];
  }
  int operator()(const std::vector<int> &A, int b, int i, int j) const {
    if (i < 0 || i >= nxb || j < 0 || j >= nyb) return 0;
    return A[b * n2 + (size_t)j * nxb + i];
  }
};

// topography_internal (grid.F90:1957-1985) + the stepped test extension (stepped_bathymetry); shared by the block distribution
// (work per block) and the grid set-up
static int kmt_rule(const pop_config &c, int km, const double latd, double lond, int ig, int jg) {
  if (lond < 0.0) lond = lond + 360.0;
  int k = km;
  if (latd > -35.0 && lond > 210.0 && lond < 250.0) k = 0;
  if (latd > 25.0 && lond > 210.0 && lond < 330.0) k = 0;
  if (latd > 60.0 && lond > 210.0 && lond < 150.0) k = 0;
  if (latd > -60.0 && lond > 110.0 && lond < 150.0) k = 0;
  if (std::fabs(latd) > 75.0) k = 0;
  // stepped_bathymetry = 1: stepped synthetic bathymetry (an extension for tests; the reference's internal topography is
  // flat, grid.F90:880-884): ocean columns of 3 ... km levels in stairs 3 cells wide in i and 2 in j, so that every
  // k > KMT / k > KMU branch and the shallow-column paths of the Thomas solves are exercised.  Integer arithmetic only.
  if (k > 0 && c.stepped_bathymetry == 1) k = std::max(3, km - ((ig / 3) * 5 + (jg / 2) * 3) % (km / 2 + 1));
  return k;
}
// global KMT at (ig, jg), 1-based: the caller's record, or the internal rule on the caller's / the internal ULAT, ULON
static int kmt_global(const HostModel &h, int ig, int jg) {
  const pop_config &c = h.c;
  const double radian = 180.0 / (4.0 * std::atan(1.0));
  const size_t p = (size_t)(jg - 1) * c.nx_global + (ig - 1);
  if (h.gin && h.gin->KMT) return h.gin->KMT[p];
  if (h.gin) return kmt_rule(c, c.km, h.gin->ULAT[p] * radian, h.gin->ULON[p] * radian, ig, jg);
  const double dlon = 360.0 / (double)c.nx_global, dlat = 180.0 / (double)c.ny_global;
  double x = ig * dlon; if (x > 180.0) x = x - 360.0;
  return kmt_rule(c, c.km, ((-90.0 + jg * dlat) / radian) * radian, (x / radian) * radian, ig, jg);
}

void make_blocks(HostModel &h) {
  const pop_config &c = h.c;
  h.nbx = (c.nx_global - 1) / c.block_size_x + 1;
  h.nby = (c.ny_global - 1) / c.block_size_y + 1;
  h.nblocks_tot = h.nbx * h.nby;
  h.nxb = c.block_size_x + 2 * NGHOST;
  h.nyb = c.block_size_y + 2 * NGHOST;
  h.n2 = (size_t)h.nxb * h.nyb;
  h.n3 = h.n2 * h.km;
  h.all_blocks.resize(h.nblocks_tot);
  int id = 0;
  for (int jbk = 0; jbk < h.nby; ++jbk)
    for (int ibk = 0; ibk < h.nbx; ++ibk, ++id) {
      BlockInfo &B = h.all_blocks[id];
      B.block_id = id + 1; B.local_id = 0; B.iblock = ibk + 1; B.jblock = jbk + 1;
      B.ib = NGHOST + 1; B.jb = NGHOST + 1; B.ie = h.nxb - NGHOST; B.je = h.nyb - NGHOST;
      B.i_glob.resize(h.nxb); B.j_glob.resize(h.nyb);
      auto wrap = [](int g, int nglob, int cyclic, int pos1, int lo, int &last) {
        // g: tentative global index of local position pos1 (1-based); returns the stored index;
        // cyclic == 2: tripole north boundary, the far-side ghost keeps its row as a negative index (blocks.F90:205-208)
        if (g < 1) g = (cyclic == 1) ? g + nglob : 0;
        if (g > nglob + NGHOST) g = 0;                       // padding
        else if (g > nglob) g = (cyclic == 1) ? g - nglob : (cyclic == 2) ? -g : 0;      // far-side ghost
        else if (g == nglob && pos1 > lo) last = pos1;       // last physical point (padded domain)
        return g;
      };
      const int js = jbk * c.block_size_y + 1, is = ibk * c.block_size_x + 1;
      for (int j = 1; j <= h.nyb; ++j) B.j_glob[j - 1] = wrap(js - NGHOST + j - 1, c.ny_global, c.ns_boundary, j, B.jb, B.je);
      for (int i = 1; i <= h.nxb; ++i) B.i_glob[i - 1] = wrap(is - NGHOST + i - 1, c.nx_global, c.ew_boundary == 1 ? 1 : 0, i, B.ib, B.ie);
    }
  // distribution: contiguous runs of block ids per rank (cartesian in j for nbx == 1);
  // clinic and tropic distributions coincide, so POP_RedistributeBlocks is the identity
  h.block_owner.assign(h.nblocks_tot, 0);
  h.block_local.assign(h.nblocks_tot, 0);
  h.local_ids.clear();
  for (int n = 0; n < h.nblocks_tot; ++n) {
    int owner = (int)(((long long)n * h.nranks) / h.nblocks_tot);
    h.block_owner[n] = owner;
  }
  if (c.distribution_type == 1 && h.nranks > 1 && h.nblocks_tot > h.nranks) {
    // load-balanced distribution (the reference balances ocean points per task with its 'rake' / 'spacecurve'
    // distributions, distribution.F90; the global KMT is read first for exactly this, grid.F90:449-456): still contiguous
    // runs of block ids -- neighbours stay neighbours -- but the cuts equalise the ocean columns per rank instead of the
    // block counts.  With land elimination a rank's time follows its ocean columns, not its blocks.
    std::vector<long long> work(h.nblocks_tot, 0), pre(h.nblocks_tot + 1, 0);
    for (int n = 0; n < h.nblocks_tot; ++n) {
      const BlockInfo &B = h.all_blocks[n];
      for (int j = B.jb; j <= B.je; ++j) for (int i = B.ib; i <= B.ie; ++i) {
        const int ig = B.i_glob[i - 1], jg = B.j_glob[j - 1];
        if (ig > 0 && jg > 0 && kmt_global(h, ig, jg) > 0) ++work[n];
      }
      work[n] += 1;                       // a land block still costs its launches
      pre[n + 1] = pre[n] + work[n];
    }
    int first = 0;
    for (int r = 0; r < h.nranks; ++r) {
      int last;                           // rank r owns blocks first .. last
      if (r == h.nranks - 1) last = h.nblocks_tot - 1;
      else {
        const double target = (double)pre[h.nblocks_tot] * (r + 1) / h.nranks;
        last = first;
        while (last + 1 < h.nblocks_tot - (h.nranks - 1 - r) &&
               std::fabs((double)pre[last + 2] - target) <= std::fabs((double)pre[last + 1] - target)) ++last;
      }
      for (int n = first; n <= last; ++n) h.block_owner[n] = r;
      first = last + 1;
    }
  }
  std::vector<int> cnt(h.nranks, 0);
  for (int n = 0; n < h.nblocks_tot; ++n) {
    h.block_local[n] = cnt[h.block_owner[n]]++;
    if (h.block_owner[n] == h.rank) { h.local_ids.push_back(n + 1); h.all_blocks[n].local_id = h.block_local[n] + 1; }
  }
  h.nblocks = (int)h.local_ids.size();
  if (h.plan_only) {   // ocean columns per rank without any field: what a decomposition check wants to know about the balance
    h.ocean_cols_local = 0; h.ocean_cols_total = 0;
    for (int n = 0; n < h.nblocks_tot; ++n) {
      const BlockInfo &B = h.all_blocks[n];
      long long w = 0;
      for (int j = B.jb; j <= B.je; ++j) for (int i = B.ib; i <= B.ie; ++i) {
        const int ig = B.i_glob[i - 1], jg = B.j_glob[j - 1];
        if (ig > 0 && jg > 0 && kmt_global(h, ig, jg) > 0) ++w;
      }
      h.ocean_cols_total += w;
      if (h.block_owner[n] == h.rank) h.ocean_cols_local += w;
    }
  }
}

int make_vertical(HostModel &h) {
  const int km = h.km;
  for (auto *v : {&h.dz, &h.dzw, &h.zt, &h.zw, &h.c2dz, &h.dzr, &h.dz2r, &h.dzwr, &h.pressz, &h.bouss, &h.dt, &h.afac_t, &h.afac_u})
    v->assign(km + 3, 0.0);
  const double zmax = 5500.0, dz_sfc = 25.0, dz_deep = 400.0, eps = 1.0e-10;
  auto profile = [&](double zlength) {   // compute_dz: returns integrated depth, fills dz (m)
    double depth = 0.0;
    for (int k = 1; k <= km; ++k) {
      const double r = depth / zlength;
      h.dz[k] = dz_deep - (dz_deep - dz_sfc) * std::exp(-(r * r));
      depth = depth + h.dz[k];
    }
    return depth;
  };
  double zl0 = eps, zl1 = zmax, dzl = zl1 - zl0;
  double d0 = profile(zl0), d1 = profile(zl1), depth = 0.0;
  if ((d0 - zmax) * (d1 - zmax) > 0.0) { h.err = "vert_grid: km levels cannot span zmax = 5500 m"; return 1; }
  while (dzl / zmax > eps) {
    const double zl = zl0 + 0.5 * dzl;
    depth = profile(zl);
    if ((d0 - zmax) * (depth - zmax) < 0.0) { d1 = depth; zl1 = zl; }
    else if ((d1 - zmax) * (depth - zmax) < 0.0) { d0 = depth; zl0 = zl; }
    else { h.err = "vert_grid: zero point not in interval"; return 1; }
    dzl = zl1 - zl0;
  }
  for (int k = 1; k <= km; ++k) h.dz[k] = h.dz[k] * 100.0;
  h.dzw[0] = 0.5 * h.dz[1]; h.dzw[km] = 0.5 * h.dz[km]; h.dzwr[0] = 1.0 / h.dzw[0];
  h.zw[1] = h.dz[1]; h.zt[1] = h.dzw[0];
  for (int k = 1; k < km; ++k) {
    h.dzw[k] = 0.5 * (h.dz[k] + h.dz[k + 1]);
    h.zw[k + 1] = h.zw[k] + h.dz[k + 1];
    h.zt[k + 1] = h.zt[k] + h.dzw[k];
  }
  for (int k = 1; k <= km; ++k) {
    h.c2dz[k] = 2.0 * h.dz[k]; h.dzr[k] = 1.0 / h.dz[k]; h.dz2r[k] = 1.0 / h.c2dz[k]; h.dzwr[k] = 1.0 / h.dzw[k];
    const double d = h.zt[k] * 0.01;
    h.pressz[k] = 0.059808 * (std::exp(-0.025 * d) - 1.0) + 0.100766 * d + 2.28405e-7 * (d * d);
    h.bouss[k] = h.c.lbouss_correct ? 1.0 / (1.02819 + 4.4004e-5 * h.pressz[k] - 2.93161e-4 * std::exp(-0.05 * h.pressz[k])) : 1.0;
    h.afac_t[k] = h.c.aidif * h.dzwr[k]; h.afac_u[k] = h.c.aidif * h.dzwr[k];
  }
  return 0;
}

}  // namespace

// ---- single-process halo over ALL blocks (init-time fields) -------------------------------
template <class T>
static void halo_all(const HostModel &h, T *a, int nz, T fill) {
  const pop_config &c = h.c;
  for (int b = 0; b < h.nblocks_tot; ++b) {
    const BlockInfo &B = h.all_blocks[b];
    for (int j = 1; j <= h.nyb; ++j)
      for (int i = 1; i <= h.nxb; ++i) {
        if (i >= B.ib && i <= B.ie && j >= B.jb && j <= B.je) continue;
        const int gi = B.i_glob[i - 1], gj = B.j_glob[j - 1];
        for (int k = 0; k < nz; ++k) {
          T v = fill;
          if (gi > 0 && gj > 0) {
            const int sbx = (gi - 1) / c.block_size_x, sby = (gj - 1) / c.block_size_y;
            const int sb = sby * h.nbx + sbx;
            const int si = gi - sbx * c.block_size_x + NGHOST, sj = gj - sby * c.block_size_y + NGHOST;
            v = a[((size_t)sb * nz + k) * h.n2 + (size_t)(sj - 1) * h.nxb + (si - 1)];
          }
          a[((size_t)b * nz + k) * h.n2 + (size_t)(j - 1) * h.nxb + (i - 1)] = v;
        }
      }
  }
}
void host_halo_r8(const HostModel &h, double *a, int nz, double fill) { halo_all<double>(h, a, nz, fill); }
void host_halo_i4(const HostModel &h, int *a, int nz, int fill) { halo_all<int>(h, a, nz, fill); }

// tripole pass (HaloPlan::tripole): two phases, because the symmetrised top row reads physical cells it also writes
template <class T, class Avg>
static void tripole_all(const HostModel &h, T *a, int nz, int loc, int kind, Avg avg) {
  const TripolePlan &P = h.halo.tripole_g[loc];   // host arrays hold every block
  const T isign = (kind == 0) ? (T)1 : (T)-1;
  const size_t n2 = h.n2;
  std::vector<T> tmp(P.dst.size());
  auto at = [&](int cell, int k) -> T & { return a[((size_t)(cell / (int)n2) * nz + k) * n2 + cell % (int)n2]; };
  for (int k = 0; k < nz; ++k) {
    for (size_t e = 0; e < P.dst.size(); ++e) {
      const T x = at(P.a[e], k);
      if (P.b[e] < 0) tmp[e] = isign * x;
      else { const T y = at(P.b[e], k); const T m = avg(x < 0 ? -x : x, y < 0 ? -y : y); tmp[e] = (x < 0) ? -m : m; }
    }
    for (size_t e = 0; e < P.dst.size(); ++e) at(P.dst[e], k) = tmp[e];
  }
}
void host_halo_r8_loc(const HostModel &h, double *a, int nz, double fill, int loc, int kind) {
  halo_all<double>(h, a, nz, fill);
  if (h.c.ns_boundary == 2) tripole_all<double>(h, a, nz, loc, kind, [](double x, double y) { return 0.5 * (x + y); });
}
void host_halo_i4_loc(const HostModel &h, int *a, int nz, int fill, int loc, int kind) {
  halo_all<int>(h, a, nz, fill);
  if (h.c.ns_boundary == 2) tripole_all<int>(h, a, nz, loc, kind, [](int x, int y) { return (int)std::lround(0.5 * ((double)x + (double)y)); });
}

// sum over the physical domain, j outer / i inner per block, block sums added in block-id
// order (serial/global_reductions.F90:237-262; b4b form mpi/POP_ReductionsMod.F90:348-383)
double host_global_sum(const HostModel &h, const double *a, const double *mask) {
  double g = 0.0;
  for (int b = 0; b < h.nblocks_tot; ++b) {
    const BlockInfo &B = h.all_blocks[b];
    double s = 0.0;
    for (int j = B.jb; j <= B.je; ++j)
      for (int i = B.ib; i <= B.ie; ++i) {
        const size_t p = b * h.n2 + (size_t)(j - 1) * h.nxb + (i - 1);
        s = mask ? s + a[p] * mask[p] : s + a[p];
      }
    g = g + s;
  }
  return g;
}

// tripole grids: fields on north faces / NE corners hold the top row twice; the points with i_glob > nx/2 are
// subtracted again from their block sum (mpi/POP_ReductionsMod.F90:308-341)
double host_global_sum_loc(const HostModel &h, const double *a, const double *mask, int loc) {
  if (h.c.ns_boundary != 2 || (loc != 1 && loc != 2)) return host_global_sum(h, a, mask);
  double g = 0.0;
  for (int b = 0; b < h.nblocks_tot; ++b) {
    const BlockInfo &B = h.all_blocks[b];
    double s = 0.0;
    for (int j = B.jb; j <= B.je; ++j)
      for (int i = B.ib; i <= B.ie; ++i) {
        const size_t p = b * h.n2 + (size_t)(j - 1) * h.nxb + (i - 1);
        s = mask ? s + a[p] * mask[p] : s + a[p];
      }
    if (B.j_glob[B.je] < 0)
      for (int i = B.ib; i <= B.ie; ++i)
        if (B.i_glob[i - 1] > h.c.nx_global / 2) {
          const size_t p = b * h.n2 + (size_t)(B.je - 1) * h.nxb + (i - 1);
          s = mask ? s - a[p] * mask[p] : s - a[p];
        }
    g = g + s;
  }
  return g;
}

int host_build(HostModel &h) {
  const pop_config &c = h.c;
  h.km = c.km; h.nt = c.nt;
  if (c.nt < 2 || c.nt > MAXNT) { h.err = "nt must be in [2,8]"; return 1; }
  if (c.km < 2) { h.err = "km must be >= 2"; return 1; }
  // block sizes that do not divide the domain: the last column / row of blocks is padded (blocks.F90:174-265; make_blocks)
  // (round 4: also with the tripole fold, P-CSI and the EVP preconditioner -- the fold and the halo plan work on global indices, the
  // Lanczos bounds and the sub-block tables on every block's own ib .. ie / jb .. je, POP_SolversMod.F90:2483-2488)
  if (c.ns_boundary < 0 || c.ns_boundary > 2) { h.err = "ns_boundary: 0 closed, 1 cyclic, 2 tripole"; return 1; }
  if (c.ns_boundary == 2 && (c.ew_boundary != 1 || c.nx_global % 2 || c.block_size_y < NGHOST + 1)) {
    h.err = "tripole needs a cyclic east-west boundary, even nx_global and blocks of at least nghost+1 rows"; return 1;
  }
  make_blocks(h);
  if (h.nblocks == 0) { h.err = "rank owns no blocks (more ranks than blocks)"; return 1; }
  build_halo_plan(h);   // needs the blocks only; the tripole pass of the grid-time halo updates below reads it
  if (h.halo.tripole_split) { h.err = "tripole: the top row of blocks must belong to one rank (use full-width blocks, block_size_x = nx_global, or fewer ranks)"; return 1; }
  if (h.plan_only) return 0;   // POP_CREATE_PLAN_ONLY: the block table, the distribution and the halo plan are all a decomposition check reads
  if (make_vertical(h)) return 1;

  const int nxb = h.nxb, nyb = h.nyb, NB = h.nblocks_tot;
  const size_t n2 = h.n2, A2 = n2 * NB;
  const Shift S{nxb, nyb, n2};
  auto idx = [&](int b, int i, int j) { return b * n2 + (size_t)j * nxb + i; };   // 0-based i,j
  auto newf = [&](const char *n) -> std::vector<double> & { auto &v = h.f2[n]; v.assign(A2, 0.0); return v; };
  auto newi = [&](const char *n) -> std::vector<int> & { auto &v = h.i2[n]; v.assign(A2, 0); return v; };

  // ---------------- horizontal grid (uniform lat-lon) ----------------
  const double pi = 4.0 * std::atan(1.0), radian = 180.0 / pi;
  const int nxg = c.nx_global, nyg = c.ny_global;
  const double dlon = 360.0 / (double)nxg, dlat = 180.0 / (double)nyg;
  auto ulat_g = [&](int jg) { return (-90.0 + jg * dlat) / radian; };          // jg 1-based
  auto ulon_g = [&](int ig) { double x = ig * dlon; if (x > 180.0) x = x - 360.0; return x / radian; };
  auto kmt_ll = [&](const double latd, double lond, int ig, int jg) { return kmt_rule(c, h.km, latd, lond, ig, jg); };
  auto kmt_g = [&](int ig, int jg) { return kmt_ll(ulat_g(jg) * radian, ulon_g(ig) * radian, ig, jg); };
  auto &ULAT = newf("ULAT"), &ULON = newf("ULON"), &TLAT = newf("TLAT");
  auto &HTN = newf("HTN"), &HTE = newf("HTE"), &HUS = newf("HUS"), &HUW = newf("HUW");
  auto &DXU = newf("DXU"), &DYU = newf("DYU"), &DXT = newf("DXT"), &DYT = newf("DYT");
  auto &KMT = newi("KMT"), &KMU = newi("KMU");
  const double cell = dlat * RADIUS / radian, cellx = dlon * RADIUS / radian;
  const pop_grid_input *gin = h.gin;
  if (gin) {
    // horiz_grid_opt = 'file' (grid.F90:1314-1542 read_horiz_grid) and topography_opt = 'file' (:2025-2107) on the
    // caller's global arrays.  scatter_global (mpi/gather_scatter.F90:862-1161): every local cell, ghosts included,
    // reads its global address; zero global index -> 0; ghost rows beyond a tripole boundary read the address mirrored
    // with the offsets of the field location (centre 1,1; NE corner 0,0; E face 0,1; N face 1,0), no sign factor.
    auto scatter = [&](auto &A, auto G, int loc) {
      const int xo = (loc == 0 || loc == 2) ? 1 : 0, yo = (loc == 0 || loc == 3) ? 1 : 0;
      for (int b = 0; b < NB; ++b) {
        const BlockInfo &B = h.all_blocks[b];
        for (int j = 0; j < nyb; ++j) for (int i = 0; i < nxb; ++i) {
          const int ig = B.i_glob[i], jg = B.j_glob[j];
          auto &dst = A[idx(b, i, j)];
          dst = 0;
          if (ig == 0 || jg == 0) continue;
          if (jg > 0) { dst = G(ig, jg); continue; }
          const int js = nyg + yo + (jg + nyg);
          int is = nxg + xo - ig;
          if (is < 1) is += nxg;
          if (is > nxg) is -= nxg;
          dst = G(is, js);
        }
      }
    };
    auto rec = [&](const double *R) { return [R, nxg](int i, int j) { return R[(size_t)(j - 1) * nxg + (i - 1)]; }; };
    const auto gHTN = rec(gin->HTN), gHTE = rec(gin->HTE);
    scatter(ULAT, rec(gin->ULAT), 1); scatter(ULON, rec(gin->ULON), 1);
    scatter(HTN, gHTN, 2);
    scatter(DXU, [&](int i, int j) { return 0.5 * (gHTN(i, j) + gHTN(i == nxg ? 1 : i + 1, j)); }, 1);
    scatter(DXT, [&](int i, int j) { return 0.5 * (gHTN(i, j) + gHTN(i, j == 1 ? nyg : j - 1)); }, 0);
    scatter(HTE, gHTE, 3);
    scatter(DYT, [&](int i, int j) { return 0.5 * (gHTE(i, j) + gHTE(i == 1 ? nxg : i - 1, j)); }, 0);
    scatter(DYU, [&](int i, int j) {
      if (c.ns_boundary == 2 && j == nyg) return gHTE(i, j);       // tripole-grid correction :1495-1500
      return 0.5 * (gHTE(i, j) + gHTE(i, j == nyg ? 1 : j + 1)); }, 1);
    scatter(HUS, rec(gin->HUS), 3); scatter(HUW, rec(gin->HUW), 2);
    for (auto *F : {&HTN, &HTE, &HUS, &HUW, &DXU, &DYU, &DXT, &DYT}) for (double &v : *F) if (v <= 0.0) v = 1.0;
    if (gin->KMT) { const int *K = gin->KMT; scatter(KMT, [K, nxg](int i, int j) { return K[(size_t)(j - 1) * nxg + (i - 1)]; }, 0); }
    else {   // topography_internal on the supplied ULAT / ULON
      const auto gLAT = rec(gin->ULAT), gLON = rec(gin->ULON);
      scatter(KMT, [&](int i, int j) { return kmt_ll(gLAT(i, j) * radian, gLON(i, j) * radian, i, j); }, 0);
    }
  }
  for (int b = 0; b < NB; ++b) {
    const BlockInfo &B = h.all_blocks[b];
    for (int j = 0; j < nyb && !gin; ++j) {
      const int jg = B.j_glob[j];
      int jm1 = jg - 1; if (jm1 < 1) jm1 = nyg;
      for (int i = 0; i < nxb; ++i) {
        const size_t p = idx(b, i, j);
        const int ig = B.i_glob[i];
        HTN[p] = cellx; HTE[p] = cell; HUS[p] = cellx; HUW[p] = cell; DYT[p] = cell; DYU[p] = cell;
        if (ig > 0 && jg > 0) {
          ULON[p] = ulon_g(ig); ULAT[p] = ulat_g(jg);
          HTN[p] = HTN[p] * std::cos(ULAT[p]);
          DXU[p] = HTN[p];
          const double lathalf = (-90.0 + (jg - 0.5) * dlat) / radian;
          HUS[p] = HUS[p] * std::cos(lathalf);
          DXT[p] = dlon * RADIUS / radian * 0.5 * (std::cos(ulat_g(jg)) + std::cos(ulat_g(jm1)));
          KMT[p] = kmt_g(ig, jg);
        } else {
          ULON[p] = 0.0; ULAT[p] = 0.0; HTN[p] = 1.0; HUS[p] = 1.0; DXU[p] = 1.0; KMT[p] = 0;
        }
      }
    }
    // closed boundaries: extend the physical-edge metrics into the ghost cells
    auto extend = [&](int i, int j, int si, int sj) {
      for (auto *F : {&DXU, &DYU, &DXT, &DYT}) (*F)[idx(b, i, j)] = (*F)[idx(b, si, sj)];
    };
    if (B.i_glob[0] == 0) for (int j = 0; j < nyb; ++j) for (int i = 0; i < B.ib - 1; ++i) extend(i, j, B.ib - 1, j);
    if (B.i_glob[B.ie] == 0) for (int j = 0; j < nyb; ++j) for (int i = B.ie; i < nxb; ++i) extend(i, j, B.ie - 1, j);
    if (B.j_glob[0] == 0) for (int j = 0; j < B.jb - 1; ++j) for (int i = 0; i < nxb; ++i) extend(i, j, i, B.jb - 1);
    if (B.j_glob[B.je] == 0) for (int j = B.je; j < nyb; ++j) for (int i = 0; i < nxb; ++i) extend(i, j, i, B.je - 1);
    // padded blocks: the cells beyond the ghost cells of a short block (global index 0) take the metrics of their neighbour towards
    // the block, so that reciprocals stay finite; they are land (KMT = 0) and nothing reads them
    for (int i = B.ie; i < nxb; ++i) if (B.i_glob[i] == 0 && B.i_glob[B.ie] != 0) for (int j = 0; j < nyb; ++j) extend(i, j, i - 1, j);
    for (int j = B.je; j < nyb; ++j) if (B.j_glob[j] == 0 && B.j_glob[B.je] != 0) for (int i = 0; i < nxb; ++i) extend(i, j, i, j - 1);
  }
  auto &DXUR = newf("DXUR"), &DYUR = newf("DYUR"), &DXTR = newf("DXTR"), &DYTR = newf("DYTR");
  auto &UAREA = newf("UAREA"), &TAREA = newf("TAREA"), &UAREA_R = newf("UAREA_R"), &TAREA_R = newf("TAREA_R");
  for (size_t p = 0; p < A2; ++p) {
    DXUR[p] = 1.0 / DXU[p]; DYUR[p] = 1.0 / DYU[p];
    UAREA[p] = DXU[p] * DYU[p]; UAREA_R[p] = 1.0 / UAREA[p];
    DXTR[p] = 1.0 / DXT[p]; DYTR[p] = 1.0 / DYT[p];
    TAREA[p] = DXT[p] * DYT[p]; TAREA_R[p] = 1.0 / TAREA[p];
  }
  auto &AU0 = newf("AU0"), &AUN = newf("AUN"), &AUE = newf("AUE"), &AUNE = newf("AUNE");
  for (int b = 0; b < NB; ++b)
    for (int j = 0; j < nyb; ++j) for (int i = 0; i < nxb; ++i) {
      const size_t p = idx(b, i, j);
      AU0[p] = TAREA[p] * 0.25 * UAREA_R[p];
      AUN[p] = S(TAREA, b, i, j + 1) * 0.25 * UAREA_R[p];
      AUE[p] = S(TAREA, b, i + 1, j) * 0.25 * UAREA_R[p];
      AUNE[p] = S(TAREA, b, i + 1, j + 1) * 0.25 * UAREA_R[p];
    }
  // T-point latitude by Cartesian averaging of the 4 surrounding U points
  for (int b = 0; b < NB; ++b) {
    const BlockInfo &B = h.all_blocks[b];
    for (int j = 1; j < nyb; ++j) for (int i = 1; i < nxb; ++i) {
      double x[4], y[4], z[4];
      const int di[4] = {0, 0, -1, -1}, dj[4] = {0, -1, 0, -1};   // c, s, w, sw
      for (int q = 0; q < 4; ++q) {
        const size_t p = idx(b, i + di[q], j + dj[q]);
        const double cz = std::cos(ULAT[p]);
        x[q] = std::cos(ULON[p]) * cz; y[q] = std::sin(ULON[p]) * cz; z[q] = std::sin(ULAT[p]);
      }
      const double tx = 0.25 * (x[0] + x[1] + x[2] + x[3]), ty = 0.25 * (y[0] + y[1] + y[2] + y[3]);
      const double tz = 0.25 * (z[0] + z[1] + z[2] + z[3]);
      const double da = std::sqrt(tx * tx + ty * ty + tz * tz);
      TLAT[idx(b, i, j)] = std::asin(tz / da);
    }
    if (B.j_glob[B.jb - 1] == 1)
      for (int i = B.ib - 1; i < B.ie; ++i) TLAT[idx(b, i, B.jb - 1)] = 2.0 * TLAT[idx(b, i, B.jb)] - TLAT[idx(b, i, B.jb + 1)];
  }
  host_halo_r8_loc(h, TLAT.data(), 1, 0.0, 0, 0);                  // centre, scalar (grid.F90:3073)

  // ---------------- masks and depths ----------------
  for (int b = 0; b < NB; ++b)
    for (int j = 0; j < nyb - 1; ++j) for (int i = 0; i < nxb - 1; ++i) {
      int m = KMT[idx(b, i, j)];
      m = std::min(m, KMT[idx(b, i + 1, j)]); m = std::min(m, KMT[idx(b, i, j + 1)]); m = std::min(m, KMT[idx(b, i + 1, j + 1)]);
      KMU[idx(b, i, j)] = m;
    }
  host_halo_i4_loc(h, KMU.data(), 1, 0, 1, 0);                     // NE corner, scalar (grid.F90:987)
  // ---------------- partial bottom cells (grid.F90:916-1020) ----------------
  // DZT(i,j,k) = DZBC(i,j) where k = KMT(i,j), dz(k) elsewhere; DZU = min of the four surrounding DZT, halo-updated (NE corner,
  // fill 0), dz(k) below the bottom.  Below KMU = min(KMT) every surrounding DZT is dz(k), so DZU differs from dz(k) at level KMU
  // only: the device keeps the two 2-D fields DZBC and DZUB = DZU(:,:,KMU) and forms DZT / DZU where it needs them
  // (levels 0 and km+1 are 0 as in the reference's arrays).
  if (c.partial_bottom_cells) {
    auto &DZBC = newf("DZBC"), &DZUB = newf("DZUB");
    for (int b = 0; b < NB; ++b) {
      const BlockInfo &B = h.all_blocks[b];
      for (int j = 0; j < nyb; ++j) for (int i = 0; i < nxb; ++i) {
        // read_bottom_cell :2116-2186: scatter_global as a centre scalar (ghosts beyond a tripole fold read the mirrored address)
        const int ig = B.i_glob[i], jg = B.j_glob[j];
        int is = ig, js = jg;
        if (ig == 0 || jg == 0) continue;
        if (jg < 0) { js = nyg + 1 + (jg + nyg); is = nxg + 1 - ig; if (is < 1) is += nxg; if (is > nxg) is -= nxg; }
        const size_t p = idx(b, i, j);
        if (gin && gin->DZBC) DZBC[p] = gin->DZBC[(size_t)(js - 1) * nxg + (is - 1)];
        else if (KMT[p] > 0)   // no record: synthetic thickness in (0.25, 1] dz(KMT) (TEST EXTENSION; the oracle uses the same integer rule)
          DZBC[p] = (0.25 + 0.75 * (double)((is * 7 + js * 13) % 16 + 1) / 16.0) * h.dz[KMT[p]];
      }
    }
    for (int b = 0; b < NB; ++b)
      for (int j = 0; j < nyb - 1; ++j) for (int i = 0; i < nxb - 1; ++i) {
        const size_t q[4] = {idx(b, i, j), idx(b, i + 1, j), idx(b, i, j + 1), idx(b, i + 1, j + 1)};
        int kmu = KMT[q[0]];
        for (int t = 1; t < 4; ++t) kmu = std::min(kmu, KMT[q[t]]);
        if (kmu < 1) continue;
        double v = (KMT[q[0]] == kmu) ? DZBC[q[0]] : h.dz[kmu];
        for (int t = 1; t < 4; ++t) v = std::min(v, (KMT[q[t]] == kmu) ? DZBC[q[t]] : h.dz[kmu]);
        DZUB[q[0]] = v;
      }
    host_halo_r8_loc(h, DZUB.data(), 1, 0.0, 1, 0);                 // NE corner, scalar, fillValue 0 (:958-960)
    for (size_t p = 0; p < A2; ++p) if (KMU[p] < 1) DZUB[p] = 0.0;
  }
  auto &HT = newf("HT"), &HU = newf("HU"), &HUR = newf("HUR"), &RCALCT = newf("RCALCT"), &RCALCU = newf("RCALCU");
  auto &FCOR = newf("FCOR"), &FCORT = newf("FCORT");
  const std::vector<double> *pDZBC = c.partial_bottom_cells ? &h.f2["DZBC"] : nullptr, *pDZUB = c.partial_bottom_cells ? &h.f2["DZUB"] : nullptr;
  auto &KMTN = newi("KMTN"), &KMTS = newi("KMTS"), &KMTE = newi("KMTE"), &KMTW = newi("KMTW"), &KMTEE = newi("KMTEE"), &KMTNN = newi("KMTNN");
  for (int b = 0; b < NB; ++b)
    for (int j = 0; j < nyb; ++j) for (int i = 0; i < nxb; ++i) {
      const size_t p = idx(b, i, j);
      if (c.partial_bottom_cells) {   // :1001-1020
        if (KMT[p] >= 1) HT[p] = h.zw[KMT[p] - 1] + (*pDZBC)[p];
        if (KMU[p] >= 1) { HU[p] = h.zw[KMU[p] - 1] + (*pDZUB)[p]; HUR[p] = 1.0 / HU[p]; }
      } else {
      if (KMT[p] >= 1) HT[p] = h.zw[KMT[p]];
      if (KMU[p] >= 1) { HU[p] = h.zw[KMU[p]]; HUR[p] = 1.0 / h.zw[KMU[p]]; }
      }
      RCALCT[p] = KMT[p] >= 1 ? 1.0 : 0.0;
      RCALCU[p] = KMU[p] >= 1 ? 1.0 : 0.0;
      KMTN[p] = S(KMT, b, i, j + 1); KMTS[p] = S(KMT, b, i, j - 1);
      KMTE[p] = S(KMT, b, i + 1, j); KMTW[p] = S(KMT, b, i - 1, j);
      KMTEE[p] = S(KMT, b, i + 2, j); KMTNN[p] = S(KMT, b, i, j + 2);
      FCOR[p] = 2.0 * OMEGA * std::sin(ULAT[p]);
      FCORT[p] = 2.0 * OMEGA * std::sin(TLAT[p]);
    }
  {   // uarea_equator: UAREA at the ocean U point(s) of smallest |ULAT| (min taken over ties)
    double wmin = 1.0e300, amin = 1.0e300;
    for (int pass = 0; pass < 2; ++pass)
      for (int b = 0; b < NB; ++b) {
        const BlockInfo &B = h.all_blocks[b];
        for (int j = B.jb - 1; j < B.je; ++j) for (int i = B.ib - 1; i < B.ie; ++i) {
          const size_t p = idx(b, i, j);
          if (KMU[p] < 1) continue;
          const double w = std::fabs(ULAT[p]);
          if (pass == 0) { if (w < wmin) wmin = w; }
          else { const double v = (w == wmin) ? UAREA[p] : 1.e+20; if (v < amin) amin = v; }
        }
      }
    h.uarea_equator = amin;
  }

  if (c.hmix_tracer == 3) {   // Gent-McWilliams (init_meso_mixing, hmix_gm_submeso_share.F90:131-137; init_gm, hmix_gm.F90:889-894)
    auto &GHYX = newf("gmHYX"), &GHXY = newf("gmHXY"), &GRBR = newf("gmRBR");
    for (size_t p = 0; p < A2; ++p) {
      GHYX[p] = HTE[p] / HUS[p];
      GHXY[p] = HTN[p] / HUW[p];
      double r = std::fabs(FCORT[p]) / 200.0;          // |f| / Cg, Cg = 200 cm/s
      r = std::min(r, 1.0 / 1.5e+6);                   // Rossby radius >= 15 km
      r = std::max(r, 1.e-7);                          //               <= 100 km
      GRBR[p] = r;
    }
  }
  // ---------------- del2 operator weights, metric advection coefficients ----------------
  auto &AMF = newf("AMF"), &AHF = newf("AHF");
  for (size_t p = 0; p < A2; ++p) { AMF[p] = 1.0; AHF[p] = 1.0; }
  if (c.lvariable_hmix && (c.hmix_momentum == 2 || c.hmix_tracer == 2)) {
    double ref = 2.0 * pi * RADIUS / nxg; ref = ref * ref;
    for (size_t p = 0; p < A2; ++p) { AMF[p] = std::sqrt(UAREA[p] / ref); AHF[p] = std::sqrt(TAREA[p] / ref); }
    host_halo_r8_loc(h, AMF.data(), 1, 0.0, 1, 0);                 // NE corner (hmix_del2.F90:262)
    host_halo_r8_loc(h, AHF.data(), 1, 0.0, 0, 0);                 // centre (:583)
  }
  auto &DUC = newf("DUC"), &DUN = newf("DUN"), &DUS = newf("DUS"), &DUE = newf("DUE"), &DUW = newf("DUW");
  auto &DMC = newf("DMC"), &DMN = newf("DMN"), &DMS = newf("DMS"), &DME = newf("DME"), &DMW = newf("DMW"), &DUM = newf("DUM");
  auto &DTN = newf("DTN"), &DTS = newf("DTS"), &DTE = newf("DTE"), &DTW = newf("DTW"), &KXU = newf("KXU"), &KYU = newf("KYU");
  {
    std::vector<double> WS(A2), WW(A2), KXT(A2), KYT(A2), W2a(A2), W2b(A2), W2c(A2), W2d(A2), WN(A2), WE(A2);
    auto each = [&](auto fn) { for (int b = 0; b < NB; ++b) for (int j = 0; j < nyb; ++j) for (int i = 0; i < nxb; ++i) fn(b, i, j, idx(b, i, j)); };
    each([&](int b, int i, int j, size_t p) {
      WS[p] = (HUS[p] / HTE[p]) * 0.5 * (AMF[p] + S(AMF, b, i, j - 1));
      WW[p] = (HUW[p] / HTN[p]) * 0.5 * (AMF[p] + S(AMF, b, i - 1, j));
      KXU[p] = (S(HUW, b, i + 1, j) - HUW[p]) * UAREA_R[p];
      KYU[p] = (S(HUS, b, i, j + 1) - HUS[p]) * UAREA_R[p];
      KXT[p] = (HTE[p] - S(HTE, b, i - 1, j)) * TAREA_R[p];
      KYT[p] = (HTN[p] - S(HTN, b, i, j - 1)) * TAREA_R[p];
      WN[p] = (HTN[p] / HUW[p]) * 0.5 * (AHF[p] + S(AHF, b, i, j + 1));
      WE[p] = (HTE[p] / HUS[p]) * 0.5 * (AHF[p] + S(AHF, b, i + 1, j));
    });
    each([&](int b, int i, int j, size_t p) {
      DUS[p] = WS[p] * UAREA_R[p]; DUN[p] = S(WS, b, i, j + 1) * UAREA_R[p];
      DUW[p] = WW[p] * UAREA_R[p]; DUE[p] = S(WW, b, i + 1, j) * UAREA_R[p];
      W2a[p] = 0.5 * (KXT[p] + S(KXT, b, i, j + 1)) * 0.5 * (S(AMF, b, i - 1, j) + AMF[p]);   // for DXKX
      W2b[p] = 0.5 * (KXT[p] + S(KXT, b, i + 1, j)) * 0.5 * (S(AMF, b, i, j - 1) + AMF[p]);   // for DYKX
      W2c[p] = 0.5 * (KYT[p] + S(KYT, b, i + 1, j)) * 0.5 * (S(AMF, b, i, j - 1) + AMF[p]);   // for DYKY
      W2d[p] = 0.5 * (KYT[p] + S(KYT, b, i, j + 1)) * 0.5 * (S(AMF, b, i - 1, j) + AMF[p]);   // for DXKY
      DTN[p] = WN[p] * TAREA_R[p]; DTS[p] = S(WN, b, i, j - 1) * TAREA_R[p];
      DTE[p] = WE[p] * TAREA_R[p]; DTW[p] = S(WE, b, i - 1, j) * TAREA_R[p];
    });
    each([&](int b, int i, int j, size_t p) {
      const double DXKX = (S(W2a, b, i + 1, j) - W2a[p]) * DXUR[p];
      const double DYKX = (S(W2b, b, i, j + 1) - W2b[p]) * DYUR[p];
      const double DYKY = (S(W2c, b, i, j + 1) - W2c[p]) * DYUR[p];
      const double DXKY = (S(W2d, b, i + 1, j) - W2d[p]) * DXUR[p];
      DUM[p] = -(DXKX + DYKY + 2.0 * AMF[p] * (KXU[p] * KXU[p] + KYU[p] * KYU[p]));
      DMC[p] = DXKY - DYKX;
      const double w1 = (S(AMF, b, i, j + 1) - S(AMF, b, i, j - 1)) / (HTE[p] + S(HTE, b, i, j + 1));
      DME[p] = (2.0 * AMF[p] * KYU[p] + w1) / (HTN[p] + S(HTN, b, i + 1, j));
      const double w2 = (S(AMF, b, i + 1, j) - S(AMF, b, i - 1, j)) / (HTN[p] + S(HTN, b, i + 1, j));
      DMN[p] = -(2.0 * AMF[p] * KXU[p] + w2) / (HTE[p] + S(HTE, b, i, j + 1));
      DUC[p] = -(DUN[p] + DUS[p] + DUE[p] + DUW[p]);
      DMW[p] = -DME[p]; DMS[p] = -DMN[p];
    });
  }

  // ---------------- del4 operator weights (hmix_del4.F90:218-246, 262-370, 512-577) ----------------
  if (c.hmix_momentum == 4 || c.hmix_tracer == 4) {
    auto &A4 = newf("D4AMF"), &H4 = newf("D4AHF");
    for (size_t p = 0; p < A2; ++p) { A4[p] = 1.0; H4[p] = 1.0; }
    if (c.lvariable_hmix) {
      for (size_t p = 0; p < A2; ++p) { A4[p] = std::pow(UAREA[p] / h.uarea_equator, 1.5); H4[p] = std::pow(TAREA[p] / h.uarea_equator, 1.5); }
      host_halo_r8_loc(h, A4.data(), 1, 0.0, 1, 0);                // NE corner (hmix_del4.F90:237)
      host_halo_r8_loc(h, H4.data(), 1, 0.0, 0, 0);                // centre (:536)
    }
    auto &eUC = newf("d4DUC"), &eUN = newf("d4DUN"), &eUS = newf("d4DUS"), &eUE = newf("d4DUE"), &eUW = newf("d4DUW");
    auto &eMC = newf("d4DMC"), &eMN = newf("d4DMN"), &eMS = newf("d4DMS"), &eME = newf("d4DME"), &eMW = newf("d4DMW"), &eUM = newf("d4DUM");
    auto &eTN = newf("d4DTN"), &eTS = newf("d4DTS"), &eTE = newf("d4DTE"), &eTW = newf("d4DTW");
    std::vector<double> RS(A2), RW(A2), KXT(A2), KYT(A2), Xx(A2), Xy(A2), Yy(A2), Yx(A2), RN(A2), RE(A2);
    auto each = [&](auto fn) { for (int b = 0; b < NB; ++b) for (int j = 0; j < nyb; ++j) for (int i = 0; i < nxb; ++i) fn(b, i, j, idx(b, i, j)); };
    each([&](int b, int i, int j, size_t p) {
      RS[p] = HUS[p] / HTE[p]; RW[p] = HUW[p] / HTN[p];
      KXT[p] = (HTE[p] - S(HTE, b, i - 1, j)) * TAREA_R[p];
      KYT[p] = (HTN[p] - S(HTN, b, i, j - 1)) * TAREA_R[p];
      RN[p] = HTN[p] / HUW[p]; RE[p] = HTE[p] / HUS[p];
    });
    each([&](int b, int i, int j, size_t p) {
      eUS[p] = RS[p] * UAREA_R[p]; eUN[p] = S(RS, b, i, j + 1) * UAREA_R[p];
      eUW[p] = RW[p] * UAREA_R[p]; eUE[p] = S(RW, b, i + 1, j) * UAREA_R[p];
      Xx[p] = S(KXT, b, i + 1, j) - KXT[p];     // d/dx of KXT
      Xy[p] = S(KXT, b, i, j + 1) - KXT[p];     // d/dy of KXT
      Yy[p] = S(KYT, b, i, j + 1) - KYT[p];
      Yx[p] = S(KYT, b, i + 1, j) - KYT[p];
      eTN[p] = RN[p] * TAREA_R[p]; eTS[p] = S(RN, b, i, j - 1) * TAREA_R[p];
      eTE[p] = RE[p] * TAREA_R[p]; eTW[p] = S(RE, b, i - 1, j) * TAREA_R[p];
    });
    each([&](int b, int i, int j, size_t p) {
      const double DXKX = 0.5 * (Xx[p] + S(Xx, b, i, j + 1)) * DXUR[p];
      const double DYKX = 0.5 * (Xy[p] + S(Xy, b, i + 1, j)) * DYUR[p];
      const double DYKY = 0.5 * (Yy[p] + S(Yy, b, i + 1, j)) * DYUR[p];
      const double DXKY = 0.5 * (Yx[p] + S(Yx, b, i, j + 1)) * DXUR[p];
      eUM[p] = -(DXKX + DYKY + 2.0 * (KXU[p] * KXU[p] + KYU[p] * KYU[p]));
      eMC[p] = DXKY - DYKX;
      eME[p] = 2.0 * KYU[p] / (HTN[p] + S(HTN, b, i + 1, j));
      eMN[p] = -2.0 * KXU[p] / (HTE[p] + S(HTE, b, i, j + 1));
      eUC[p] = -(eUN[p] + eUS[p] + eUE[p] + eUW[p]);
      eMW[p] = -eME[p]; eMS[p] = -eMN[p];
    });
  }

  // ---------------- third-order upwind interpolation weights (advection.F90:420-562) ----------------
  // Defined where the reference defines them: zonal on ib-1..ie x jb..je, poloidal on ib..ie x jb-1..je.
  if (c.tadvect == 2) {
    const int km = h.km;
    const std::vector<double> &dz = h.dz;
    std::vector<double> dzc(km + 2);
    dzc[0] = dz[1];
    for (int k = 1; k <= km; ++k) dzc[k] = dz[k];
    dzc[km + 1] = dzc[km];
    for (auto &v : h.upw_z) v.assign(km + 1, 0.0);
    std::vector<double> &azp = h.upw_z[0], &bzp = h.upw_z[1], &gzp = h.upw_z[2], &azm = h.upw_z[3], &bzm = h.upw_z[4], &dzm = h.upw_z[5];
    for (int k = 1; k <= km - 1; ++k) {
      azp[k] = dz[k] * (2.0 * dz[k] + dzc[k - 1]) / ((dz[k] + dz[k + 1]) * (dzc[k - 1] + 2.0 * dz[k] + dz[k + 1]));
      bzp[k] = dz[k + 1] * (2.0 * dz[k] + dzc[k - 1]) / ((dz[k] + dz[k + 1]) * (dz[k] + dzc[k - 1]));
      gzp[k] = -(dz[k] * dz[k + 1]) / ((dz[k] + dzc[k - 1]) * (dz[k + 1] + dzc[k - 1] + 2.0 * dz[k]));
    }
    bzp[1] = bzp[1] + gzp[1]; gzp[1] = 0.0;
    azp[km] = 0.0; bzp[km] = 0.0; gzp[km] = 0.0;
    for (int k = 1; k <= km - 1; ++k) {
      azm[k] = dz[k] * (2.0 * dz[k + 1] + dzc[k + 2]) / ((dz[k] + dz[k + 1]) * (dz[k + 1] + dzc[k + 2]));
      bzm[k] = dz[k + 1] * (2.0 * dz[k + 1] + dzc[k + 2]) / ((dz[k] + dz[k + 1]) * (dz[k] + dzc[k + 2] + 2.0 * dz[k + 1]));
      dzm[k] = -(dz[k] * dz[k + 1]) / ((dz[k + 1] + dzc[k + 2]) * (dz[k] + dzc[k + 2] + 2.0 * dz[k + 1]));
    }
    azm[km - 1] = azm[km - 1] + dzm[km - 1]; dzm[km - 1] = 0.0;
    azm[km] = 0.0; bzm[km] = 0.0; dzm[km] = 0.0;
    auto &AXP = newf("TALFXP"), &BXP = newf("TBETXP"), &GXP = newf("TGAMXP"), &AXM = newf("TALFXM"), &BXM = newf("TBETXM"), &DXM = newf("TDELXM");
    auto &AYP = newf("TALFYP"), &BYP = newf("TBETYP"), &GYP = newf("TGAMYP"), &AYM = newf("TALFYM"), &BYM = newf("TBETYM"), &DYM = newf("TDELYM");
    for (int b = 0; b < NB; ++b) {
      const BlockInfo &Bk = h.all_blocks[b];
      const int ib = Bk.ib - 1, ie = Bk.ie - 1, jb = Bk.jb - 1, je = Bk.je - 1;   // 0-based physical domain of the block
      for (int j = jb; j <= je; ++j) for (int i = ib - 1; i <= ie; ++i) {
        const size_t p = idx(b, i, j);
        const double dxc = DXT[p], dxcw = DXT[p - 1], dxce = DXT[p + 1], dxce2 = DXT[p + 2];
        AXP[p] = dxc * (2.0 * dxc + dxcw) / ((dxc + dxce) * (dxcw + 2.0 * dxc + dxce));
        BXP[p] = dxce * (2.0 * dxc + dxcw) / ((dxc + dxcw) * (dxc + dxce));
        GXP[p] = -(dxc * dxce) / ((dxc + dxcw) * (dxcw + 2.0 * dxc + dxce));
        AXM[p] = dxc * (2.0 * dxce + dxce2) / ((dxc + dxce) * (dxce + dxce2));
        BXM[p] = dxce * (2.0 * dxce + dxce2) / ((dxc + dxce) * (dxc + 2.0 * dxce + dxce2));
        DXM[p] = -(dxc * dxce) / ((dxce2 + dxce) * (dxc + 2.0 * dxce + dxce2));
      }
      for (int j = jb - 1; j <= je; ++j) for (int i = ib; i <= ie; ++i) {
        const size_t p = idx(b, i, j);
        const double dyc = DYT[p], dycs = DYT[p - nxb], dycn = DYT[p + nxb], dycn2 = DYT[p + 2 * (size_t)nxb];
        AYP[p] = dyc * (2.0 * dyc + dycs) / ((dyc + dycn) * (dycs + 2.0 * dyc + dycn));
        BYP[p] = dycn * (2.0 * dyc + dycs) / ((dyc + dycn) * (dycs + dyc));
        GYP[p] = -(dyc * dycn) / ((dyc + dycs) * (dycs + 2.0 * dyc + dycn));
        AYM[p] = dyc * (2.0 * dycn + dycn2) / ((dyc + dycn) * (dycn + dycn2));
        BYM[p] = dycn * (2.0 * dycn + dycn2) / ((dyc + dycn) * (dyc + 2.0 * dycn + dycn2));
        DYM[p] = -(dyc * dycn) / ((dycn2 + dycn) * (dyc + 2.0 * dycn + dycn2));
      }
    }
  }

  // ---------------- barotropic operator, null-space fields ----------------
  auto &WNE = newf("btropWgtNE"), &WEa = newf("btropWgtEast"), &WNo = newf("btropWgtNorth"), &WC0 = newf("centerWgtIndep");
  auto &mMask = newf("mMask"), &CHECKER = newf("CHECKER"), &CONSTNT = newf("CONSTNT");
  newf("centerWgt");
  // the two U-point terms every off-centre weight is made of (r3: the two-cell step B of the fused pcg reads these two fields and
  // forms WNE, WEa, WNo of its stencil from them -- the additions below, same operands, same order -- instead of reading three)
  auto &XW = newf("btropXW"), &YW = newf("btropYW");
  for (size_t p = 0; p < A2; ++p) { XW[p] = 0.25 * HU[p] * DXUR[p] * DYU[p]; YW[p] = 0.25 * HU[p] * DYUR[p] * DXU[p]; }
  {
    std::vector<double> area2(A2, 0.0), CA(A2, 0.0), KA(A2, 0.0);
    for (int b = 0; b < NB; ++b) {
      const BlockInfo &B = h.all_blocks[b];
      for (int j = 1; j < nyb; ++j) for (int i = 1; i < nxb; ++i) {
        auto q = [&](int di, int dj) { return idx(b, i + di, j + dj); };
        auto xw = [&](size_t p) { return 0.25 * HU[p] * DXUR[p] * DYU[p]; };
        auto yw = [&](size_t p) { return 0.25 * HU[p] * DYUR[p] * DXU[p]; };
        const double xne = xw(q(0, 0)), xse = xw(q(0, -1)), xnw = xw(q(-1, 0)), xsw = xw(q(-1, -1));
        const double yne = yw(q(0, 0)), yse = yw(q(0, -1)), ynw = yw(q(-1, 0)), ysw = yw(q(-1, -1));
        const size_t p = q(0, 0);
        WNE[p] = xne + yne;
        const double ase = xse + yse, anw = xnw + ynw, asw = xsw + ysw;
        WEa[p] = xne + xse - yne - yse;
        WNo[p] = yne + ynw - xne - xnw;
        WC0[p] = -(WNE[p] + ase + anw + asw);
        area2[p] = TAREA[p] * TAREA[p];
        mMask[p] = RCALCT[p];
      }
      for (int j = 0; j < nyb; ++j) for (int i = 0; i < nxb; ++i) {
        const size_t p = idx(b, i, j);
        const int n = B.i_glob[i] + std::abs(B.j_glob[j]);
        if (KMT[p] > 0) { CHECKER[p] = 2 * (n % 2) - 1; CONSTNT[p] = 1.0; CA[p] = CHECKER[p] * TAREA[p]; KA[p] = TAREA[p]; }
      }
    }
    h.residualNorm = 1.0 / host_global_sum(h, area2.data(), mMask.data());
    h.convergenceCriterion = (c.convergence_criterion * c.convergence_criterion) / h.residualNorm;
    const double sum_check = host_global_sum(h, CHECKER.data(), nullptr), sum_const = host_global_sum(h, CONSTNT.data(), nullptr);
    const double acheck = host_global_sum(h, CA.data(), nullptr) / host_global_sum(h, KA.data(), nullptr);
    h.rcheck = acheck / (sum_const - acheck * sum_check);
    h.rconst = 1.0 / (sum_const - acheck * sum_check);
  }

  // ---------------- time step ----------------
  h.dtt = 86400.0 / (double)c.steps_per_day;
  h.nsteps_per_interval = c.steps_per_day;
  if (c.tmix_opt == 2) {
    const int f = c.time_mix_freq;
    if (f <= 3) { h.err = "time_mix_freq must be > 3 for avgfit"; return 1; }
    int full = std::max(1, c.steps_per_day), half = (f + full) / (f - 1);
    if ((full + half) % f == 0) { full += 1; half = (f + full) / (f - 1); }
    if (full == 1 && half == 1) full += 1;
    h.nsteps_per_interval = full + half;
    h.dtt = 86400.0 / (full + 0.5 * half);
  }
  h.dtu = h.dtt; h.dtp = h.dtt;
  for (int k = 1; k <= h.km; ++k) h.dt[k] = h.dtt * 1.0;

  // ---------------- Robert filter coefficients, budget areas and volumes ----------------
  // time_management.F90:897-945; step_mod.F90:1577-1615 (MASK_TRBUDGET = KMT >= k; without region masks the
  // open-ocean mask is KMT >= k .and. RCALCT > 0, grid.F90:1112-1121)
  if (c.tmix_opt == 3) {
    double alpha = c.robert_alpha, nu = c.robert_nu;
    if (alpha == 0.0) alpha = 0.53;
    if (nu == 0.0) nu = 0.20;
    h.robert_curtime = 0.5 * nu * alpha;
    h.robert_newtime = 0.5 * nu * (alpha - 1.0);
    h.rf_nonzero_newtime = !(h.robert_newtime == 0.0);
    if (h.rf_nonzero_newtime && std::fabs(alpha - 1.0) <= 1.0e-6) { h.robert_curtime = 0.5 * nu; h.robert_newtime = 0.0; h.rf_nonzero_newtime = 0; }
    std::vector<double> mask(A2);
    const std::vector<int> &KMTi = h.i2["KMT"];
    const std::vector<double> &RC = h.f2["RCALCT"];
    for (int k = 1; k <= h.km; ++k) {
      for (size_t p = 0; p < A2; ++p) mask[p] = (KMTi[p] >= k) ? 1.0 : 0.0;
      const double bg = host_global_sum(h, TAREA.data(), mask.data());
      if (k == 1) h.bgtarea_t_1 = bg;
      const double rfthick = bg * h.dz[k];
      if (k >= 2) h.rf_volume_2_km = h.rf_volume_2_km + rfthick;
      for (size_t p = 0; p < A2; ++p) mask[p] = (KMTi[p] >= k && RC[p] > 0.0) ? 1.0 : 0.0;
      const double oo = host_global_sum(h, TAREA.data(), mask.data());
      if (k >= 2) h.open_ocean_volume_2_km = h.open_ocean_volume_2_km + oo * h.dz[k];
    }
  }

  // ---------------- analytic wind stress ----------------
  auto &SMFX = newf("SMF1"), &SMFY = newf("SMF2"), &SMFTX = newf("SMFT1"), &SMFTY = newf("SMFT2");
  for (size_t p = 0; p < A2; ++p) {
    const double s = -std::cos(3.0 * ULAT[p]), st = -std::cos(3.0 * TLAT[p]);
    SMFY[p] = -std::sin(0.0) * s; SMFTY[p] = -std::sin(0.0) * st;
    SMFX[p] = std::cos(0.0) * s; SMFTX[p] = std::cos(0.0) * st;
  }

  // ---------------- initial T,S on the local blocks (Levitus 1992 mean + perturbation) ------
  static const double zlev[33] = {0, 10, 20, 30, 50, 75, 100, 125, 150, 200, 250, 300, 400, 500, 600, 700, 800, 900,
    1000, 1100, 1200, 1300, 1400, 1500, 1750, 2000, 2500, 3000, 3500, 4000, 4500, 5000, 5500};
  static const double tlev[33] = {18.27, 18.22, 18.09, 17.87, 17.17, 16.11, 15.07, 14.12, 13.29, 11.87, 10.78, 9.94,
    8.53, 7.35, 6.38, 5.65, 5.06, 4.57, 4.13, 3.80, 3.51, 3.26, 3.05, 2.86, 2.47, 2.19, 1.78, 1.49, 1.26, 1.05, 0.91,
    0.87, 1.00};
  static const double slev[33] = {34.57, 34.67, 34.73, 34.79, 34.89, 34.97, 35.01, 35.03, 35.03, 34.98, 34.92, 34.86,
    34.76, 34.68, 34.63, 34.60, 34.59, 34.60, 34.61, 34.63, 34.65, 34.66, 34.68, 34.70, 34.72, 34.74, 34.75, 34.74,
    34.74, 34.73, 34.73, 34.72, 34.72};
  auto &T0 = h.f3["TEMP0"], &S0 = h.f3["SALT0"];
  T0.assign(h.n3 * h.nblocks, 0.0); S0.assign(h.n3 * h.nblocks, 0.0);
  const double amp = c.init_ts_perturbation;
  for (int lb = 0; lb < h.nblocks; ++lb) {
    const int gb = h.local_ids[lb] - 1;
    const BlockInfo &B = h.all_blocks[gb];
    for (int k = 1; k <= h.km; ++k) {
      const double dm = h.zt[k] * 0.01;
      int kk = 0;
      while (kk < 31 && !(dm >= zlev[kk] && dm < zlev[kk + 1])) ++kk;
      const double w = (dm - zlev[kk]) / (zlev[kk + 1] - zlev[kk]);
      const double tin = (1.0 - w) * tlev[kk] + w * tlev[kk + 1], sin_ = (1.0 - w) * slev[kk] + w * slev[kk + 1];
      for (int j = 0; j < nyb; ++j) for (int i = 0; i < nxb; ++i)
        if (k <= KMT[idx(gb, i, j)]) {
          // ghost rows beyond a tripole fold carry no global index of their own (j_glob < 0): the value of the centre cell they
          // mirror, row ny + 1 - n at column nx + 1 - i -- what a halo update of the physical cells would deliver
          int ige = B.i_glob[i], jge = B.j_glob[j];
          if (jge < 0) { jge = 2 * nyg + 1 + jge; ige = nxg + 1 - ige; if (ige < 1) ige += nxg; if (ige > nxg) ige -= nxg; }
          const double dT = amp * std::sin(2.0 * pi * ige / (double)nxg) * std::cos(pi * jge / (double)nyg);
          const size_t q = lb * h.n3 + (size_t)(k - 1) * n2 + (size_t)j * nxb + i;
          T0[q] = tin + dT; S0[q] = sin_ * 1.e-3;
        }
    }
  }
  // ---------------- tripole: redundant top-row points of N-face / NE-corner fields ----------------
  if (c.ns_boundary == 2) {
    auto &DUP = newf("TRIPOLE_DUP");
    for (int b = 0; b < NB; ++b) {
      const BlockInfo &B = h.all_blocks[b];
      if (!(B.j_glob[B.je] < 0)) continue;
      for (int i = B.ib; i <= B.ie; ++i) if (B.i_glob[i - 1] > c.nx_global / 2) DUP[idx(b, i - 1, B.je - 1)] = 1.0;
    }
  }
  // ---------------- P-CSI preprocessing (POP_SolversPrep) ----------------
  if (use_evp(c) || c.solver_choice == 3) {   // POP_SolversPrep (POP_SolversMod.F90:181-320): EVP first, then Lanczos
    const std::vector<double> C0 = host_center_init(h);
    if (use_evp(c)) h.f2["evpC0"] = C0;       // the centre weight the EVP coefficients are made of: k_evp_apply_wave2 reads it as a 2-D field
    if (use_evp(c) && host_evp_prep(h, C0)) return 1;
    if (c.solver_choice == 3 && host_pcsi_prep(h, C0)) return 1;
  }
  if (c.solver_choice < 1 || c.solver_choice > 3) { h.err = "solver_choice: 1 pcg, 2 ChronGear, 3 PCSI"; return 1; }
  return 0;
}

}  // namespace pop
