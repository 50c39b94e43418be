// halo_plan.cpp -- builds the ghost-cell update plan of one rank (host logic, no HIP).
//
// Behaviour restated from mpi/POP_HaloMod.F90:142-1640 (POP_HaloCreate) for non-tripole
// domains: every ghost point of every local block gets an explicit source address
// (i,j,block); sources on this rank become direct copies (:1895-1914), sources on another
// rank are packed per neighbour task into one message (:1865-1883), and points whose global
// index is 0 (closed boundary / padding) receive the fill value.  Here the address lists are
// flat cell indices  block_local*n2 + (j-1)*nx_block + (i-1)  so the device kernels gather /
// scatter whole levels with one indirection per cell.
#include <algorithm>
#include "pop_internal.hpp"

namespace pop {

void build_halo_plan(HostModel &h) {
  const pop_config &c = h.c;
  HaloPlan &P = h.halo;
  P = HaloPlan();
  std::map<int, PeerPlan> peers;
  auto source = [&](int gi, int gj, int &sb, int &cell) {   // owner block + cell index inside it
    const int sbx = (gi - 1) / c.block_size_x, sby = (gj - 1) / c.block_size_y;
    sb = sby * h.nbx + sbx;
    const int si = gi - sbx * c.block_size_x + NGHOST, sj = gj - sby * c.block_size_y + NGHOST;   // 1-based
    cell = (sj - 1) * h.nxb + (si - 1);
  };
  // walk every rank's blocks in that rank's local order; record what this rank receives
  // (when the walked block is ours) and what it must send (when the source block is ours)
  for (int r = 0; r < h.nranks; ++r) {
    for (int n = 0; n < h.nblocks_tot; ++n) {
      if (h.block_owner[n] != r) continue;
      const BlockInfo &B = h.all_blocks[n];
      const int dl = h.block_local[n];
      for (int j = 1; j <= h.nyb; ++j)
        for (int i = 1; i <= h.nxb; ++i) {
          if (i >= B.ib && i <= B.ie && j >= B.jb && j <= B.je) continue;
          const int gi = B.i_glob[i - 1], gj = B.j_glob[j - 1];
          const int dcell = (j - 1) * h.nxb + (i - 1);
          if (gi <= 0 || gj <= 0) {
            if (r == h.rank) P.fill_dst.push_back(dl * (int)h.n2 + dcell);
            continue;
          }
          int sb, scell;
          source(gi, gj, sb, scell);
          const int so = h.block_owner[sb];
          if (r == h.rank && so == h.rank) {
            P.copy_dst.push_back(dl * (int)h.n2 + dcell);
            P.copy_src.push_back(h.block_local[sb] * (int)h.n2 + scell);
          } else if (r == h.rank) {
            PeerPlan &pp = peers[so]; pp.rank = so;
            pp.recv_dst.push_back(dl * (int)h.n2 + dcell);
          } else if (so == h.rank) {
            PeerPlan &pp = peers[r]; pp.rank = r;
            pp.send_src.push_back(h.block_local[sb] * (int)h.n2 + scell);
          }
        }
    }
  }
  for (auto &kv : peers) {
    P.max_msg_cells += (long long)std::max(kv.second.send_src.size(), kv.second.recv_dst.size());
    P.peers.push_back(std::move(kv.second));
  }
  // ---- tripole northern boundary, written as the closed-form rule the reference's unit test states
  // (test/unit/halo/POP.F90Tripole:330-345 centre, :600-620 E face, :1112-1150 NE corner, N face alike):
  // ghost row n of a northern block mirrors global row ny+1-n (centre, E face) or ny-n (NE corner, N face) at
  // column nx-ig+1 (centre, N face) or nx-ig (0 -> nx; E face, NE corner); NE-corner and N-face fields also
  // replace their top physical row by the symmetrised value of the two degenerate points.
  // Several ranks: the fold only ever pairs cells of the top row of blocks, so it stays a rank-local operation as long as
  // that row has ONE owner (always true for full-width j-band blocks, the decomposition the multi-GPU runs use); that rank
  // gets the plan, the others an empty one.  tripole_g is the same plan with global block numbers for the set-up code,
  // whose host arrays hold every block on every rank.
  int top_owner = -1; bool top_split = false;
  if (c.ns_boundary == 2)
    for (int n = 0; n < h.nblocks_tot; ++n) {
      const BlockInfo &B = h.all_blocks[n];
      if (!(B.j_glob[B.je] < 0)) continue;
      if (top_owner < 0) top_owner = h.block_owner[n];
      else if (top_owner != h.block_owner[n]) top_split = true;
    }
  P.tripole_split = top_split;
  for (int pass = 0; pass < 2 && c.ns_boundary == 2 && !top_split; ++pass) {
    const bool global = pass == 1;
    if (!global && top_owner != h.rank) continue;
    const int nx = c.nx_global, ny = c.ny_global;
    auto cell_of = [&](int gi, int gj) { int sb, cell; source(gi, gj, sb, cell); return (global ? sb : h.block_local[sb]) * (int)h.n2 + cell; };
    for (int lc = 0; lc < 5; ++lc) {
      const int loc = lc == 4 ? 2 : lc;          // [4] = N face without the degenerate row
      TripolePlan &T = global ? P.tripole_g[lc] : P.tripole[lc];
      const int ioff = (loc == 1 || loc == 3) ? 1 : 0, joff = (loc == 1 || loc == 2) ? 1 : 0;
      for (int n = 0; n < h.nblocks_tot; ++n) {
        const BlockInfo &B = h.all_blocks[n];
        if (!(B.j_glob[B.je] < 0)) continue;                 // j_glob of local row je+1 (0-based index je)
        const int dl = global ? n : h.block_local[n];
        for (int jn = 0; jn <= NGHOST; ++jn) {               // jn = 0: top physical row
          if (jn == 0 && (!joff || lc == 4)) continue;
          const int gj = ny + 1 - jn - joff;                 // source row
          for (int i = 1; i <= h.nxb; ++i) {
            const int ig = B.i_glob[i - 1];
            if (ig <= 0) continue;
            int si = nx - ig + 1 - ioff; if (si == 0) si = nx;
            const int dcell = dl * (int)h.n2 + (B.je + jn - 1) * h.nxb + (i - 1);
            T.dst.push_back(dcell);
            if (jn == 0) {
              // row ny after symmetrisation, read at the mirrored column si: its own point is (si), its partner
              // is the column that mirrors si (ig - ... ) -- the value copied out is the symmetrised value AT si
              // times isign twice = sign(avg, F(si_partner...)); expressed on global points:
              //   NE corner: partner of column s is nx - s (s != nx/2, nx);  N face: partner of s is nx + 1 - s
              // and the copy-out address si is itself the mirror of ig, so the result is the symmetrised value of
              // column ig: sign(0.5(|F(ig)| + |F(partner(ig))|), F(ig)).
              const int partner = (loc == 1) ? ((ig == nx || ig == nx / 2) ? ig : nx - ig) : nx + 1 - ig;
              T.a.push_back(cell_of(ig, ny));
              T.b.push_back(cell_of(partner, ny));
            } else {
              T.a.push_back(cell_of(si, gj));
              T.b.push_back(-1);
            }
          }
        }
      }
    }
  }
}

// source map over ALL blocks (single-rank view of the decomposition): interior cells map to
// themselves, ghosts to the interior cell that owns their global index, -1 where the fill value applies
std::vector<int> global_srcmap(const HostModel &h) {
  const pop_config &c = h.c;
  std::vector<int> sm(h.n2 * h.nblocks_tot);
  for (int b = 0; b < h.nblocks_tot; ++b) {
    const BlockInfo &B = h.all_blocks[b];
    for (int j = 1; j <= h.nyb; ++j)
      for (int i = 1; i <= h.nxb; ++i) {
        const int cell = b * (int)h.n2 + (j - 1) * h.nxb + (i - 1);
        if (i >= B.ib && i <= B.ie && j >= B.jb && j <= B.je) { sm[cell] = cell; continue; }
        const int gi = B.i_glob[i - 1], gj = B.j_glob[j - 1];
        if (gi <= 0 || gj <= 0) { sm[cell] = -1; continue; }
        const int sbx = (gi - 1) / c.block_size_x, sby = (gj - 1) / c.block_size_y;
        const int si = gi - sbx * c.block_size_x + NGHOST, sj = gj - sby * c.block_size_y + NGHOST;
        sm[cell] = (sby * h.nbx + sbx) * (int)h.n2 + (sj - 1) * h.nxb + (si - 1);
      }
  }
  const TripolePlan &T = h.halo.tripole_g[0];   // centre scalars beyond the fold: mirrored copies of physical cells
  for (size_t e = 0; e < T.dst.size(); ++e) sm[T.dst[e]] = T.a[e];
  return sm;
}

}  // namespace pop
