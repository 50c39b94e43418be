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
  return sm;
}

}  // namespace pop
