import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import __graft_entry__ as ge
from popcfg import named_config
pkg = ge.load_package()
for rows in (2400, 300, 150, 75):
    m = pkg.PopModel(named_config("tx0.1v3", block_size_y=rows))
    print(rows, "blocks", m.nblocks, "chunks/block", m.dim("solver_chunks_per_block"), "listed/block", m.dim("solver_chunks_listed"), "active total", m.dim("solver_chunks_active"), flush=True)
    m.close()
