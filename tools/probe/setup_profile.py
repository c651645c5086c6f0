"""Where pc_setup's time goes (multicolour smoother, one part)."""
import sys, cProfile, pstats
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pr = cProfile.Profile(); pr.enable()
S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, ordering="multicolor_spmv")
pa.context().sync()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
