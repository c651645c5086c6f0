import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import cProfile, pstats
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ranks = pa.DebugArray(list(range(1, 5)))
t = time.perf_counter()
I, J, V, rows, cols = pa.laplacian_fem((n, n), (2, 2), ranks)
print("generate", round(time.perf_counter() - t, 2), "s; triplets per part", [len(i) for i in I.items])
t = time.perf_counter()
A = pa.psparse_disassembled(I, J, V, rows, cols)
pa.context().sync()
print("device route, no reuse", round(time.perf_counter() - t, 2))
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter()
A2, cache = pa.psparse_disassembled(I, J, V, rows, cols, reuse=True)
pa.context().sync()
print("reuse=True (host cache set-up)", round(time.perf_counter() - t, 2))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
t = time.perf_counter()
pa.psparse_(A2, V, cache).wait(); pa.context().sync()
print("psparse!", round(time.perf_counter() - t, 3))
