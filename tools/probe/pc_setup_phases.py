"""Third pc_setup (4 levels, 256^3, multicolour, library defaults) with PA_SETUP_TIMING=1: the library's per-phase stderr lines of
that one set-up, summed per phase name.   python tools/probe/pc_setup_phases.py 2> phases.log"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
ctx = pa.context()
r1 = pa.DebugArray([1])
for k in range(2):
    S = pa.pc_setup(r1, 1, 4, 256, 256, 256, ordering="multicolor_spmv"); ctx.sync(); del S
os.environ["PA_SETUP_TIMING"] = "1"
print("==== timed set-up starts", file=sys.stderr, flush=True)
ctx.sync(); t = time.perf_counter()
S = pa.pc_setup(r1, 1, 4, 256, 256, 256, ordering="multicolor_spmv")
ctx.sync()
print(f"==== timed set-up ends {time.perf_counter() - t:.3f} s", file=sys.stderr, flush=True)
