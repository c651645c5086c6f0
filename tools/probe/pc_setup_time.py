"""pc_setup (4 levels, 256^3, multicolour) with and without the automatic value dictionary: wall seconds, third build of each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
ctx = pa.context()
r1 = pa.DebugArray([1])
for tag, env in (("dictionary off", "0"), ("dictionary auto", None), ("dictionary off", "0"), ("dictionary auto", None)):
    if env is None:
        os.environ.pop("PA_SPMV_VALUE_DICT", None)
    else:
        os.environ["PA_SPMV_VALUE_DICT"] = env
    for ordering in ("multicolor_spmv", "sequential"):
        ctx.sync(); t = time.perf_counter()
        S = pa.pc_setup(r1, 1, 4, 256, 256, 256, ordering=ordering)
        ctx.sync()
        print(f"{tag:16s} {ordering:16s} {time.perf_counter() - t:.3f} s", flush=True)
        del S
