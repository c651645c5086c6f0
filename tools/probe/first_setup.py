import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["PA_SETUP_TIMING"] = "1"
from __graft_entry__ import load_package
pa = load_package()
ctx = pa.context()
r1 = pa.DebugArray([1])
for k in range(2):
    print(f"==== set-up {k} starts", file=sys.stderr, flush=True)
    ctx.sync(); t = time.perf_counter()
    S = pa.pc_setup(r1, 1, 4, 256, 256, 256, ordering="sequential", keep_raw_columns=True); ctx.sync()
    print(f"==== set-up {k}: {time.perf_counter() - t:.3f} s", file=sys.stderr, flush=True)
    del S
    import gc; gc.collect()
