import os, sys, time, gc
sys.path.insert(0, "/root/repo")
os.environ["PA_SPMV_VALUE_DICT"] = "0"
from __graft_entry__ import load_package
pa = load_package()
ctx = pa.context()
r1 = pa.DebugArray([1])
for rep in range(2):
    os.environ["PA_SPMV_VALUE_DICT"] = "0"
    ctx.sync(); t = time.perf_counter()
    S = pa.pc_setup(r1, 1, 4, 256, 256, 256, ordering="multicolor_spmv"); ctx.sync()
    a = time.perf_counter() - t
    del S; gc.collect()
    os.environ.pop("PA_SPMV_VALUE_DICT")
    ctx.sync(); t = time.perf_counter()
    S = pa.pc_setup(r1, 1, 4, 256, 256, 256, ordering="multicolor_spmv"); ctx.sync()
    b = time.perf_counter() - t
    del S; gc.collect()
    print(f"rep {rep}: without dictionary {a:.3f} s, with {b:.3f} s", flush=True)
