"""Which mul! shapes survive hipGraph capture (pa_mul_all through mul_c_)?  Prints one line per case before trying it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
ctx = pa.context()
cases = [((8, 8, 8), (2, 1, 1), 1), ((8, 8, 8), (2, 1, 1), 40), ((32, 32, 32), (2, 1, 1), 1), ((64, 64, 64), (2, 1, 1), 1), ((128, 128, 128), (2, 1, 1), 1),
         ((32, 32, 32), (2, 2, 2), 40), ((128, 128, 128), (2, 1, 1), 1500), ((128, 128, 128), (2, 1, 1), -30)]
which = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for k, (n, np3, eager) in enumerate(cases):
    if which >= 0 and k != which:
        continue
    P = int(np.prod(np3))
    print("case", k, n, np3, "eager calls before capture:", eager, flush=True)
    A, _ = pa.build_p_matrix(pa.DebugArray(range(1, P + 1)), *n, *(a * q for a, q in zip(n, np3)), *np3)
    x = pa.pvector_from_function(lambda ind: np.ones(ind.n_local), A.col_partition)
    y = pa.pzeros(A.row_partition)
    import pa_amd._lib as L
    if eager < 0:                      # events around the eager calls, as a timing loop has them
        e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(abs(eager)):
        pa.mul_c_(y, A, x)
    if eager < 0:
        e1 = ctx.event().record(L.STREAM_COMPUTE)
        ctx.sync()
        print("  eager ms", e0.elapsed_ms(e1) / abs(eager), flush=True)
    ctx.sync()
    print("  capturing", flush=True)
    with pa.Graph() as g:
        pa.mul_c_(y, A, x)
    print("  captured", flush=True)
    for _ in range(3):
        g.launch()
    ctx.sync()
    print("  replayed", flush=True)
