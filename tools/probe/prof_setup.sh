cd $GRAFT_REPO_ROOT
PA_SETUP_TIMING=1 python - <<'PY' > gpurun_out/r03z_setup_profile.log 2>&1
import sys, cProfile, pstats, time, io
sys.path.insert(0, '.')
from __graft_entry__ import load_package
pa = load_package()
n = 256
for ordering in ("multicolor_spmv", "sequential", "multicolor_spmv", "sequential"):
    pa.context().sync()
    pr = cProfile.Profile(); t = time.perf_counter(); pr.enable()
    S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, ordering=ordering)
    pa.context().sync()
    pr.disable(); dt = time.perf_counter() - t
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
    print(f"==== {ordering}: {dt:.2f} s", file=sys.stderr)
    print(s.getvalue()[:3000], file=sys.stderr)
    del S
PY
awk '/==== sequential/{c++} c>=1' gpurun_out/r03z_setup_profile.log | grep -v "pa setup\]\|pa arena\]" | cut -c1-200 | head -70
