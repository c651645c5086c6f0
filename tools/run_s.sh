cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "generated_on_the_device or row_subsets or multicolor or mg_ or hpcg or gauss or sequential or smoother" > gpurun_out/r03x_pytest.log 2>&1
tail -12 gpurun_out/r03x_pytest.log | cut -c1-300
PA_SETUP_TIMING=1 python - <<'PY' > gpurun_out/r03x_setup_profile.log 2>&1
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
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18)
    print(f"==== {ordering}: {dt:.2f} s", file=sys.stderr)
    print(s.getvalue()[:3400], file=sys.stderr)
    del S
PY
grep -E "====" gpurun_out/r03x_setup_profile.log | cut -c1-200
timeout 900 python tools/hpcg_driver.py 1 256 30 > gpurun_out/r03x_hpcg256.log 2>&1
tail -1 gpurun_out/r03x_hpcg256.log | cut -c1-600
