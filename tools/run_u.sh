cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r03u_pytest.log 2>&1
tail -3 gpurun_out/r03u_pytest.log
timeout 600 python bench.py > gpurun_out/r03u_bench.json 2> gpurun_out/r03u_bench.err
cut -c1-1500 gpurun_out/r03u_bench.json
timeout 900 python tools/hpcg_driver.py 1 256 30 > gpurun_out/r03u_hpcg256.log 2>&1
tail -2 gpurun_out/r03u_hpcg256.log | cut -c1-900
timeout 600 python tools/probe/mg_ab.py child $GRAFT_REPO_ROOT this 256 2>/dev/null | grep "^\[" | tee gpurun_out/r03u_mg256.log
timeout 600 python tools/probe/mg_ab.py child $GRAFT_REPO_ROOT this 128 2>/dev/null | grep "^\[" | tee gpurun_out/r03u_mg128.log
