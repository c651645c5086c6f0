cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "sliding or fused_product or x_window" > gpurun_out/r03k_pytest.log 2>&1
tail -2 gpurun_out/r03k_pytest.log
timeout 900 python tools/probe/xring_rate.py 7900,7000,5000 > gpurun_out/r03k_xring_rate.log 2>&1
cat gpurun_out/r03k_xring_rate.log | cut -c1-400
