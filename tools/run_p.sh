cd $GRAFT_REPO_ROOT
timeout 900 python tools/probe/mg_time2.py > gpurun_out/r03p_mg_time.log 2>&1
cat gpurun_out/r03p_mg_time.log | tail -4
PA_SETUP_TIMING= timeout 1500 python tools/hpcg_driver.py 1 256 30 > gpurun_out/r03p_hpcg256.log 2>&1
tail -2 gpurun_out/r03p_hpcg256.log | cut -c1-600
