cd $GRAFT_REPO_ROOT
PA_SETUP_TIMING=1 timeout 900 python tools/probe/mg_time2.py > gpurun_out/r03q_mg_time.log 2>&1
grep -E "MG-PCG|pa arena\] [+-]" gpurun_out/r03q_mg_time.log | cut -c1-200 | tail -24
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03q_pytest.log 2>&1
tail -3 gpurun_out/r03q_pytest.log
