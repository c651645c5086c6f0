cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03c_pytest.log 2>&1
tail -3 gpurun_out/r03c_pytest.log
PA_SETUP_TIMING=1 timeout 600 python tools/probe/setup_profile.py 256 > gpurun_out/r03c_setup_profile.log 2>&1
timeout 1500 bash profiles/run_utcl.sh > gpurun_out/r03c_utcl.log 2>&1
python profiles/summarize_utcl.py r03 > gpurun_out/r03c_utcl_summary.log 2>&1
cp profiles/r03_utcl.json gpurun_out/ 2>/dev/null
