cd $GRAFT_REPO_ROOT
timeout 1500 python tools/probe/mg_time3.py 2>/dev/null | grep "MG-PCG" > gpurun_out/r03r_mg_time3.log
cat gpurun_out/r03r_mg_time3.log | cut -c1-260
