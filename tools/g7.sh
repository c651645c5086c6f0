cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
bash tools/probe/mg_kstats2.sh > gpurun_out/g7_mgk.log 2>&1; cat gpurun_out/g7_mgk.log
