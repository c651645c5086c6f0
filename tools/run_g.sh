cd $GRAFT_REPO_ROOT
for v in product tilelds nogather product tilelds; do
  if [ $v = product ]; then unset PA_HIP_LIBRARY; else export PA_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/probe/build/libpa_hip_$v.so; fi
  timeout 300 python tools/probe/headline_whatif.py $v 2>/dev/null | grep "27-pt"
done > gpurun_out/r03g_tile_lds_whatif.log 2>&1
unset PA_HIP_LIBRARY
cat gpurun_out/r03g_tile_lds_whatif.log
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03g_pytest.log 2>&1
tail -3 gpurun_out/r03g_pytest.log
PA_SETUP_TIMING=1 timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err
