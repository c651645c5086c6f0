cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "sliding or fused_product or x_window or unstructured" > gpurun_out/r03o_pytest.log 2>&1
tail -2 gpurun_out/r03o_pytest.log
for lanes in 256 512; do
  echo "== PA_SPMV_XWIN_BIG_LANES=$lanes"
  PA_SPMV_XWIN_BIG_LANES=$lanes timeout 900 python tools/probe/xring_rate.py 7000,6000,5000 2>/dev/null | cut -c1-400
done > gpurun_out/r03o_xwin_big_lanes.log 2>&1
cat gpurun_out/r03o_xwin_big_lanes.log
