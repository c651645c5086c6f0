cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "sliding or fused_product or x_window" > gpurun_out/r03m_pytest.log 2>&1
tail -2 gpurun_out/r03m_pytest.log
for lanes in 256 512; do
  echo "== PA_SPMV_XRING_LANES=$lanes"
  PA_SPMV_XRING_LANES=$lanes timeout 900 python tools/probe/xring_rate.py 7900,6000 2>/dev/null | cut -c1-400
done > gpurun_out/r03m_xring_lanes.log 2>&1
cat gpurun_out/r03m_xring_lanes.log
