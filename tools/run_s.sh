cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "colour or color or multicolor or mg_ or hpcg or gauss or smoother or graph or row_subsets or fused_residual" > gpurun_out/r03z_pytest.log 2>&1
tail -8 gpurun_out/r03z_pytest.log | cut -c1-300
timeout 900 python tools/probe/color_order.py 128 2>/dev/null | head -5 | tee gpurun_out/r03z_color_order2.log
timeout 900 python tools/hpcg_driver.py 1 256 30 > gpurun_out/r03z_hpcg256a.log 2>&1
tail -1 gpurun_out/r03z_hpcg256a.log | cut -c1-600
timeout 300 python tools/probe/mg_ab.py child $GRAFT_REPO_ROOT "affinity order" 256 2>/dev/null | grep "^\["
