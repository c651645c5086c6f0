cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -k "sliding or fused_product or x_window or device_side or unstructured or fuzzer_short" > gpurun_out/r03j_pytest.log 2>&1
tail -3 gpurun_out/r03j_pytest.log
timeout 1200 python tools/probe/xring_rate.py > gpurun_out/r03j_xring_rate.log 2>&1
cat gpurun_out/r03j_xring_rate.log | cut -c1-400
