cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "colour or color or multicolor or mg_ or hpcg or gauss or smoother or graph or row_subsets or fused_residual" > gpurun_out/r03z_pytest.log 2>&1
tail -8 gpurun_out/r03z_pytest.log | cut -c1-300
for m in 1 0; do
PA_GS_LOWER=$m timeout 300 python tools/probe/mg_ab.py child $GRAFT_REPO_ROOT "lower=$m" 256 2>/dev/null | grep "^\[" | tee -a gpurun_out/r03z_mg256b.log
done
