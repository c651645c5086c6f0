cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "disassembled or fem or psparse or config5 or c5 or reassembly" > gpurun_out/r03z_pytest.log 2>&1
tail -25 gpurun_out/r03z_pytest.log | cut -c1-300
