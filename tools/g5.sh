cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export PA_HIP_LIBRARY=$PWD/partitionedarrays.jl_amd/libpa_hip.so.d6
timeout 900 python -m pytest tests/test_gpu_pattern_ell.py tests/test_gpu_spmv_kernels.py tests/test_gpu_value_dict.py tests/test_gpu_hpcg_mg.py tests/test_gpu_mul.py tests/test_gpu_blas1_cg.py tests/test_gpu_exchange_chain.py -q 2>&1 | tail -8
timeout 600 python tests/fuzz/fuzz_spmv.py 40 424200 2>&1 | tail -2
timeout 600 python tests/fuzz/fuzz_hpcg.py 30 424700 2>&1 | tail -2
timeout 900 python tools/probe/pell_lean_time.py 256 mg > gpurun_out/g6_lean.log 2>&1; grep -v "^{" gpurun_out/g6_lean.log | cut -c1-30,225-330 | tail -12
timeout 600 python bench.py > gpurun_out/g6_bench.json 2> gpurun_out/g6_bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/g6_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "lib defaults", d.get("ms_per_step_library_defaults"))
for e in d.get("extra_configs", []):
    print({k: e[k] for k in e if k in ("ms", "gflops", "frac_moved", "ms_per_iteration", "ms_per_part", "with_default_value_dictionary")}, e["workload"][:50])
for e in d.get("general_csr", []): print(e["ms"], e["frac_moved"], e["kernel"][:40], e["workload"][:60])
PY
rm -rf gpurun_out/k1_sq
bash profiles/run_k1_sq.sh 2>&1 | cut -c1-20 | tr '\n' ' '
python profiles/summarize_k1_sq.py r06_lean > gpurun_out/g6_sq_summary.json 2>&1; cp profiles/r06_lean_k1_sq.json gpurun_out/
for f in gpurun_out/k1_sq/*.log; do tail -2 $f | cut -c1-200; done > gpurun_out/g6_sq_logs.txt
rm -rf gpurun_out/k1_sq
