"""Run the HPCG benchmark driver on one GPU: python hpcg_run.py n [total_runtime] [parts]."""
import sys, json
sys.path.insert(0, '.')
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rt = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
P = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rep = pa.hpcg_benchmark(pa.DebugArray(list(range(1, P + 1))), P, n, n, n, total_runtime=rt, output_type="json", output_folder="gpurun_out/hpcg")
print(json.dumps({k: rep[k] for k in ("procs", "nr_equations", "iter_data", "times", "GFLOP/s", "GB/s", "reference_phase", "optimised_phase", "Overview")}))
