# PMC passes over the ring kernel (k_spmv_xring): where do its waves spend their cycles?  gpurun -- 'bash tools/probe/xring_pmc.sh'
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/xring_pmc
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d gpurun_out/xring_pmc/$tag -o out --output-format csv -- python tools/probe/xring_rate.py 16000,7900 > /dev/null 2>&1
  python - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/xring_pmc/{tag}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "xring" not in k: continue
        acc[(k[:60], r.get("Grid_Size"), r.get("Workgroup_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(tag, k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
done
