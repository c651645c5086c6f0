"""Per-launch averages of the counters profiles/run_k1_sq.sh collected (gpurun_out/k1_sq/<tag>/...counter_collection.csv) for the
product kernel (k_spmv_rowsplit / k_spmv_persist) -> profiles/<round>_k1_sq.json, with a few derived figures."""
import collections, csv, glob, json, os, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "gpurun_out", "k1_sq")
out = {}
for d in sorted(glob.glob(os.path.join(src, "vd*_m*_s*"))):
    if not os.path.isdir(d):
        continue
    key = "_".join(os.path.basename(d).split("_")[:2])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "k_spmv_rowsplit" not in r["Kernel_Name"] and "k_spmv_pell" not in r["Kernel_Name"]:      # (round 6: pattern blocks run on k_spmv_pell)
                continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
            out.setdefault(key, {})["kernel"] = r["Kernel_Name"][:60]
            for k in ("VGPR_Count", "SGPR_Count", "Accum_VGPR_Count", "LDS_Block_Size", "Grid_Size", "Workgroup_Size"):
                if k in r:
                    out[key][k] = r[k]
        for k in acc:
            out.setdefault(key, {})[k] = acc[k] / cnt[k]
for key, c in out.items():
    d = {}
    if c.get("SQ_WAVE_CYCLES") and c.get("SQ_BUSY_CYCLES"):
        d["mean_waves_per_SQ_busy_cycle"] = c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CYCLES"]
    if c.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if k in c:
                d[k + "_over_WAVE_CYCLES"] = c[k] / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_WAVES"):
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_SMEM"):
            if k in c:
                d[k + "_per_wave"] = c[k] / c["SQ_WAVES"]
    if c.get("TCP_TCC_READ_REQ_sum") and c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        d["TCP_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if c.get("TCP_TCC_READ_REQ_LATENCY_sum") and c.get("TCP_TCC_READ_REQ_sum"):
        d["mean_TCP_to_TCC_read_latency_cycles"] = c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"]
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum"):
        d["TCC_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    c["derived"] = d
json.dump(out, open(os.path.join(here, f"{rnd}_k1_sq.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v.get("derived") for k, v in out.items()}, indent=1, sort_keys=True))
