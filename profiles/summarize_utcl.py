"""Per-launch averages of the counters profiles/run_utcl.sh collected (gpurun_out/utcl/n<N>_<set>/...counter_collection.csv)
for the product kernel -> profiles/<tag>_utcl.json."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "gpurun_out", "utcl")
out = {}
for d in sorted(glob.glob(os.path.join(src, "n*_*"))):
    if not os.path.isdir(d):
        continue
    n = os.path.basename(d).split("_")[0]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "k_spmv_rowsplit" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
        for k in acc:
            out.setdefault(n, {})[k] = acc[k] / cnt[k]
    log = d + ".log"
    if os.path.exists(log):
        last = [l for l in open(log) if "mul!" in l]
        if last:
            out.setdefault(n, {})["line"] = last[-1].strip()
json.dump(out, open(os.path.join(here, f"{tag}_utcl.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
