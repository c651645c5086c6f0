# per-kernel time of the MG-PCG probe (256^3): rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mgks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mgks -o k -- python $R/tools/probe/mg_ab.py child $R stats 256 > /tmp/mgks.log 2>&1 < /dev/null
grep "MG-PCG" /tmp/mgks.log | cut -c1-160
f=$(find /tmp/mgks -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/r03_mg_kernel_stats.csv && python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    m = re.match(r"void (k_\w+)<([^>]*)>", r["Name"])
    name = (m.group(1) + "<" + m.group(2) + ">") if m else r["Name"].split("(")[0]
    print(f"{name[:70]:70s} calls {int(r['Calls']):6d} avg {float(r['AverageNs']) / 1e3:9.1f} us total {int(r['TotalDurationNs']) / 1e6:8.1f} ms")
PY
