# per-kernel AND per-grid-size time of the MG-PCG probe (256^3): rocprofv3 --kernel-trace; prints calls / avg us by (kernel, grid)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mgks
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/mgks -o k -- python $R/tools/probe/mg_ab.py child $R stats 256 > /tmp/mgks.log 2>&1 < /dev/null
grep "MG-PCG" /tmp/mgks.log | cut -c1-160
f=$(find /tmp/mgks -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# keep the last 60 % of the trace (the timed iterations; the set-up comes first)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.4):]
acc = collections.defaultdict(lambda: [0, 0])
for r in rows:
    m = re.match(r"void (k\w+)<([^>]*)>", r["Kernel_Name"])
    name = (m.group(1) + "<" + m.group(2) + ">") if m else r["Kernel_Name"].split("(")[0]
    key = (name[:48], int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0))
    acc[key][0] += 1; acc[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in acc.values())
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"kernel time {tot/1e6:.1f} ms of a span of {span/1e6:.1f} ms")
for (name, grid), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{name:48s} grid {grid:9d} calls {n:6d} avg {t/n/1e3:8.1f} us  {100*t/tot:5.1f} %")
PY
