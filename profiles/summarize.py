"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into the small summaries committed under profiles/.
usage: python profiles/summarize.py <round-tag>   e.g. r01"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
HEADLINE = ("k_spmv_pell", "k_spmv_rowsplit")     # the product kernel of the headline block (round 6: pattern-ELL; before: the row split)
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
out = {}
ks = os.path.join(src, "prof_kt", "kt_kernel_stats.csv")
if os.path.exists(ks):
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(os.path.dirname(__file__), f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
    out["kernel_stats"] = [{"name": r["Name"][:80], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                            "pct": float(r["Percentage"])} for r in rows[:6]]
# the headline-only pass: its timed steps (between the CLOCK_MONOTONIC bounds bench.py prints) run back to back; the event
# pass that follows brackets every launch with two event records, which leaves ~19 us between launches.  (Before bench.py
# ran its clock-ramp steps the first timed launches took up to 0.83 ms: the GPU was still coming back from the idle of the
# host-side parity gate.)
kth = os.path.join(src, "prof_kth", "kth_kernel_trace.csv")
blh = os.path.join(src, "bench_kth.log")
if os.path.exists(kth) and os.path.exists(blh):
    line = [l for l in open(blh) if l.startswith('{"metric"')]
    d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(kth)) if any(k in r["Kernel_Name"] for k in HEADLINE))
    if line and d:
        bench = json.loads(line[-1])
        lo, hi = bench["roofline"].get("timed_region_monotonic_ns", [0, 0])
        timed = [(a, b) for a, b in d if lo <= a <= hi]
        after = [(a, b) for a, b in d if a > hi][:bench["roofline"].get("launches_timed", 50)]

        def stats(seg):
            if not seg:
                return None
            dur = [(b - a) / 1e3 for a, b in seg]
            gap = [(seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(len(seg) - 1)]
            return {"launches": len(seg), "avg_us": sum(dur) / len(dur), "min_us": min(dur), "max_us": max(dur),
                    "avg_gap_to_next_us": sum(gap) / max(1, len(gap)), "period_us": (seg[-1][1] - seg[0][0]) / 1e3 / len(seg)}
        out["headline_only_pass"] = {
            "what": "k_spmv_rowsplit launches of `bench.py --no-cpu-baseline --no-value-dict --no-extra --cg-iters 0`: the timed steps "
                    "run back to back (a launch's duration then includes waiting for its predecessor's dirty lines to drain: the "
                    "period is what ms_per_step measures), the event pass leaves a gap before every launch (the kernel alone)",
            "timed_steps": stats(timed), "event_pass": stats(after), "bench_ms_per_step": bench["ms_per_step"],
            "bench_avg_launch_ms": bench["roofline"]["avg_launch_ms"],
            "bench_event_pass": bench["roofline"].get("event_pass"), "launches_total": len(d)}
# the default bench command's own timed region inside ITS trace: bench.py prints the CLOCK_MONOTONIC bounds
kt = os.path.join(src, "prof_kt", "kt_kernel_trace.csv")
bl = os.path.join(src, "bench_kt.log")
if os.path.exists(kt) and os.path.exists(bl):
    line = [l for l in open(bl) if l.startswith('{"metric"')]
    if line:
        bench = json.loads(line[-1])
        lo, hi = bench["roofline"].get("timed_region_monotonic_ns", [0, 0])
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt))
             if any(k in r["Kernel_Name"] for k in HEADLINE) and lo <= int(r["Start_Timestamp"]) <= hi]      # (only the headline product runs inside the bounds)
        # (own x ghost is the same kernel on an empty block at one part: no launch)
        out["default_command_timed_region"] = {
            "what": "launches of the headline kernel between the CLOCK_MONOTONIC bounds bench.py reports for its timed steps, "
                    "in the trace of the default command (profiles/rNN_kernel_stats.csv averages ALL launches of the process: "
                    "parity gate, warm-up, timed steps, the event pass, CG loop, value-dictionary mode, extra configs)",
            "launches": len(d), "avg_us": sum(d) / max(1, len(d)), "bench_ms_per_step": bench["ms_per_step"],
            "bench_avg_launch_ms": bench["roofline"]["avg_launch_ms"], "bench_event_pass": bench["roofline"].get("event_pass")}
        json.dump(bench, open(os.path.join(os.path.dirname(__file__), f"{tag}_bench_n1.json"), "w"), indent=1)
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d, f in (("prof_fetch", "f"), ("prof_write", "w"), ("prof_tcc", "t")):
    p = os.path.join(src, d, f"{f}_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        for k in HEADLINE:
            if k in r["Kernel_Name"]:
                pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
summary = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in pmc.items()}
for s in summary.values():
    if "FETCH_SIZE" in s:
        # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a
        # wide coalesced streaming read -> doubled.
        s["fetch_bytes_raw"] = s["FETCH_SIZE"] * 1024
        s["fetch_bytes_gfx950_corrected"] = 2 * s["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in s:
        s["write_bytes"] = s["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in s:
        s["l2_hit_rate"] = s["TCC_HIT_sum"] / (s["TCC_HIT_sum"] + s["TCC_MISS_sum"])
    if "TCC_EA0_RDREQ_sum" in s:
        r32 = s.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        s["ea_read_bytes_if_rest_are_128B"] = r32 * 32 + (s["TCC_EA0_RDREQ_sum"] - r32) * 128
        s["ea_read_bytes_if_rest_are_64B"] = r32 * 32 + (s["TCC_EA0_RDREQ_sum"] - r32) * 64
out["pmc_per_launch"] = summary
json.dump(out, open(os.path.join(os.path.dirname(__file__), f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
