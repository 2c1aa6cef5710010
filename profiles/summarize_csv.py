#!/usr/bin/env python3
"""Summarises the raw rocprofv3 CSVs of profiles/collect_r1.sh (gpurun_out/<tag>/) into profiles/<tag>_summary.txt and
profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

HBM bytes per launch = 2 * FETCH_SIZE*1024 + WRITE_SIZE*1024: on gfx950 FETCH_SIZE reports half the bytes of wide
coalesced reads (/opt/skills/guides/MI355X_MICROARCH.md "HBM"; confirmed here on k_accumulate, whose known read volume is
passes*frame bytes); WRITE_SIZE is used as reported."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
here = os.path.dirname(os.path.abspath(__file__))


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(os.path.join(src, sub, "p_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


lines = []
lines.append("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline   (%s)" % tag)
with open(os.path.join(src, "trace", "p_kernel_stats.csv")) as f:
    rows = list(csv.DictReader(f))
lines.append("%-88s %6s %12s %12s %8s %12s %12s" % ("kernel", "calls", "total_ms", "avg_ms", "pct", "min_ms", "max_ms"))
for r in rows:
    lines.append("%-88s %6s %12.4f %12.4f %8s %12.4f %12.4f" % (r["Name"][:88], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                            float(r["AverageNs"]) / 1e6, r["Percentage"],
                                                            float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
lines.append("")
lines.append("# bench line of that run:")
lines.append(open(os.path.join(src, "bench_line.json")).read().strip())
lines.append("")
fetch, write, sq = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_sq")
traffic = {}
lines.append("# PMC (separate passes, mean per launch). FETCH_SIZE / WRITE_SIZE in KiB as reported; HBM bytes = 2*FETCH + WRITE.")
for k in sorted(fetch):
    if not k.startswith(("void mgpu", "mgpu")):
        continue
    fe, wr = fetch[k].get("FETCH_SIZE", 0.0), write.get(k, {}).get("WRITE_SIZE", 0.0)
    hbm = 2 * fe * 1024 + wr * 1024
    traffic[k.split("(")[0].replace("void ", "")] = int(hbm)
    lines.append("%-60s FETCH_SIZE %12.0f KiB  WRITE_SIZE %12.0f KiB  -> HBM %8.1f MB / launch" % (k[:60], fe, wr, hbm / 1e6))
lines.append("")
lines.append("# SQ counters, mean per launch")
for k in sorted(sq):
    if k.startswith(("void mgpu", "mgpu")):
        lines.append(k[:100])
        for c, v in sorted(sq[k].items()):
            lines.append("    %-24s %.4g" % (c, v))
open(os.path.join(here, "%s_summary.txt" % tag), "w").write("\n".join(lines) + "\n")
sq_out = {k.split("(")[0].replace("void ", ""): {c: v for c, v in d.items()} for k, d in sq.items() if k.startswith(("void mgpu", "mgpu"))}
json.dump({"tag": tag, "bytes_per_launch": traffic, "sq_per_launch": sq_out,
           "method": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024, separate rocprofv3 --pmc passes (see %s_summary.txt)" % tag},
          open(os.path.join(here, "hbm_traffic.json"), "w"), indent=1)
print("\n".join(lines))
