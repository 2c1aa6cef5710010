#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (profiles/collect_pmc.sh) into profiles/<tag>_summary.txt, profiles/<tag>_bench_line.json and
profiles/pmc_current.json.  usage: python profiles/summarize_pmc.py <tag>"""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", tag)
lines = []
# kernel stats of the bench run
ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if ks:
    lines.append("== rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras ==")
    lines.append("%-78s %8s %12s %12s %12s %7s" % ("kernel", "calls", "avg us", "min us", "max us", "%"))
    for r in csv.DictReader(open(ks[0])):
        lines.append("%-78s %8s %12.1f %12.1f %12.1f %7.2f" % (r["Name"].split("(")[0][-78:], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                            float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
bl = os.path.join(src, "bench_line.json")
if os.path.exists(bl) and os.path.getsize(bl):
    open(os.path.join(ROOT, "profiles", tag + "_bench_line.json"), "w").write(open(bl).read())
    lines.append("")
    lines.append("bench line of the traced run: " + open(bl).read().strip()[:400] + " ...")
sha = open(os.path.join(src, "so_sha256.txt")).read().strip()
sys.path.insert(0, ROOT)
from mallie_amd import build as _build
out = {"so_sha256": sha, "source_sha256": _build.source_digest(), "tag": tag, "workloads": {}}
for d in sorted(glob.glob(os.path.join(src, "c?_*"))):
    if not os.path.isdir(d):
        continue
    w = os.path.basename(d).split("_")[0]
    acc = collections.defaultdict(lambda: [0.0, set()])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_render_sm" not in r["Kernel_Name"]:
                continue
            a = acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
    FRAMES = 3  # tools/pmc_workload.py <workload> 3: frames rendered per pass of the collection
    for (k, c), (v, disp) in acc.items():
        e = out["workloads"].setdefault(w, {"kernel": k})
        # per FRAME (a frame whose pass planes exceed 1 GiB is rendered in several launches: C3's 64 passes take two)
        e[c] = v / FRAMES
        e["launches_" + c] = len(disp)
        e["launches_per_frame"] = len(disp) / FRAMES
lines.append("")
lines.append("== PMC passes (per FRAME: sum over the render kernel's launches of a frame; library sha256 %s) ==" % sha)
for w, e in sorted(out["workloads"].items()):
    lines.append("%s  %s" % (w, e["kernel"]))
    for c, v in sorted(e.items()):
        if c != "kernel" and not c.startswith("launches_") and c != "launches_per_frame":
            lines.append("    %-26s %.6g   (%d launches)" % (c, v, e["launches_" + c]))
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        lines.append("    HBM bytes per frame = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes) = %.4g   (%.2g launches per frame)" % (2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024, e["launches_per_frame"]))
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_current.json"), "w"), indent=1, sort_keys=True)
open(os.path.join(ROOT, "profiles", tag + "_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
