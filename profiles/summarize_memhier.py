#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/pmc_memhier.sh) -> profiles/<tag>_memhier.txt and .json: the render kernel's memory-hierarchy counters per
FRAME (tools/pmc_workload.py renders 3) and what they say about where the HBM-resident walk's fetches are served.
usage: python profiles/summarize_memhier.py <tag>"""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", tag)
FRAMES = 3
out = {"tag": tag, "so_sha256": open(os.path.join(src, "so_sha256.txt")).read().strip() if os.path.exists(os.path.join(src, "so_sha256.txt")) else None,
       "workloads": {}}
for d in sorted(glob.glob(os.path.join(src, "c?_*"))):
    if not os.path.isdir(d):
        continue
    w = os.path.basename(d).split("_")[0]
    acc = collections.defaultdict(float)
    kern = None
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_render_sm" not in r["Kernel_Name"]:
                continue
            kern = r["Kernel_Name"].split("(")[0]
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    e = out["workloads"].setdefault(w, {})
    if kern:
        e["kernel"] = kern
    for c, v in acc.items():
        e[c] = v / FRAMES
lines = ["== memory hierarchy of the render kernel, per frame (rocprofv3 --pmc, separate passes; library sha256 %s) ==" % (out["so_sha256"] or "?")[:140]]
for w, e in sorted(out["workloads"].items()):
    g = lambda k: e.get(k)
    lines.append("%s  %s" % (w, e.get("kernel", "?")))
    for k in sorted(e):
        if k != "kernel" and k != "derived":
            lines.append("    %-42s %.6g" % (k, e[k]))
    d = {}
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
        d["l1_hit_rate"] = 1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")
    if g("TCP_TCC_READ_REQ_sum") and g("TCP_TCC_READ_REQ_LATENCY_sum"):
        d["l1_miss_latency_cycles"] = g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
        d["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("TCC_EA0_RDREQ_sum") and g("TCC_EA0_RDREQ_DRAM_sum") is not None:
        d["fabric_reads_to_dram_share"] = g("TCC_EA0_RDREQ_DRAM_sum") / g("TCC_EA0_RDREQ_sum")
        if g("TCC_EA0_RDREQ_32B_sum") is not None:
            # The counter tallies a request at 32 or 64 bytes; on gfx950 the requests of wide reads are 128 bytes tallied at 64
            # (/opt/skills/guides/MI355X_MICROARCH.md, HBM: FETCH_SIZE = TCC_EA0_RDREQ x 64 B reports HALF the bytes; this repo's own
            # calibration, profiles/pmc_current.json "calibration": x 1.9997 on 1 GiB of 16-byte reads).  `fabric_read_bytes` carries that
            # factor 2 -- the same quantity as bench.py's 2 x FETCH_SIZE --, the uncorrected tally is kept beside it.
            d["fabric_read_bytes_as_tallied"] = 32.0 * g("TCC_EA0_RDREQ_32B_sum") + 64.0 * (g("TCC_EA0_RDREQ_sum") - g("TCC_EA0_RDREQ_32B_sum"))
            d["fabric_read_bytes"] = 2.0 * d["fabric_read_bytes_as_tallied"]
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") is not None:
        d["wave_cycles_waiting"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_INSTS_VMEM_RD") and g("TCP_TOTAL_CACHE_ACCESSES_sum"):
        d["l1_tag_accesses_per_vmem_read_inst"] = g("TCP_TOTAL_CACHE_ACCESSES_sum") / g("SQ_INSTS_VMEM_RD")
    e["derived"] = d
    for k, v in sorted(d.items()):
        lines.append("    -> %-39s %.4g" % (k, v))
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_memhier.json"), "w"), indent=1, sort_keys=True)
open(os.path.join(ROOT, "profiles", tag + "_memhier.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
