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
traced_kernel_avg_ms = None
if ks:
    lines.append("== rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras ==")
    lines.append("%-78s %8s %12s %12s %12s %7s" % ("kernel", "calls", "avg us", "min us", "max us", "%"))
    for r in csv.DictReader(open(ks[0])):
        if "k_render_sm" in r["Name"] and traced_kernel_avg_ms is None:
            traced_kernel_avg_ms = float(r["AverageNs"]) / 1e6  # the dominant kernel's average in the traced bench run (bench.py: roofline.frac)
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
# ---- calibration of FETCH_SIZE / WRITE_SIZE on known byte counts -----------------------------------------------------------
def counter_sum(dirname, kernel_substr, counter):
    tot, disp = 0.0, set()
    for f in glob.glob(os.path.join(src, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
                tot += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    return tot, len(disp)


GIB = float(1 << 30)
calib = {}
for kern, ctr, d, known, what in (
        ("k_calib_read16", "FETCH_SIZE", "calib_fetch", GIB, "1 GiB streamed, 16 bytes per lane, coalesced"),
        ("k_calib_write16", "WRITE_SIZE", "calib_write", GIB, "1 GiB written, 16 bytes per lane, coalesced"),
        ("k_calib_write12", "WRITE_SIZE", "calib_write", 0.75 * GIB, "0.75 GiB written, three dwords per lane at 12-byte pitch"),
        # (cornellbox_suzanne is a scene of grey materials: its planes hold one float per pixel and pass -- mgpu_api.hip, mono_planes)
        ("k_accumulate_tiled", "FETCH_SIZE", "acc_fetch", 16 * 4.0 * 1920 * 1080, "k_accumulate_tiled_mono of a C2 frame: 16 planes of 8.3 MB read"),
        ("k_accumulate_tiled", "WRITE_SIZE", "acc_write", 12.0 * 1920 * 1080, "k_accumulate_tiled_mono of a C2 frame: the 24.9 MB image written")):
    v, n = counter_sum(d, kern, ctr)
    if n:
        per = v / n * 1024.0  # KiB per launch -> bytes
        calib["%s:%s" % (kern, ctr)] = {"known_bytes": known, "counter_bytes": per, "bytes_per_counted_byte": known / per if per else None,
                                        "launches": n, "what": what}
if calib:
    out["calibration"] = calib
    lines.append("")
    lines.append("== calibration: known bytes / bytes the counter reports (counter value x 1024) ==")
    for k, c in sorted(calib.items()):
        lines.append("    %-34s known %.4g B, counted %.4g B -> x %.3f   (%s; %d launches)" % (k, c["known_bytes"], c["counter_bytes"],
                                                                                             c["bytes_per_counted_byte"] or 0.0, c["what"], c["launches"]))
lines.append("")
lines.append("== PMC passes (per FRAME: sum over the render kernel's launches of a frame; library sha256 %s) ==" % sha)
for w, e in sorted(out["workloads"].items()):
    lines.append("%s  %s" % (w, e["kernel"]))
    for c, v in sorted(e.items()):
        if c != "kernel" and not c.startswith("launches_") and c != "launches_per_frame":
            lines.append("    %-26s %.6g   (%d launches)" % (c, v, e["launches_" + c]))
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        lines.append("    HBM bytes per frame = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes) = %.4g   (%.2g launches per frame)" % (2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024, e["launches_per_frame"]))
if traced_kernel_avg_ms and "c2" in out["workloads"]:
    out["workloads"]["c2"]["traced_kernel_avg_ms"] = traced_kernel_avg_ms
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_current.json"), "w"), indent=1, sort_keys=True)
open(os.path.join(ROOT, "profiles", tag + "_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

# ---- DESIGN.md tables, generated: the full bench line of the same library (collect_pmc.sh runs it after the passes, so its
#      roofline objects carry these PMC numbers) -> profiles/<tag>_tables.md and the marked block of DESIGN.md ------------------
bf = os.path.join(src, "bench_full.json")
if os.path.exists(bf) and os.path.getsize(bf):
    open(os.path.join(ROOT, "profiles", tag + "_bench_full.json"), "w").write(open(bf).read())
    b = json.loads(open(bf).read())
    roof = b["roofline"]
    t = ["<!-- generated by profiles/summarize_pmc.py %s from gpurun_out/%s (rocprofv3 PMC passes + the bench line of the same library, sha256 %s); do not edit -->" % (tag, tag, sha[:16]), ""]
    t.append("| config (1 MI355X) | ms / frame | kernel ms / frame | Mrays/s | nodes + tris / ray | HBM bytes / frame (PMC) | HBM GB/s (of 8 TB/s) | VALU issue busy | wave-cycles waiting | VALU wave-insts / ray |")
    t.append("|---|---|---|---|---|---|---|---|---|---|")
    def row(name, ms, kms, val, npr, tpr, traffic, valu_busy, wait, ipr):
        gbs = traffic / (kms * 1e-3) / 1e9 if traffic and kms else None
        return "| %s | %.2f | %.2f | %.0f | %.1f + %.1f | %s | %s | %s | %s | %s |" % (
            name, ms, kms, val, npr, tpr, "%.3g" % traffic if traffic else "-", "%.0f (%.1f %%)" % (gbs, gbs / 80.0) if gbs else "-",
            "%.0f %%" % (100 * valu_busy) if valu_busy else "-", "%.0f %%" % (100 * wait) if wait else "-", "%.1f" % ipr if ipr else "-")
    w2 = out["workloads"].get("c2", {})
    wait2 = w2.get("SQ_WAIT_ANY", 0) / w2["SQ_WAVE_CYCLES"] if w2.get("SQ_WAVE_CYCLES") else None
    t.append(row("C2 cornellbox_suzanne 1080p 16 spp (BVH in LDS)", b["ms_per_step"], roof["kernel_avg_ms"], b["value"], b["config"]["nodes_per_ray"],
                 b["config"]["tris_per_ray"], roof.get("traffic"), (roof.get("valu") or {}).get("issue_busy"), wait2,
                 (roof.get("valu") or {}).get("wave_insts_per_ray")))
    for k, e in sorted(b.get("extra_configs", {}).items()):
        if "error" in e:
            continue
        r = e["roofline"]
        w = out["workloads"].get(k, {})
        ipr = w["SQ_INSTS_VALU"] / e["rays_per_frame"] if w.get("SQ_INSTS_VALU") else None
        t.append(row("%s %s" % (k.upper(), e["config"].split(",")[0]) + " " + e["config"].split(", ")[1] + " " + e["config"].split(", ")[2], e["ms_per_frame"],
                     e["kernel_avg_ms"], e["value"], e["nodes_per_ray"], e["tris_per_ray"], r.get("traffic"), r.get("valu_issue_busy"),
                     r.get("wave_cycles_waiting"), ipr))
    if calib:
        t.append("")
        t.append("Counter calibration (known bytes / counted bytes, `profiles/microbench/pmc_calib.hip` and `k_accumulate_tiled`): " + "; ".join(
            "%s x %.2f" % (k, c["bytes_per_counted_byte"]) for k, c in sorted(calib.items()) if c["bytes_per_counted_byte"]) + ".")
    occ = roof.get("lane_occupancy") or {}
    if occ:
        t.append("")
        t.append("C2 lane occupancy (instrumented pass of the bench run): NODE %.2f, TRI %.2f, SHADE %.2f, weighted %.2f; useful share of the issue peak %s." % (
            occ.get("node_frac") or 0, occ.get("tri_frac") or 0, occ.get("shade_frac") or 0, occ.get("weighted") or 0,
            "%.2f" % roof["useful_frac"] if roof.get("useful_frac") else "-"))
    # what a "frame" is in that line, and the CPU figures of the same run (so that DESIGN.md never quotes a stale one by hand)
    t.append("")
    fr_parts = ["`value` / `ms_per_step` = the frame of SURVEY 8(d), passes + one read-back taken by the caller under the next frame: %.3f ms (%.0f Mrays/s)" % (b["ms_per_step"], b["value"])]
    for key2, label in (("frame_resident_in_hbm", "the same frames left in HBM, one in flight (the headline of rounds 1-3)"),
                        ("frame_with_synchronous_readback", "with a synchronous read-back before the next frame starts"), ("tile_order_off", "without the cost-ordered hand-out")):
        if key2 in b and "ms_per_frame" in b[key2]:
            fr_parts.append("%s: %.3f ms" % (label, b[key2]["ms_per_frame"]))
    t.append("Frames of that run: " + "; ".join(fr_parts) + ".")
    cb, cr = b.get("cpu_baseline") or {}, b.get("cpu_reference") or {}
    if cb.get("value"):
        t.append("")
        t.append("CPU on the same host (%s): the oracle (`kind: port`) %.1f Mrays/s on %s threads (first frame byte-equal to the GPU's: %s)%s." % (
            ("%s, %s physical cores" % (cb.get("cpu_model", "host"), cb.get("physical_cores"))), cb["value"], cb.get("cores"), cb.get("gpu_frame_byte_equal"),
            "; Mallie's own OpenMP `Render()` (`kind: reference`, kMaxPathLength 16) %.2f Mrays/s on %s threads" % (cr["value"], cr.get("cores")) if cr.get("value") else ""))
    rs = b.get("reference_stream_1080p") or {}
    if rs.get("resolve_ms_per_pass"):
        t.append("")
        t.append("The reference's own random stream, one 1080p pass (`reference_stream_1080p`): resolved across the chip in %.1f ms per pass (first call, with the camera's "
                 "classification: %.1f ms; %d uncertain pixels per pass) against %.0f ms for the whole call with the one-workgroup kernel of round 3." % (
                     rs["resolve_ms_per_pass"], rs["resolve_ms_first_call"], rs["uncertain_pixels_per_pass"], rs["serial_kernel_call_ms"]))
    txt = "\n".join(t) + "\n"
    open(os.path.join(ROOT, "profiles", tag + "_tables.md"), "w").write(txt)
    dp = os.path.join(ROOT, "DESIGN.md")
    d = open(dp).read()
    a, z = "<!-- pmc-tables:begin -->", "<!-- pmc-tables:end -->"
    if a in d and z in d:
        d = d[:d.index(a) + len(a)] + "\n" + txt + d[d.index(z):]
        open(dp, "w").write(d)
        print("DESIGN.md: generated tables updated")
