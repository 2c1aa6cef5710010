#!/usr/bin/env python3
"""rocprofv3 counter passes of the render kernel on one BASELINE configuration, and the parsing of what they leave behind.

Two users:
  * bench.py -- when profiles/pmc_current.json was collected on another build of the library than the one loaded (every source
    edit does that), it collects the passes ITSELF, after its timed region, so that `roofline` is never empty on a box that has
    rocprofv3;
  * profiles/summarize_pmc.py / the command line (`python tools/pmc_collect.py c2 [c4 ...] [--out DIR]`).

Every pass is a process of its own: `rocprofv3 --pmc <counters> --kernel-include-regex k_render -- python tools/pmc_workload.py <w> <frames>`,
counters never together with tracing, TCC counters (FETCH_SIZE, WRITE_SIZE) each in a pass of their own, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes; the kernel average comes from a separate `--kernel-trace --stats` pass of the
same command.  Numbers are per FRAME (summed over the render kernel's launches of a frame, averaged over the frames of a pass).
Measurement only: nothing here is on the product path."""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "k_render_"  # k_render_sm<...> (and k_render_w5 under MGPU_W5=1); the panoramic / AOV kernels are never launched by the workloads
PASSES = collections.OrderedDict((
    ("sq", "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"),
    ("fetch", "FETCH_SIZE"),
    ("write", "WRITE_SIZE"),
    ("sq2", "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE"),
))


def so_sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def parse_counters(dirs, frames, kernel=KERNEL):
    """{counter: per-frame value, launches_<counter>: n, launches_per_frame, kernel} from the *counter_collection.csv under `dirs`."""
    acc = collections.defaultdict(lambda: [0.0, set()])
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if kernel not in r["Kernel_Name"]:
                        continue
                    a = acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])]
                    a[0] += float(r["Counter_Value"])
                    a[1].add(r["Dispatch_Id"])
    e = {}
    for (k, c), (v, disp) in sorted(acc.items()):
        e["kernel"] = k
        e[c] = e.get(c, 0.0) + v / frames
        e["launches_" + c] = e.get("launches_" + c, 0) + len(disp)
        e["launches_per_frame"] = e["launches_" + c] / frames
    return e


def parse_kernel_avg_ms(d, kernel=KERNEL):
    """(average ms, calls) of the render kernel in a --kernel-trace --stats pass under `d`, or (None, 0)."""
    tot_ns, calls = 0.0, 0
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel in r["Name"]:
                    tot_ns += float(r["TotalDurationNs"])
                    calls += int(r["Calls"])
    return (tot_ns / calls / 1e6 if calls else None), calls


def rocprof():
    return shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)


def child_env():
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return env


def collect(workload, out_dir, frames=3, passes=("sq", "fetch", "write", "sq2"), trace=False, trace_frames=10, timeout=300, deadline=None, log=None):
    """Runs the passes (each bounded by `timeout`; none is started after `deadline`, a time.monotonic() value) and returns the
    per-frame entry -- {} when rocprofv3 is missing or nothing could be read.  `log` collects one line per pass."""
    log = log if log is not None else []
    exe = rocprof()
    if exe is None:
        log.append("no rocprofv3 on this box")
        return {}
    os.makedirs(out_dir, exist_ok=True)
    env = child_env()
    cmd_tail = [sys.executable, os.path.join(ROOT, "tools", "pmc_workload.py"), workload]
    dirs = []
    for name in passes:
        if deadline is not None and time.monotonic() > deadline:
            log.append("%s_%s: skipped (time budget of the collection spent)" % (workload, name))
            continue
        d = os.path.join(out_dir, "%s_%s" % (workload, name))
        shutil.rmtree(d, ignore_errors=True)
        t0 = time.monotonic()
        try:
            r = subprocess.run([exe, "--pmc"] + PASSES[name].split() + ["--kernel-include-regex", KERNEL, "--output-format", "csv", "-d", d, "-o", "p", "--"]
                               + cmd_tail + [str(frames)], cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            log.append("%s_%s: rc %d, %.1f s" % (workload, name, r.returncode, time.monotonic() - t0))
            with open(d + ".log", "w") as f:
                f.write(r.stdout[-20000:] + r.stderr[-20000:])
        except (OSError, subprocess.SubprocessError) as e:
            log.append("%s_%s: %r" % (workload, name, e))
            continue
        dirs.append(d)
    e = parse_counters(dirs, frames)
    if trace and not (deadline is not None and time.monotonic() > deadline):
        d = os.path.join(out_dir, "%s_trace" % workload)
        shutil.rmtree(d, ignore_errors=True)
        t0 = time.monotonic()
        try:
            r = subprocess.run([exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd_tail + [str(trace_frames)],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            log.append("%s_trace: rc %d, %.1f s" % (workload, r.returncode, time.monotonic() - t0))
            avg, calls = parse_kernel_avg_ms(d)
            if avg is not None and e:
                # per FRAME, like the counters: a frame whose pass planes exceed 1 GiB takes several launches
                e["traced_kernel_avg_ms"] = avg * calls / trace_frames
                e["traced_kernel_calls"] = calls
                e["traced_command"] = "rocprofv3 --kernel-trace --stats -- python tools/pmc_workload.py %s %d" % (workload, trace_frames)
        except (OSError, subprocess.SubprocessError) as ex:
            log.append("%s_trace: %r" % (workload, ex))
    return e


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="+", choices=("c2", "c3", "c4", "c5"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc_collect"))
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--trace", action="store_true")
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    lib = os.environ.get("MALLIE_MGPU_LIB") or os.path.join(ROOT, "mallie_amd", "libmallie_mgpu.so")
    from mallie_amd import build as _b
    out = {"so_sha256": so_sha256(lib), "source_sha256": _b.source_digest(), "tag": os.path.basename(a.out), "workloads": {}}
    log = []
    for w in a.workloads:
        e = collect(w, a.out, frames=a.frames, trace=a.trace, log=log)
        if e:
            out["workloads"][w] = e
    print("\n".join(log))
    with open(os.path.join(a.out, "pmc_current.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out["workloads"], indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
