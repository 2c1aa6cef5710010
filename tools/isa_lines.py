#!/usr/bin/env python3
"""Static instruction mix of ONE kernel by SOURCE LINE: hipcc -S -gline-tables-only keeps a `.loc file line col` directive in front of the
instructions every source line produced (inlined callees keep THEIR file and line), so the device ISA can be attributed to the lines
of mallie_amd/csrc/*.hip / *.hpp that made it.  Classes as rocprofv3's SQ_INSTS_* count them: VALU (v_*), SALU (s_* but waitcnt / nop /
branches), branches, LDS (ds_*), VMEM (global_ / buffer_ / flat_ / scratch_), SMEM (s_load / s_buffer_load).
usage: python tools/isa_lines.py <file.hip> '<demangled kernel substring>' [--top N] [--ranges a-b:name,c-d:name ...] [-D...]
  --ranges: line ranges of the MAIN file to sum under a name (e.g. 372-451:NODE,452-604:TRI,605-1043:SHADE); lines of other files are
  summed per file and function-sized range automatically (per 'file:line' otherwise)."""
import collections, os, re, subprocess, sys, tempfile

def classify(m):
    if m.startswith("v_"): return "valu"
    if m.startswith("ds_"): return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if m.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache")): return "smem"
    if m.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call", "s_endpgm")): return "branch"
    if m.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")): return "wait"
    if m.startswith("s_"): return "salu"
    return "other"

def main():
    a = sys.argv[1:]
    src, kern = a[0], a[1]
    top = int(a[a.index("--top") + 1]) if "--top" in a else 40
    ranges = []
    if "--ranges" in a:
        for r in a[a.index("--ranges") + 1].split(","):
            lohi, name = r.split(":")
            lo, hi = lohi.split("-")
            ranges.append((int(lo), int(hi), name))
    defs = [x for x in a if x.startswith("-D")]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-gline-tables-only",
                        src, "-o", out] + defs, check=True, capture_output=True)
        lines = open(out).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m: files[int(m.group(1))] = os.path.basename(m.group(3))
        else:
            m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"', l)
            if m: files.setdefault(int(m.group(1)), os.path.basename(m.group(2)))
    # the kernel: first label whose demangled name contains `kern`
    labels = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    dem = subprocess.run(["c++filt"] + [n for _, n in labels], capture_output=True, text=True).stdout.splitlines()
    start = next(i for (i, _), d in zip(labels, dem) if kern in d)
    name = next(d for (i, _), d in zip(labels, dem) if i == start)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    cur = (0, 0)
    by = collections.defaultdict(collections.Counter)
    tot = collections.Counter()
    main_file = os.path.basename(src)
    for i in range(start, end + 1):
        t = lines[i].strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        c = classify(t.split()[0])
        f = files.get(cur[0], "?")
        key = "%s:%d" % (f, cur[1])
        if f == main_file:
            for lo, hi, nm in ranges:
                if lo <= cur[1] <= hi:
                    key = nm + " (own lines)"
                    break
        by[key][c] += 1
        tot[c] += 1
    print(name.split("(")[0])
    print("total: " + "  ".join("%s %d" % (k, tot[k]) for k in ("valu", "salu", "branch", "lds", "vmem", "smem", "wait")))
    rows = sorted(by.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["salu"]))
    print("%-44s %6s %6s %6s %5s %5s" % ("source", "valu", "salu", "branch", "lds", "vmem"))
    for k, c in rows[:top]:
        print("%-44s %6d %6d %6d %5d %5d" % (k, c["valu"], c["salu"], c["branch"], c["lds"], c["vmem"]))

if __name__ == "__main__":
    main()
