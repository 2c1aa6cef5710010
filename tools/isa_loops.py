#!/usr/bin/env python3
"""Static instruction mix of ONE kernel by LOOP: hipcc annotates every basic block of its -S output with the loop it belongs to
(`in Loop: Header=BBx_y Depth=d`), so the device ISA can be summed per loop nest -- the wave loop of k_render_sm at depth 1, the NODE
repetition / TRI trip / hand-out loops at depth 2, the pop loop inside a NODE repetition at depth 3 -- and a loop's own blocks (without
its inner loops) give the instructions one trip can issue at most.  With -gline-tables-only the first source line of a header names it.
Classes as rocprofv3's SQ_INSTS_* count them (VALU, SALU without waitcnt / nop / branches, branch, LDS, VMEM, SMEM).
usage: python tools/isa_loops.py <file.hip> '<demangled kernel substring>' [-D...]"""
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


def analyse(src, kern, defs=()):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-gline-tables-only",
                        src, "-o", out] + list(defs), check=True, capture_output=True)
        lines = open(out).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l) or re.match(r'\s*\.file\s+(\d+)\s+()"([^"]*)"', l)
        if m: files.setdefault(int(m.group(1)), os.path.basename(m.group(3)))
    labels = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    dem = subprocess.run(["c++filt"] + [n for _, n in labels], capture_output=True, text=True).stdout.splitlines()
    start = next(i for (i, _), d in zip(labels, dem) if kern in d)
    name = next(d for (i, _), d in zip(labels, dem) if i == start)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    # blocks: a label line or a `; %bb.N:` marker starts one; its annotation names the innermost loop (or makes it a header)
    loops = {}          # header -> dict(depth, parent, own Counter, line, f64)
    order = []
    cur_loop, cur_loc = None, None
    stack = []          # enclosing headers by depth
    blk_first = False
    for i in range(start + 1, end + 1):
        l = lines[i]
        mb = re.match(r"^(\.LBB\d+_\d+):|^; %bb\.(\d+):", l)
        if mb:
            lab = mb.group(1) or ("bb." + mb.group(2))
            ann = l
            j = i + 1
            while j <= end and re.match(r"^\s+; (=>|  )", lines[j]):  # continuation lines of the annotation
                ann += lines[j]
                j += 1
            mh = re.search(r"=>\s*This (Inner )?Loop Header: Depth=(\d+)", ann)
            mi = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", ann)
            if mh:
                d = int(mh.group(2))
                parent = None
                mp = re.search(r"Parent Loop (BB\d+_\d+) Depth=%d" % (d - 1), ann)
                if mp: parent = "." + "L" + mp.group(1) if False else mp.group(1)
                key = lab.replace(".L", "")
                loops[key] = dict(depth=d, parent=parent, own=collections.Counter(), line=None, f64=0)
                order.append(key)
                cur_loop = key
                blk_first = True
            elif mi:
                cur_loop = mi.group(1)
                blk_first = False
            else:
                cur_loop = None
            continue
        t = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur_loc = "%s:%s" % (files.get(int(m.group(1)), "?"), m.group(2))
            continue
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        c = classify(t.split()[0])
        key = cur_loop if cur_loop in loops else None
        if key is None:
            loops.setdefault("(outside loops)", dict(depth=0, parent=None, own=collections.Counter(), line=None, f64=0))
            key = "(outside loops)"
            if key not in order: order.append(key)
        L = loops[key]
        L["own"][c] += 1
        L["f64"] += "_f64" in t
        if L["line"] is None and cur_loc: L["line"] = cur_loc
    return name, loops, order


def main():
    a = sys.argv[1:]
    name, loops, order = analyse(a[0], a[1], [x for x in a[2:] if x.startswith("-D")])
    print(name.split("(")[0])
    print("%-16s %-5s %-28s %6s %6s %6s %6s %5s %5s %5s   (a loop's OWN blocks: inner loops listed separately)" % ("loop header", "depth", "first source line", "valu", "f64", "salu", "branch", "lds", "vmem", "wait"))
    tot = collections.Counter()
    for k in order:
        L = loops[k]
        c = L["own"]
        tot.update(c)
        print("%-16s %-5d %-28s %6d %6d %6d %6d %5d %5d %5d" % ("  " * max(L["depth"] - 1, 0) + k, L["depth"], L["line"] or "-", c["valu"], L["f64"], c["salu"], c["branch"], c["lds"], c["vmem"], c["wait"]))
    print("total: " + "  ".join("%s %d" % (k, tot[k]) for k in ("valu", "salu", "branch", "lds", "vmem", "smem", "wait")))


if __name__ == "__main__":
    main()
