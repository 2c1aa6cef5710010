#!/usr/bin/env python3
"""Where a render kernel's instructions go, DYNAMICALLY, without a GPU: compiles the kernels' gfx950 ISA with line tables (hipcc -S
-gline-tables-only), runs a BASELINE configuration at reduced resolution through the tests' ISA interpreter (tools/isa_run.py,
tests/emu/isa_interp.cc) and sums the per-instruction execution counts by the SOURCE that produced them -- the body of the wave loop
(scheduler / NODE / TRI / SHADE parts, by line range of mgpu_render_sm.hip) for the kernel's own lines, the function for inlined code of
mgpu_device.hpp -- and by instruction class.  Counts are wave-instructions, as SQ_INSTS_VALU / SQ_INSTS_SALU count them; on C4 the
interpreter's totals per ray are within 1 % of round 4's hardware counters.
usage: python tools/isa_profile.py c2|c4|... [W H spp] [--src mgpu_render_sm.hip] [-D...] [--top N] [--keep file.tsv]"""
import collections, json, os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(R, "mallie_amd", "csrc")
# bodies of k_render_sm's wave loop by line range of mgpu_render_sm.hip (kept next to the source: tools/isa_profile.py --check-ranges prints the lines)
MARKERS = [("prologue", r"__global__ __launch_bounds__"), ("scheduler", r"^  for \(;;\) \{$"), ("NODE step", r"=+ NODE step =+"), ("TRI step", r"=+ TRI step =+"),
           ("SHADE: finish the ray", r"=+ SHADE step =+"), ("SHADE: path hand-out", r"---- \(2\) path hand-out"), ("SHADE: start path / arm ray", r"---- \(3\) next path / next traversal"),
           ("epilogue", r"---- counters: one atomic per wave and word")]


def ranges_of(path):
    lines = open(path).read().split("\n")
    out, cur = [], None
    for i, l in enumerate(lines, 1):
        for name, pat in MARKERS:
            if re.search(pat, l) and (not out or out[-1][0] != name) and all(o[0] != name for o in out):
                out.append((name, i))
    return [(name, lo, (out[k + 1][1] - 1 if k + 1 < len(out) else len(lines))) for k, (name, lo) in enumerate(out)]


def functions_of(path):
    """(first line, last line, name) of the functions / lambdas-free blocks of a header: a definition starts at a line matching a function head."""
    lines = open(path).read().split("\n")
    heads = []
    for i, l in enumerate(lines, 1):
        m = re.match(r"^(?:template <[^>]*>\s*)?(?:static )?(?:__host__ )?__device__ (?:__forceinline__ )?[\w:<> \*&]+?\b(\w+)\(", l) or re.match(r"^\s+__device__ __forceinline__ [\w:<> \*&]+?\b(\w+)\(", l)
        if m:
            heads.append((i, m.group(1)))
    return [(lo, (heads[k + 1][0] - 1 if k + 1 < len(heads) else len(lines)), name) for k, (lo, name) in enumerate(heads)]


def main():
    a = sys.argv[1:]
    key = a[0]
    nums = [x for x in a[1:] if x.isdigit()]
    W, H, spp = (int(nums[0]), int(nums[1]), int(nums[2])) if len(nums) >= 3 else (480, 270, 4)
    defs = [x for x in a if x.startswith("-D")]
    top = int(a[a.index("--top") + 1]) if "--top" in a else 45
    keep = a[a.index("--keep") + 1] if "--keep" in a else None
    srcs = ["mgpu_render_sm.hip", "mgpu_kernels.hip"]
    with tempfile.TemporaryDirectory() as tmp:
        dumps = []
        procs = []
        for s in srcs:
            out = os.path.join(tmp, s.replace(".hip", ".s"))
            dumps.append(out)
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-gline-tables-only",
                                           os.path.join(CSRC, s), "-o", out] + defs, stderr=subprocess.DEVNULL))
        for p in procs:
            if p.wait() != 0:
                raise SystemExit("hipcc -S failed")
        prof = keep or os.path.join(tmp, "profile.tsv")
        sys.path.insert(0, os.path.join(R, "tests", "emu"))
        import build_emu
        env = dict(os.environ, MGPU_EMU_ISA=":".join(dumps))
        env.setdefault("MALLIE_MGPU_LIB", build_emu.build())  # (the emulator library: built when missing or stale)
        if defs:
            sys.stderr.write("note: -D flags change the ISA only; the emulator library (host side, launch parameters) is the default build\n")
        r = subprocess.run([sys.executable, os.path.join(R, "tools", "isa_run.py"), key, str(W), str(H), str(spp), prof], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            raise SystemExit("isa_run failed:\n" + r.stdout[-2000:] + r.stderr[-3000:])
        run = json.loads(line[-1])
        rows = [l.rstrip("\n").split("\t") for l in open(prof)]
    rays = run["rays"]
    print(json.dumps(run))
    rng = ranges_of(os.path.join(CSRC, "mgpu_render_sm.hip"))
    fdev = functions_of(os.path.join(CSRC, "mgpu_device.hpp"))
    by_kernel = collections.defaultdict(lambda: collections.defaultdict(collections.Counter))
    for kname, sline, cls, cnt, loc, text in rows:
        f, _, ln = loc.rpartition(":")
        f, ln = os.path.basename(f), int(ln or 0)
        where = "%s:%d" % (f, ln)
        if f == "mgpu_render_sm.hip":
            where = next(("k_render_sm: " + n for n, lo, hi in rng if lo <= ln <= hi), where)
        elif f == "mgpu_device.hpp":
            where = next(("mgpu_device.hpp: %s()" % n for lo, hi, n in fdev if lo <= ln <= hi), where)
        elif f in ("mgpu_sincos.hpp", "mgpu_kernels.hpp"):
            where = f
        elif f.endswith(".h"):
            where = "compiler / OCML headers (%s)" % f
        by_kernel[kname][where][cls] += int(cnt)
    for kname, groups in sorted(by_kernel.items(), key=lambda kv: -sum(sum(c.values()) for c in kv[1].values())):
        tot = collections.Counter()
        for c in groups.values():
            tot.update(c)
        dem = subprocess.run(["c++filt", kname], capture_output=True, text=True).stdout.strip().split("(")[0]
        if tot["valu"] < 0.002 * sum(sum(sum(c.values()) for c in g.values()) for g in by_kernel.values()):
            continue
        print("\n%s" % dem)
        print("  per ray: VALU %.2f  SALU %.2f  branch %.2f  LDS %.2f  VMEM %.3f   (SALU / VALU %.3f)" % (tot["valu"] / rays, tot["salu"] / rays, tot["branch"] / rays, tot["lds"] / rays,
                                                                                                  tot["vmem"] / rays, tot["salu"] / max(tot["valu"], 1)))
        print("  %-52s %9s %7s %9s %9s %8s" % ("source", "VALU/ray", "share", "SALU/ray", "br/ray", "LDS/ray"))
        for where, c in sorted(groups.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["salu"]))[:top]:
            print("  %-52s %9.3f %6.1f%% %9.3f %9.3f %8.3f" % (where[:52], c["valu"] / rays, 100.0 * c["valu"] / max(tot["valu"], 1), c["salu"] / rays, c["branch"] / rays, c["lds"] / rays))


if __name__ == "__main__":
    if "--check-ranges" in sys.argv:
        for r in ranges_of(os.path.join(CSRC, "mgpu_render_sm.hip")):
            print(r)
        for f in functions_of(os.path.join(CSRC, "mgpu_device.hpp")):
            print(f)
    else:
        main()
