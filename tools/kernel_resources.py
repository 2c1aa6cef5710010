#!/usr/bin/env python3
"""Register / spill / LDS table of every kernel in a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py mallie_amd/csrc/mgpu_render_sm.hip [extra -D flags]"""
import re, subprocess, sys
src, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c",
       "-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if not rows:
    sys.stderr.write(err[-3000:])
    raise SystemExit("no kernels reported (compile error?)")
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.splitlines()
print("%-70s %5s %5s %7s %6s %6s %4s %7s" % ("kernel", "VGPR", "SGPR", "scratch", "vspill", "sspill", "occ", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n)
    print("%-70s %5s %5s %7s %6s %6s %4s %7s" % (n[-70:], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"),
          r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
