#!/usr/bin/env python3
"""Is the DEVICE code of the working tree instruction for instruction that of an earlier commit?  Compiles every kernel file of both
trees with hipcc -S for gfx950 and compares the instruction streams (labels, symbol names and comments normalised away).
usage: python tools/isa_same_as.py <commit> [file.hip ...]     (round 5: 9fdbfd3, the last commit the GPU suite ran on)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
commit = sys.argv[1]
files = sys.argv[2:] or [f for f in sorted(os.listdir(os.path.join(ROOT, "mallie_amd", "csrc"))) if f.endswith(".hip")]


def stream(src):
    with tempfile.NamedTemporaryFile(suffix=".s") as t:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", t.name, src],
                           capture_output=True, text=True)
        if r.returncode:
            return None
        out = []
        for l in open(t.name):
            x = l.split(';')[0].strip()
            if not re.match(r'^(s_|v_|ds_|global_|buffer_|scratch_|flat_|exp|image_)\S*', x):
                continue
            x = re.sub(r'[A-Za-z_\.\$][\w\.\$]*@\w+(\+\d+)?', 'SYM', x)
            x = re.sub(r'\.LBB\d+_\d+', 'LBL', x)
            out.append(re.sub(r'_ZN?[\w\$\.]+', 'SYM', x))
        return out


with tempfile.TemporaryDirectory() as d:
    subprocess.run("git -C %s archive %s mallie_amd/csrc include | tar -x -C %s" % (ROOT, commit, d), shell=True, check=True)
    for f in files:
        old_src = os.path.join(d, "mallie_amd", "csrc", f)
        if not os.path.exists(old_src):
            print("%-26s new file" % f)
            continue
        a, b = stream(old_src), stream(os.path.join(ROOT, "mallie_amd", "csrc", f))
        print("%-26s %s" % (f, "does not compile" if a is None or b is None else ("SAME (%d instructions)" % len(a) if a == b else "DIFFERENT (%d -> %d instructions)" % (len(a), len(b)))))
