"""the CPU restatement (oracle/mallie_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer.
`make -C oracle asan` builds oracle/libmallie_oracle_asan.so; the golden pins of tests/test_oracle_golden.py (29 reference-generated
fixtures: camera, BVH builds, 7 250 trace records, renders in the reference's own stream, AOVs, panoramas) then run against it in
a child interpreter with the sanitizer runtimes preloaded.  Any report fails the test (halt_on_error, -fno-sanitize-recover)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_golden_pins_under_asan_and_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, capture_output=True)
    lib = os.path.join(ROOT, "oracle", "libmallie_oracle_asan.so")
    rts = [subprocess.run(["gcc", "-print-file-name=" + n], check=True, capture_output=True, text=True).stdout.strip()
           for n in ("libasan.so", "libubsan.so")]
    assert all(os.path.isabs(r) and os.path.exists(r) for r in rts), rts
    env = dict(os.environ, MALLIE_ORACLE_LIB=lib, LD_PRELOAD=":".join(rts),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, cwd=ROOT)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-3000:]
    assert r.returncode == 0 and " passed" in out, out[-3000:]
    # the child really loaded the sanitized library: its symbols are there
    nm = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
    assert "__asan_init" in nm and "__ubsan_handle" in nm
