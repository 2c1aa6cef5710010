"""The library's kernels on a CPU: tests/emu compiles mallie_amd/csrc as plain C++ against a wave64 emulator (every lane a fiber,
cross-lane operations evaluated over the whole wave, workgroup barriers, LDS) and tests/emu/cases_emu.py runs render, trace, stream and
multi-rank frame cases through the C ABI against the oracle -- the CPU / GPU differential SURVEY.md 5 asks for in place of
compute-sanitizer.  TEST INFRASTRUCTURE: the emulator build is loaded here and nowhere else (mallie_amd refuses it without
MALLIE_ALLOW_EMULATOR=1; the product has no CPU path)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernels_on_the_wave_emulator_match_the_oracle():
    if not os.path.isfile(os.path.join(ROOT, "tests", "emu", "build_emu.py")):
        pytest.skip("tests/emu is not on this box (it is kept off the GPU box: .gpurunignore)")
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    if not os.path.isfile(build_emu.CXX):
        pytest.skip("no %s here: the emulator build needs ROCm's clang++" % build_emu.CXX)
    lib = build_emu.build()
    env = dict(os.environ, MALLIE_MGPU_LIB=lib, MALLIE_ALLOW_EMULATOR="1", MALLIE_NO_TORCH="1")
    for k in [k for k in env if k.startswith("MGPU_")]:
        del env[k]
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "emu", "cases_emu.py"), "-q", "-x", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=1500)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]


def test_the_product_refuses_the_emulator_build():
    """mallie_amd has no CPU path: pointed at the emulator build without the tests' switch it raises."""
    env = dict(os.environ, MALLIE_MGPU_LIB=os.path.join(ROOT, "tests", "emu", "libmallie_mgpu_emu.so"), MALLIE_NO_TORCH="1")
    env.pop("MALLIE_ALLOW_EMULATOR", None)
    r = subprocess.run([sys.executable, "-c", "import mallie_amd as M; M.device_count()"], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "not the product library" in r.stderr
    # ... and nothing in the bench, the entry points or the package (but the loader's refusal itself) names the emulator build
    paths = ["bench.py", "__graft_entry__.py"] + [os.path.join("mallie_amd", f) for f in os.listdir(os.path.join(ROOT, "mallie_amd")) if f.endswith(".py") and f != "mgpu.py"]
    for path in paths:
        txt = open(os.path.join(ROOT, path)).read()
        assert "tests/emu" not in txt and "mgpu_emu" not in txt and "ALLOW_EMULATOR" not in txt, path


def test_kernels_run_from_hipcc_gfx950_machine_code_match_the_oracle(tmp_path):
    """tests/emu/isa_interp.cc: the emulator executes the gfx950 instruction streams `hipcc -S` makes of the kernels (render, five-wave
    render, batched trace, accumulate, tile order, scene layout) instead of their host-compiled C++ -- what the COMPILER made, OCML's inlined
    division / sqrt sequences, ds_bpermute shuffles and spills included -- and every case of tests/emu/cases_emu.py still equals the oracle."""
    if not os.path.isfile(os.path.join(ROOT, "tests", "emu", "build_emu.py")):
        pytest.skip("tests/emu is not on this box")
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.isfile(hipcc):
        pytest.skip("no hipcc here")
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    lib = build_emu.build()
    dumps, procs = [], []
    for src in ("mgpu_render_sm.hip", "mgpu_kernels.hip", "mgpu_render_w5.hip", "mgpu_trace_sm.hip"):
        out = str(tmp_path / src.replace(".hip", ".s"))
        dumps.append(out)
        procs.append(subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                                       os.path.join(ROOT, "mallie_amd", "csrc", src), "-o", out], stderr=subprocess.DEVNULL))
    assert all(p.wait() == 0 for p in procs)
    env = dict(os.environ, MALLIE_MGPU_LIB=lib, MALLIE_ALLOW_EMULATOR="1", MALLIE_NO_TORCH="1", MGPU_EMU_ISA=":".join(dumps))
    for k in [k for k in env if k.startswith("MGPU_") and k != "MGPU_EMU_ISA"]:
        del env[k]
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "emu", "cases_emu.py"), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "render_kernels or five_wave or batched_trace or gfx950"],
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=1500)
    assert r.returncode == 0 and " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-4000:] + r.stderr[-2000:]
