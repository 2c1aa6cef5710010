#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: builds tests/emu/libmallie_mgpu_emu.so -- the library's own sources (mallie_amd/csrc: kernels, C ABI, host side)
compiled as plain C++ for the host against the wave64 emulator (tests/emu/include/hip/hip_runtime.h, emu_runtime.cc).  Loaded by tests
only, through MALLIE_MGPU_LIB; never by the product.   usage: python tests/emu/build_emu.py [--force] [extra -D flags]"""
import os, subprocess, sys, concurrent.futures as cf
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from mallie_amd import build as b
OUT = os.path.join(HERE, "libmallie_mgpu_emu.so")
OBJ = os.path.join(HERE, "_obj")
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-std=c++17", "-O1", "-g", "-fPIC", "-ffp-contract=off", "-fno-strict-aliasing", "-pthread", "-I", os.path.join(HERE, "include"),
         "-Wno-unknown-attributes", "-Wno-ignored-attributes", "-Wno-unused-value", "-Wno-macro-redefined", "-Wno-deprecated-declarations"]


def stale(extra):
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(b.CSRC, f) for f in b.SOURCES + b.HEADERS] + [os.path.join(HERE, "emu_runtime.cc"), os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, extra=()):
    if not force and not extra and not stale(extra):
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    srcs = [os.path.join(b.CSRC, f) for f in b.SOURCES] + [os.path.join(HERE, "emu_runtime.cc")]

    def one(src):
        obj = os.path.join(OBJ, os.path.basename(src).replace(".", "_") + ".o")
        cmd = [CXX] + FLAGS + list(extra) + ["-x", "c++", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emulator build failed: %s\n%s" % (" ".join(cmd), r.stderr[-6000:]))
        return obj
    with cf.ThreadPoolExecutor(6) as ex:
        objs = list(ex.map(one, srcs))
    r = subprocess.run([CXX, "-shared", "-pthread", "-o", OUT] + [e for e in extra if e.startswith("-fsanitize") or e in ("-shared-libasan", "-shared-libsan")] + objs + ["-ldl"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("emulator link failed:\n" + r.stderr[-4000:])
    return OUT


def build_asan():
    """The same library under AddressSanitizer + UndefinedBehaviorSanitizer (device code included: every kernel's loads and stores are
    checked against the hipMalloc'ed blocks they belong to -- what compute-sanitizer's memcheck would do).  -> (library, runtime to preload)"""
    global OUT, OBJ
    keep = OUT, OBJ
    OUT, OBJ = os.path.join(HERE, "libmallie_mgpu_emu_asan.so"), os.path.join(HERE, "_obj_asan")
    try:
        lib = build(True, ["-fsanitize=address,undefined", "-fno-sanitize=alignment,vptr,function", "-fno-sanitize-recover=undefined", "-shared-libasan", "-fno-omit-frame-pointer"])
    finally:
        OUT, OBJ = keep
    rt = subprocess.run([CXX, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    return lib, rt


def build_tsan():
    """... and under ThreadSanitizer: the host side's threads (enqueue pool, submission queue, render-ahead) with the kernels running
    underneath as fibers.  -> (library, runtime to preload)"""
    global OUT, OBJ
    keep = OUT, OBJ
    OUT, OBJ = os.path.join(HERE, "libmallie_mgpu_emu_tsan.so"), os.path.join(HERE, "_obj_tsan")
    try:
        lib = build(True, ["-fsanitize=thread", "-shared-libsan", "-fno-omit-frame-pointer"])
    finally:
        OUT, OBJ = keep
    rt = subprocess.run([CXX, "-print-file-name=libclang_rt.tsan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    return lib, rt


if __name__ == "__main__":
    args = sys.argv[1:]
    if args[:1] == ["tsan"]:
        print(*build_tsan())
    elif args[:1] == ["asan"]:
        print(*build_asan())
    else:
        print(build("--force" in args, [a for a in args if a != "--force"]))
