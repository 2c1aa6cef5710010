"""Builds libmallie_mgpu.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

Usage: python -m mallie_amd.build [--force]
The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmallie_mgpu.so")
LIB_OCC = os.path.join(HERE, "libmallie_mgpu_occ.so")
LIB_LITERAL = os.path.join(HERE, "libmallie_mgpu_literal.so")
SOURCES = ["mgpu_kernels.hip", "mgpu_render_sm.hip", "mgpu_render_w5.hip", "mgpu_trace_sm.hip", "mgpu_trace_server.hip", "mgpu_render_env.hip", "mgpu_bvh_build.hip", "mgpu_api.hip", "mgpu_frame.hip", "mgpu_stream.hip", "mgpu_render_f32.hip", "host/bvh_build.cc", "host/camera.cc", "host/scene_render.cc",
           "host/mesh_io.cc"]
HEADERS = ["mgpu_device.hpp", "mgpu_kernels.hpp", "mgpu_sincos.hpp", "mgpu_enqueue_pool.hpp", "host/mesh_io.hpp", os.path.join("..", "..", "include", "mgpu.h"), os.path.join("..", "..", "include", "mgpu_internal.h"),
           os.path.join("..", "..", "include", "mallie", "mallie_api.hpp")]
# -ffp-contract=off: the parity contract (no FMA contraction on device or host), see csrc/mgpu_device.hpp
# -ffile-prefix-map: __FILE__ (error messages) does not carry the checkout's path.  The binary still depends on that path
# (clang's per-file ids of internal device symbols); rebuilt in the same place it is bit-identical, and bench.py matches the
# committed PMC passes to it by its sha256 or, failing that, by source_digest() below
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function", "-ffile-prefix-map=%s=." % os.path.dirname(HERE)]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; this package has no CPU build)")


def source_digest():
    """sha256 over the library's sources, headers and compiler flags (paths excluded): identifies the kernels a binary was
    built from wherever it was built (the binary's own sha256 depends on the checkout path through clang's per-file ids)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        h.update(os.path.basename(f).encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(x for x in FLAGS if not x.startswith("-ffile-prefix-map")).encode())
    return h.hexdigest()


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(out_path, extra_flags):
    """A/B builds for kernel experiments: same sources, extra -D flags, separate output (load via MALLIE_MGPU_LIB)."""
    cmd = [hipcc()] + FLAGS + list(extra_flags) + ["-o", out_path] + [os.path.join(CSRC, f) for f in SOURCES]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building %s" % out_path)
    return out_path


def build(force=False, verbose=False):
    """libmallie_mgpu.so (the product) and, side by side, two builds of the same sources that only tests and bench.py load:
    libmallie_mgpu_occ.so (-DMGPU_OCC=1: the render kernel's active-lane accounting, bench.py's occupancy pass) and
    libmallie_mgpu_literal.so (-DMGPU_SAMPLE_MATH=0: SampleDiffuseIS through the literal acos / sincos calls of
    render.cc:325-333 instead of their algebraic reduction, checked against the oracle by tests/test_gpu_parity.py)."""
    variants = ((LIB_OCC, ["-DMGPU_OCC=1"]), (LIB_LITERAL, ["-DMGPU_SAMPLE_MATH=0"]))
    if not force and not is_stale() and all(os.path.exists(v) and os.path.getmtime(v) >= os.path.getmtime(LIB) for v, _ in variants):
        return LIB
    srcs = [os.path.join(CSRC, f) for f in SOURCES]
    procs = [(out, subprocess.Popen([hipcc()] + FLAGS + extra + ["-o", out] + srcs, stdout=subprocess.PIPE,
                                    stderr=subprocess.STDOUT, text=True))
             for out, extra in ((LIB + ".tmp", []),) + variants]
    for out, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(log)
            raise RuntimeError("hipcc failed building %s" % out)
        if verbose:
            sys.stderr.write(log)
    os.replace(LIB + ".tmp", LIB)  # the product library last: a variant is never older than it
    for v, _ in variants:
        os.utime(v, None)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
