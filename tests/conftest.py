import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "opt_in_experiment: exercises a code path that is OFF by default (an environment switch turns it on) "
                                       "and that no driver-run GPU suite has executed yet; such tests run LAST")


def pytest_collection_modifyitems(config, items):
    """Tests of opt-in experiments (k_render_w5 behind MGPU_W5=1, the enqueue threads behind MGPU_FRAME_ENQUEUE_THREADS=1) were written
    while GPU access was closed and have only ever run on the tests' emulator.  The suite is run with -x: they are moved behind every
    test of the default path, so that a failure in an experiment cannot hide the product's parity results.  Nothing is deselected."""
    last = [it for it in items if it.get_closest_marker("opt_in_experiment")]
    if last:
        items[:] = [it for it in items if not it.get_closest_marker("opt_in_experiment")] + last


# ---- image-parity report: what the parity suite SAW, not just that it passed --------------------------------------------
# tests/test_gpu_parity.py::assert_images_match books every image comparison here: byte-equal, or -- only with
# MALLIE_STRICT_PARITY=0 -- inside north_star's 1e-4 per-pixel L2.  The default is strict: an image that is not byte-equal
# to the oracle's / the reference's FAILS, whatever its distance.  The summary line goes to the terminal report (and so into
# the driver's GPUTEST record) and each test carries its own comparisons as a junit property.
PARITY = {"byte_equal": 0, "within_tolerance": 0, "notes": []}


def strict_parity():
    return os.environ.get("MALLIE_STRICT_PARITY", "1") != "0"


@pytest.fixture(autouse=True)
def _image_parity_property(record_property):
    before = (PARITY["byte_equal"], PARITY["within_tolerance"])
    yield
    eq, tol = PARITY["byte_equal"] - before[0], PARITY["within_tolerance"] - before[1]
    if eq or tol:
        record_property("image_parity", "%d byte-equal, %d within 1e-4 only" % (eq, tol))


def pytest_terminal_summary(terminalreporter):
    if PARITY["byte_equal"] or PARITY["within_tolerance"]:
        terminalreporter.write_line("image parity (%s): %d image comparisons byte-equal, %d inside the 1e-4 tolerance only%s" % (
            "strict: anything but byte-equal fails" if strict_parity() else "MALLIE_STRICT_PARITY=0",
            PARITY["byte_equal"], PARITY["within_tolerance"],
            "".join("\n  " + n for n in PARITY["notes"])))


# ---- the parity suite on the tests' wave emulator (tests/emu; MALLIE_MGPU_LIB = .../libmallie_mgpu_emu*.so, MALLIE_ALLOW_EMULATOR=1) ----
# There "device memory" is host memory and a launch has run when the call returns: torch's CPU tensors serve as the device buffers the
# tests hand to the C ABI.  The substitutions below exist in that mode only; on a GPU box nothing here runs.
def _on_emulator():
    return "_emu" in os.path.basename(os.environ.get("MALLIE_MGPU_LIB", "")) and os.environ.get("MALLIE_ALLOW_EMULATOR") == "1"


@pytest.fixture(autouse=True)
def _emulator_device_buffers(monkeypatch):
    if not _on_emulator():
        yield
        return
    import torch

    def host(fn):
        def wrapped(*a, **k):
            if "device" in k and k["device"] is not None and str(k["device"]).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return wrapped
    for name in ("empty", "full", "zeros", "ones", "tensor", "empty_like", "zeros_like", "full_like"):
        monkeypatch.setattr(torch, name, host(getattr(torch, name)))

    class _Stream:
        cuda_stream = 0

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def synchronize(self):
            pass

        def wait_stream(self, other):
            pass
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, "Stream", lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: s)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    yield
