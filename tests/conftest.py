import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
