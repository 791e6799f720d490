import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    g = load_golden(request.param)
    g["name"] = request.param
    return g


@pytest.fixture(scope="session", autouse=True)
def _fresh_library():
    """Rebuild librgcn_mi355x.so when a source is newer (hipcc cross-compiles without a GPU), so a test
    run never exercises a stale binary."""
    import __graft_entry__ as g
    g.build()


def pytest_terminal_summary(terminalreporter):
    """How much of the slack over a flat 1e-5 / 1e-5 the parity checks used (oracle/tolerance.py bound (2))."""
    from oracle.tolerance import SLACK_LOG
    if not SLACK_LOG:
        return
    over = sorted((e for e in SLACK_LOG if e[1] > 0), key=lambda e: -e[1])
    tr = terminalreporter
    tr.write_line(f"parity: {len(SLACK_LOG)} tensor checks, {len(over)} used slack over flat 1e-5 + 1e-5|ref|")
    for what, excess, cpu_err in over[:12]:
        c = "n/a" if cpu_err is None else f"{cpu_err:.2e}"
        tr.write_line(f"   {what:40s} worst excess over flat {excess:.2e}   fp32 CPU loop's own worst error {c}")
