import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


REPORT = []       # one-line facts the parity tests want in the run's tail whatever the verbosity (`conftest.REPORT.append`)


def report(line):
    REPORT.append(str(line))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Printed after the result lines even under -q: which full-size batch really ran, its flat-gradient rel-L2 and the
    number of replayed ties, the 3-step SGD trajectory's worst parameter, ... (captured `print`s of passing tests are not)."""
    if REPORT:
        terminalreporter.write_line("---- parity facts ----")
        for line in REPORT:
            terminalreporter.write_line(line)


def pytest_sessionfinish(session, exitstatus):
    """NEF_TIE_LOG=<path>: dump the tie ratios the decision-replaying tests saw (tests/decisions.py: |pre-activation| / rms
    of every site with a flipped decision) -- the evidence behind decisions.TIE_REL (profiles/r03_tie_ratios.md)."""
    path = os.environ.get("NEF_TIE_LOG")
    if not path:
        return
    try:
        import json
        import decisions
        with open(path, "w") as fh:
            json.dump([[s, float(r), int(f), int(n)] for s, r, f, n in decisions.TIE_LOG], fh)
    except Exception as exc:       # diagnostics must never fail a run
        sys.stderr.write(f"NEF_TIE_LOG: {exc}\n")
