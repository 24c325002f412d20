"""Top-level alias of `electrocardio_panorama_amd.utils`: with this directory on sys.path the import lines of the
reference's entry scripts (`from utils import ...`, codes/main.py:1-9, train_net.py:1-8, solver/solver.py:10-13)
resolve to the MI355X build unchanged.  No code lives here."""
import importlib
import sys

_real = importlib.import_module("electrocardio_panorama_amd.utils")
for _alias, _target in (
        ("checkpointer", "checkpointer"),
        ("seed_torch", "seed_torch"),
        ("mertic", "metric"),
        ("metric", "metric"),
):
    sys.modules[__name__ + "." + _alias] = importlib.import_module("electrocardio_panorama_amd.utils." + _target)
sys.modules[__name__] = _real
