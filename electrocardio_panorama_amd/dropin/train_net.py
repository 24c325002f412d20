"""Top-level alias of `electrocardio_panorama_amd.train_net` (reference codes/train_net.py)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("electrocardio_panorama_amd.train_net")
