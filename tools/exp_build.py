"""Build A/B variants of conv_mfma.hip in parallel:  python tools/exp_build.py name=DEF1,DEF2 name2=DEF ...
-> electrocardio_panorama_amd/csrc/variants/lib<name>.so (select at run time with NEF_LIB=<path>)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from electrocardio_panorama_amd.csrc import build as b  # noqa: E402

b.build(force=False, verbose=True)
jobs = []
for arg in sys.argv[1:]:
    name, _, defs = arg.partition("=")
    src = "conv_mfma.hip"
    if ":" in name:
        src, name = name.split(":")
    jobs.append((name, [d for d in defs.split(",") if d], src))
with ThreadPoolExecutor(4) as ex:
    for lib in ex.map(lambda j: b.build_variant(j[0], j[1], sources=(j[2],)), jobs):
        print(lib)
