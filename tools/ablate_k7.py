"""Timing-only ablation table for the three K=7 matrix kernels: where does the time between the
executed-MFMA rate and the 157.3 TFLOP/s fp32 matrix peak go?

    python tools/ablate_k7.py build      # here (CPU box): csrc/variants/libabl<N>.so for every NEF_ABL mask below
    python tools/ablate_k7.py run        # on the GPU box: one bench_conv.py process per variant, UN-profiled, random data

NEF_ABL bits (conv_mfma.hip): 1 = no weight/A (bwd-weight: no global) fetches in the main loop, 2 = no activation fetch +
LDS staging stores, 4 = no LDS fragment reads / transform VALU, 8 = no epilogue.  15 = the MFMA stream and its barriers.
The ablated builds compute garbage; only their durations mean anything."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MASKS = [0, 8, 1, 2, 4, 3, 7, 15]
LABEL = {0: "full kernel", 8: "no epilogue", 1: "no A / global fetch in loop", 2: "no X fetch + LDS staging", 4: "no LDS reads + transforms",
         3: "no fetches, no staging", 7: "MFMA + barriers + epilogue", 15: "MFMA stream + barriers only"}
# executed MFMA flops per launch of the [256, 384, 1250] K=7 grouped conv: algorithmic 220.2 GFLOP x executed/algorithmic
ALG = 2.0 * 256 * 384 * 1250 * 128 * 7
# (round 3, final forms: forward F(2,4)+F(2,3) executes 9/14, backward-data F(4,4)+F(4,3) 13/28 of the algorithmic multiplies; the weight
# gradient has its own builds -- NEF_GL_ABL in conv_bww_glds.hip, profiles/r03_glds_weight_gradient.md)
EXEC = {"fwd F(2,4)+F(2,3)": 9 / 14, "bwd-data F(4,4)+F(4,3)": 13 / 28}


def build():
    from electrocardio_panorama_amd.csrc import build as b
    b.build(force=False, verbose=True)
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(lambda m: b.build_variant(f"abl{m}", [f"NEF_ABL={m}"]), MASKS))


def bench(lib, f4):
    env = dict(os.environ, NEF_LIB=lib, F4=f4, ITERS="20", WARM="40", ONLY_WHAT="wino")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_conv.py"), "enc k7"], env=env, text=True,
                         capture_output=True)
    res = {}
    for line in out.stdout.splitlines():
        m = re.search(r"\s(wino|bwd_ww)\s+([0-9.]+) ms", line)
        if m:
            res[m.group(1)] = float(m.group(2))
    if not res:
        sys.stderr.write(out.stderr[-2000:])
    return res


def run():
    vdir = os.path.join(ROOT, "electrocardio_panorama_amd", "csrc", "variants")
    rows = {}
    for rep in range(2):            # two rounds, variants interleaved, so box drift shows up as a spread
        for m in MASKS:
            lib = os.path.join(vdir, f"libabl{m}.so")
            a = bench(lib, "0")
            b = bench(lib, "1")
            r = rows.setdefault(m, {k: [] for k in EXEC})
            r["fwd F(2,4)+F(2,3)"].append(a.get("wino"))
            r["bwd-data F(4,4)+F(4,3)"].append(b.get("wino"))
    print("| NEF_ABL | variant | " + " | ".join(f"{k}: ms (executed TFLOP/s, frac of 157.3)" for k in EXEC) + " |")
    print("|---|---|" + "---|" * len(EXEC))
    for m in MASKS:
        cells = []
        for k in EXEC:
            v = [x for x in rows[m][k] if x]
            if not v:
                cells.append("n/a")
                continue
            ms = min(v)
            tf = ALG * EXEC[k] / ms / 1e9
            cells.append(f"{' / '.join(f'{x:.3f}' for x in v)} ({tf:.1f}, {tf / 157.3:.3f})")
        print(f"| {m} | {LABEL[m]} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
