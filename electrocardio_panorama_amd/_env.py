"""Environment switches of the package, in three groups (README.md lists them):

PRODUCT      honoured always: they choose between supported paths of the product.
DIAGNOSTICS  A/B forms and timing experiments of the kernel work: honoured ONLY under NEF_DIAG=1, otherwise ignored with one warning
             -- a stray variable cannot change which kernel a production run takes.  (The C side reads its own diagnostics switches
             the same way: csrc/nef_common.h, nef_diag_env.)
TEST HOOKS   two ranks on one GPU, gradients over gloo, ...: honoured only under NEF_TEST_HOOKS=1 (parallel._hook).
"""
import os
import sys

PRODUCT = {
    "NEF_H2": "1 (default): K = 1 / 3 / 7 convolutions on the split-fp16 kernels; 0: the fp32 Winograd / direct kernels (strict fp32, "
              "required for data whose dynamic range exceeds ops.H2_HEADROOM per step)",
    "NEF_H2_ALLOW_CLAMP": "1: Solver warns instead of raising when a split-fp16 launch clamped outside a protected train step",
    "NEF_WINOGRAD": "fp32 path: 4 (default) F(4,.) forms where allowed, 2 / 1: F(2,.) only, 0: direct kernels only",
    "NEF_SIDE_STREAM": "auto (default) / 1 / 0: weight gradients on a second HIP stream",
    "NEF_SOLVER_GRAPH": "0: Solver never replays the captured step",
    "NEF_GRAPH_SPLIT": "0: data-parallel graphed step as ONE graph + one exposed all-reduce (default: two graphs, early bucket between them)",
    "NEF_EARLY_REDUCE": "0: eager data-parallel step reduces one bucket at the optimiser step (default: early bucket under the encoder's backward)",
    "NEF_LIB": "path of an alternative libnefnet_hip.so (A/B builds: csrc/build.py build_variant)",
    "NEF_BENCH_DUMP": "bench.py: file for the per-launch event times of the breakdown steps",
    "NEF_TIE_LOG": "tests: file for the tie ratios of the decision-replaying tests",
    "NEF_DIAG": "1: honour the diagnostics switches",
    "NEF_TEST_HOOKS": "1: honour the test hooks (NEF_SHARE_GPU, NEF_DIST_BACKEND, NEF_DIST_FORCE)",
}
DIAGNOSTICS = {
    "NEF_FUSE_L2", "NEF_FUSE_STATS", "NEF_BNB_UP", "NEF_POLY", "NEF_POLY_FWD", "NEF_POLY_W", "NEF_PANO_L4_WIDE", "NEF_FOLD_CHSCALE", "NEF_FOLD_CHSCALE_BWD", "NEF_BWD_F4", "NEF_GRAPH_SIDE", "NEF_PANO_FUSE_PAIR", "NEF_PANO_FUSE_TAIL", "NEF_BW_WINO4", "NEF_BW7_F42",
    "NEF_H2_FWD", "NEF_H2_BWD", "NEF_H2_K", "NEF_H2_64", "NEF_H2_MIN_T", "NEF_H2_W", "NEF_H2_WK", "NEF_H2_AMAX", "NEF_H2_PACK",
    "NEF_H2_MIN_WGS",
    # read by the C side (nef_diag_env): listed for the README
    "NEF_H2_TM1", "NEF_H2W_V", "NEF_H2W_64", "NEF_H2W_ROUNDS", "NEF_H2P", "NEF_H2P_WGS", "NEF_BWW_GLDS", "NEF_DEBUG_LDS",
}
_warned = set()


def get(name, default=None):
    """Value of a PRODUCT switch, or of a DIAGNOSTICS switch when NEF_DIAG=1 (else `default`, with one warning if it is set)."""
    v = os.environ.get(name)
    if v is None:
        return default
    if name in PRODUCT:
        return v
    if name in DIAGNOSTICS:
        if os.environ.get("NEF_DIAG") == "1":
            return v
        if name not in _warned:
            _warned.add(name)
            sys.stderr.write(f"[nefnet] {name}={v} ignored: diagnostics switches need NEF_DIAG=1\n")
        return default
    raise KeyError(f"{name}: not a switch of this package (electrocardio_panorama_amd/_env.py)")
