"""Phase timeline of conv_wino4_kernel workgroups (a -DNEF_TRACE build: tools/exp_build.py trace=NEF_TRACE, NEF_LIB=...).
Runs the decoder forward + backward once with the trace buffer armed per launch and prints, per traced launch, the mean
duration of each phase of a workgroup in shader-clock cycles: entry -> loads issued -> first tile staged -> per stage
(MFMA loop | store + barrier) -> epilogue -> stores drained; plus the workgroup lifetime and the launch's span."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

os.environ.setdefault("NEF_SIDE_STREAM", "0")
from electrocardio_panorama_amd import _lib, ops  # noqa: E402
from electrocardio_panorama_amd.ops import GV  # noqa: E402

L = _lib.load()
L.nef_debug_set_trace.argtypes = [C.c_void_p]
dev = torch.device("cuda")
SH = {  # name: (G, Cig, Cog, B, T, pro_mode, up)
    "c4 fwd 64->64 aff": (1, 64, 64, 768, 5000, 1),
    "c3 fwd 128->64 aff+up": (1, 128, 64, 768, 2500, 3),
    "c2 fwd 128->128 aff": (1, 128, 128, 768, 2500, 1),
    "c3 bwd 64->128": (1, 64, 128, 768, 5000, 0),
    "c4 bwd 64->64": (1, 64, 64, 768, 5000, 0),
}
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, (G, Cig, Cog, B, T, pm) in SH.items():
    if only and only not in name:
        continue
    x = torch.randn(B, G * Cig, T, device=dev)
    w = torch.randn(G * Cog, Cig, 3, device=dev) * 0.05
    T_out = 2 * T if pm & 2 else T
    wp = ops.pack_weight(w, G, T=T_out, f4=True)
    pa, pb = torch.rand(3, Cig, device=dev) + 0.5, torch.randn(3, Cig, device=dev) * 0.1
    pro = (pm, pa, pb, B // 3) if pm else None
    stats = ops.conv_stats_buffer(wp, B, G, Cog, T_out, dev) if pm else None
    fn = lambda: ops.conv(GV.dense(x, G), wp, Cog, 3, pro=pro, stats=stats)
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    nwg = 1 << 16
    buf = torch.zeros(nwg * 24, dtype=torch.int64, device=dev)
    assert L.nef_debug_set_trace(buf.data_ptr()) == 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    L.nef_debug_set_trace(None)
    t = buf.cpu().numpy().reshape(nwg, 24)
    t = t[t[:, 0] != 0]
    hw = t[:, 22]
    first = t[t[:, 23] < 768] if False else t
    print("   HW_ID.WAVE_ID histogram (all traced workgroups):", np.bincount((hw & 15).astype(np.int64), minlength=10).tolist())
    nst = Cig // 16
    base = t[:, 0].min()
    life = t[:, 21] - t[:, 0]
    print(f"== {name}: {s.elapsed_time(e) * 1e3:.0f} us, {len(t)} traced workgroups (every 16th), launch span {(t[:, 21].max() - base)} ticks "
          f"=> {(t[:, 21].max() - base) / (s.elapsed_time(e) * 1e3):.1f} ticks/us")
    cols = [("issue loads", 0, 1), ("first tile staged", 1, 2)]
    for q in range(nst):
        cols.append((f"stage {q} mfma loop", 2 + 2 * q, 3 + 2 * q))
        cols.append((f"stage {q} store+barrier", 3 + 2 * q, 4 + 2 * q))
    cols += [("epilogue", 2 + 2 * nst, 20), ("stores drained", 20, 21)]
    for nm, a, b in cols:
        d = (t[:, b] - t[:, a]).astype(np.float64)
        print(f"   {nm:26s} mean {d.mean():9.0f}  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}")
    print(f"   {'workgroup lifetime':26s} mean {life.mean():9.0f}  p10 {np.percentile(life, 10):9.0f}  p90 {np.percentile(life, 90):9.0f}")
