"""Slot timeline of conv_w4r_kernel (-DNEF_TRACE build of conv_w4r.hip): per group, iteration 2 of every 8th workgroup."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("NEF_SIDE_STREAM", "0")
from electrocardio_panorama_amd import _lib, ops
from electrocardio_panorama_amd.ops import GV
L = _lib.load()
L.nef_debug_set_trace_w4r.argtypes = [C.c_void_p]
dev = torch.device("cuda")
SH = {"c4 fwd 64->64 aff": (1, 64, 64, 768, 5000, 1), "c2 fwd 128->128 aff": (1, 128, 128, 768, 2500, 1),
      "c3 bwd 64->128": (1, 64, 128, 768, 5000, 0)}
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, (G, Cig, Cog, B, T, pm) in SH.items():
    if only and only not in name:
        continue
    x = torch.randn(B, G * Cig, T, device=dev)
    w = torch.randn(G * Cog, Cig, 3, device=dev) * 0.05
    wp = ops.pack_weight(w, G, T=T, f4=True)
    pa, pb = torch.rand(3, Cig, device=dev) + 0.5, torch.randn(3, Cig, device=dev) * 0.1
    pro = (pm, pa, pb, B // 3) if pm else None
    stats = ops.conv_stats_buffer(wp, B, G, Cog, T, dev)
    fn = lambda: ops.conv(GV.dense(x, G), wp, Cog, 3, pro=pro, stats=stats)
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    buf = torch.zeros(32 * 3 * 32, dtype=torch.int64, device=dev)
    assert L.nef_debug_set_trace_w4r(buf.data_ptr()) == 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    L.nef_debug_set_trace_w4r(None)
    t = buf.cpu().numpy().reshape(-1, 32).astype(np.float64)
    t = t[t[:, 0] != 0]
    S = Cig // 16
    print(f"== {name}: {s.elapsed_time(e) * 1e3:.0f} us, {len(t)} traced (group, workgroup) pairs")
    def col(nm, a, b):
        d = t[:, b] - t[:, a]
        print(f"   {nm:34s} mean {d.mean():8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}")
    for j in range(S):
        col(f"stage {j} mfma loop", 0 if j == 0 else 3 * j, 1 + 3 * j)
        col(f"stage {j} store next", 1 + 3 * j, 2 + 3 * j)
        col(f"stage {j} barrier wait", 2 + 3 * j, 3 + 3 * j)
    col("NM0: decode+tables", 3 * S, 25)
    col("NM0: epilogue half 0 + issue", 25, 26)
    col("NM0..: barrier wait(s)", 26, 27)
    col("NM1: store first stage", 27, 28)
    col("NM1: epilogue half 1 + take over", 28, 29)
    col("NM1: barrier wait", 29, 30)
    col("whole period", 0, 30)
