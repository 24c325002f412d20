"""One fp16 panorama-decoder layer in isolation (pano_h.hip).  usage: bench_hconv.py <layer 1..4 | 12> [pairs] [T_out]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from electrocardio_panorama_amd import ops

layer = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
if layer == 12:      # layers 1 + 2 in one pass (nef_pano_h_conv_pair), T <= 256
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    nq = 16
    g = torch.Generator(device="cuda").manual_seed(1)
    w1 = torch.randn(128, 256, 3, device="cuda", generator=g) * (2.0 / (3 * 256)) ** 0.5
    w2 = torch.randn(128, 128, 3, device="cuda", generator=g) * (2.0 / (3 * 128)) ** 0.5
    b1 = torch.randn(128, device="cuda", generator=g) * 0.1
    b2 = torch.randn(128, device="cuda", generator=g) * 0.1
    wp1, wp2 = ops.pano_h_pack_weight(w1), ops.pano_h_pack_weight(w2)
    x = torch.randn(N // nq, T // 2, 256, device="cuda", generator=g).half()
    sc = torch.randn(N // nq, nq, 256, device="cuda", generator=g)
    y = torch.empty(N, T, 128, device="cuda", dtype=torch.float16)
    run = lambda: ops.pano_h_conv_pair(x, wp1, b1, (sc, nq * 256, 256), wp2, b2, N, nq, nq, out=y)
    for _ in range(2): run()
    torch.cuda.synchronize()
    n = 5; s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    fl = 2.0 * N * T * 128 * (256 + 128) * 3
    by = 2.0 * (N * T * 128 + x.numel())
    print(f"layers 1+2 fused N={N} T={T}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TFLOP/s  {by/ms/1e6:.0f} GB/s")
    sys.exit(0)
cin, cout, up = {1: (256, 128, True), 2: (128, 128, False), 3: (128, 64, True), 4: (64, 64, False)}[layer]
T = int(sys.argv[3]) if len(sys.argv) > 3 else (256 if layer < 3 else 512)
nq = 16
g = torch.Generator(device="cuda").manual_seed(1)
w = torch.randn(cout, cin, 3, device="cuda", generator=g) * (2.0 / (3 * cin)) ** 0.5
bias = torch.randn(cout, device="cuda", generator=g) * 0.1
wp = ops.pano_h_pack_weight(w)
Tin = T // 2 if up else T
if layer == 1:
    x = (torch.randn(N // nq, Tin, cin, device="cuda", generator=g)).half()
    sc = torch.randn(N // nq, nq, cin, device="cuda", generator=g)
    run = lambda: ops.pano_h_conv(x, wp, bias, cout, N=N, upsample=True, scale=(sc, nq * cin, cin), x_div=nq, nq=nq, out=y)
else:
    x = (torch.randn(N, Tin, cin, device="cuda", generator=g)).half()
    run = lambda: ops.pano_h_conv(x, wp, bias, cout, upsample=up, out=y)
y = torch.empty(N, T, cout, device="cuda", dtype=torch.float16)
for _ in range(2): run()
torch.cuda.synchronize()
n = 5; s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(n): run()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
fl = 2.0 * N * T * cout * cin * 3
by = 2.0 * (N * T * cout + x.numel())
print(f"layer {layer} {cin}->{cout} N={N} T={T}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TFLOP/s  {by/ms/1e6:.0f} GB/s")
