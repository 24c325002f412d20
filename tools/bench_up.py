import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from electrocardio_panorama_amd import ops
from electrocardio_panorama_amd.ops import GV
B, Cig, Cog, T = 768, 256, 128, 2500
x = torch.randn(B, Cig, T // 2, device="cuda"); gy = torch.randn(B, Cog, T, device="cuda")
fn = lambda: ops.conv_bwd_weight(GV.dense(x, 1), GV.dense(gy, 1), 3, pro=(2, None, None, 1), wino=4)
for _ in range(30): fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): fn()
e.record(); torch.cuda.synchronize()
print("dec k3 256->128 with the x2 prologue: %.3f ms" % (s.elapsed_time(e) / 20))
