"""Decoder forward + backward alone at the configs[1] shapes (three Standin passes of B=256, T=1250 latents), every launch
bracketed with HIP events on one stream: per-tag mean ms over ITERS rounds after WARM rounds.  The launches carry their real
prologues / epilogues (BatchNorm slot sums, bnb sums, upsampling), unlike tools/bench_conv.py.  A/B: NEF_LIB=<variant .so>."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

os.environ.setdefault("NEF_SIDE_STREAM", "0")
from electrocardio_panorama_amd import engine, ops  # noqa: E402

B, T = int(os.environ.get("B", 256)), int(os.environ.get("T", 1250))
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
P, Bf = {}, {}
cin = 256
for blk, cv, bn, cout in engine._DEC:
    P[f"{blk}.double_conv.{cv}.weight"] = torch.randn(cout, cin, 3, device=dev, generator=g) * (1.0 / (3 * cin) ** 0.5)
    P[f"{blk}.double_conv.{cv}.bias"] = torch.randn(cout, device=dev, generator=g) * 0.1
    pre = f"{blk}.double_conv.{bn}"
    P[pre + ".weight"] = torch.rand(cout, device=dev, generator=g) + 0.5
    P[pre + ".bias"] = torch.randn(cout, device=dev, generator=g) * 0.1
    Bf[pre + ".running_mean"] = torch.zeros(cout, device=dev)
    Bf[pre + ".running_var"] = torch.ones(cout, device=dev)
    Bf[pre + ".num_batches_tracked"] = torch.zeros((), device=dev, dtype=torch.long)
    cin = cout
P["decoder.4.weight"] = torch.randn(1, 64, 3, device=dev, generator=g) * 0.1
P["decoder.4.bias"] = torch.zeros(1, device=dev)
D2 = torch.randn(2 * B, 256, T, device=dev, generator=g).abs_()
g_out = torch.randn(3 * B, 1, 4 * T, device=dev, generator=g) * 1e-3


def once():
    out, dsv = engine.decoder_fwd(D2, P, Bf, 3, True, True, shared_B=B)
    grads = {}
    engine.decoder_bwd(dsv, g_out, P, grads)
    return out


for _ in range(int(os.environ.get("WARM", 3))):
    once()
torch.cuda.synchronize()
iters = int(os.environ.get("ITERS", 4))
ops.PROFILE = []
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    once()
e.record()
torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
tab = {}
for tag, a, b in prof:
    k = "/".join(str(x) for x in (tag[:2] if tag[0] == "hbm" else tag))
    tab.setdefault(k, []).append(a.elapsed_time(b))
tot = 0.0
for k, v in tab.items():
    per = len(v) // iters
    ms = sum(v) / iters
    tot += ms
    print(f"{k:44s} x{per}  {ms / per:7.3f} ms  (min {min(v):.3f})")
print(f"sum of bracketed launches {tot:.3f} ms; wall (events incl.) {s.elapsed_time(e) / iters:.3f} ms per fwd+bwd")
if os.environ.get("JSON"):
    json.dump({k: sum(v) / iters for k, v in tab.items()}, open(os.environ["JSON"], "w"))
