"""A/B of the two forms of the split-fp16 forward / backward-data convolution: csrc/conv_h2.hip (every wave stages, multiplies and
stores) against csrc/conv_h2p.hip (producer / consumer waves, persistent).  Same arithmetic in the same order, so every output
must be BIT-IDENTICAL; prints the time of both per shape.  usage: h2p_check.py [name filter] (ITERS, WARM, CHECK=0 env)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from electrocardio_panorama_amd import _lib, ops
from electrocardio_panorama_amd.ops import GV

L = _lib.load()
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(5)


def rnd(*s, scale=1.0):
    return (torch.rand(*s, device=dev, generator=g) * 2 - 1) * scale


# (name, K, G, Cig, Cog, B, T_out, pro_mode, extras)
CASES = [
    ("enc k7 128->128 g3", 7, 3, 128, 128, 256, 1250, 0, "relu,drop"),
    ("enc k7 bwd", 7, 3, 128, 128, 256, 1250, 0, "gate"),
    ("w_conv k3 g3", 3, 3, 128, 128, 256, 1250, 0, "relu,drop,sc"),
    ("w_conv k3 g3 res", 3, 3, 128, 128, 256, 1250, 0, "relu,res"),
    ("z1 k3 64->128 g3", 3, 3, 64, 128, 256, 1250, 0, "relu"),
    ("k1 64->128 g3", 1, 3, 64, 128, 256, 1250, 0, "bias"),
    ("dec1 k3 128->128 g2 up", 3, 2, 128, 128, 512, 2500, 2, "bias"),
    ("dec2 k3 128->128 aff", 3, 1, 128, 128, 768, 2500, 1, "bias,stats"),
    ("dec3 k3 128->64 aff+up", 3, 1, 128, 64, 768, 5000, 3, "bias,stats"),
    ("dec4 k3 64->64 aff", 3, 1, 64, 64, 768, 5000, 1, "bias,stats"),
    ("dec bwd k3 64->64 bnb", 3, 1, 64, 64, 768, 5000, 0, "bnb"),
    ("dec bwd k3 64->128 bnbup", 3, 1, 64, 128, 768, 5000, 0, "bnbup"),
    ("dec bwd k3 128->128 bnb", 3, 1, 128, 128, 768, 2500, 0, "bnb"),
    ("loc k3 64->64 T=5000", 3, 1, 64, 64, 768, 5000, 0, "relu"),
    ("loc k3 64->64 T=256 B=15000", 3, 1, 64, 64, 15000, 256, 0, "relu"),
    ("loc k3 64->64 T=1024 B=3750", 3, 1, 64, 64, 3750, 1024, 0, "relu"),
    ("ragged k3 B=13 T=1250", 3, 3, 64, 64, 13, 1250, 0, "relu,res"),
    ("ragged k7 B=9 T=386", 7, 2, 48, 64, 9, 386, 0, "relu"),
    ("small up B=5 T=512", 3, 1, 64, 64, 5, 512, 3, "bias,stats"),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
forms = tuple(int(f) for f in os.environ.get("FORMS", "0,1").split(","))      # FORMS=0: the default form only (profiling runs)
with_wgrad = os.environ.get("WG", "0") == "1"      # WG=1: also launch the weight gradient of the case (same prologue) once per iteration
iters, warm = int(os.environ.get("ITERS", 10)), int(os.environ.get("WARM", 20))
check = os.environ.get("CHECK", "1") == "1"
bad = 0
for name, K, G, Cig, Cog, B, T, pm, extra in CASES:
    if only and only not in name:
        continue
    ex = set(extra.split(","))
    Tin = T // 2 if pm & 2 else T
    x = rnd(B, G * Cig, Tin)
    w = rnd(G * Cog, Cig, K, scale=0.05)
    wp = ops.pack_weight(w, G, T=T)
    assert getattr(wp, "nef_wino", 0) == 3, name
    kw = dict(x_scale=16.0)
    if "relu" in ex:
        kw["relu"] = True
    if "bias" in ex:
        kw["bias"] = rnd(G * Cog)
    if "drop" in ex:
        kw.update(drop_p=0.2, drop_scale=1.25, seed=77)
    if "sc" in ex:
        kw["in_scale"] = (rnd(B, G * Cig) + 1.5, G * Cig, Cig)
    if "res" in ex:
        kw["res"] = GV.dense(rnd(B, G * Cog, T), G)
    if "gate" in ex:
        kw.update(gate=GV.dense(rnd(B, G * Cog, T), G), gate_scale=1.25, role="conv_bwd_data")
    P = 3 if B % 3 == 0 else 1
    if pm:
        kw["pro"] = (pm, rnd(P, G * Cig) + 0.5, rnd(P, G * Cig) * 0.3, B // P)
    if "bnb" in ex or "bnbup" in ex:
        up = "bnbup" in ex
        bx = rnd(B, G * Cog, T // 2 if up else T)
        kw["role"] = "conv_bwd_data"

    def run():
        k2 = dict(kw)
        if "stats" in ex:
            k2["stats"] = ops.conv_stats_buffer(wp, B, G, Cog, T, dev)
            k2["stats"][0].zero_()
        if "bnb" in ex or "bnbup" in ex:
            sl = ops.conv_stats_buffer(wp, B, G, Cog, T, dev)
            sl[0].zero_()
            k2["bnb"] = (bx, mean, invstd, ba, bb, B // P, sl, int("bnbup" in ex))
        y = ops.conv(GV.dense(x, G), wp, Cog, K, **k2)
        return y, (k2.get("stats") or k2.get("bnb", [None] * 7)[6] or [None])[0]

    if "bnb" in ex or "bnbup" in ex:
        mean, invstd, ba, bb = rnd(P, G * Cog), rnd(P, G * Cog) + 1.5, rnd(P, G * Cog) + 0.5, rnd(P, G * Cog) * 0.3
    res = {}
    if with_wgrad and K in (3, 7) and not (ex & {"bnb", "bnbup", "gate"}):
        gyw = rnd(B, G * Cog, T)
        run_conv = run

        def run():
            out = run_conv()
            ops.conv_bwd_weight(GV.dense(x, G), GV.dense(gyw, G), K, pro=kw.get("pro"), in_scale=kw.get("in_scale"), h2=True,
                                x_scale=16.0, gy_scale=16.0)
            return out
    for form in forms:
        L.nef_set_option(_lib.OPT_H2_FORM, form)
        y, st = run()
        torch.cuda.synchronize()
        for _ in range(warm):
            run()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            run()
        e.record()
        torch.cuda.synchronize()
        res[form] = (y, st, s.elapsed_time(e) / iters)
    msg = ""
    if len(forms) < 2:
        print(f"{name:28s} form {forms[0]} {res[forms[0]][2]:7.3f} ms", flush=True)
        continue
    if check:
        same = torch.equal(res[0][0], res[1][0])
        sst = res[0][1] is None or torch.equal(res[0][1], res[1][1])
        if not (same and sst):
            bad += 1
            d = (res[0][0] - res[1][0]).abs()
            msg = f"  MISMATCH y: {int((d > 0).sum())} of {d.numel()} differ, max {float(d.max()):.3e}, slots equal: {sst}"
            idx = (d > 0).nonzero()
            if len(idx):
                msg += f" first {idx[0].tolist()} last {idx[-1].tolist()}"
        else:
            msg = "  bit-identical" + (" (+slots)" if res[0][1] is not None else "")
    print(f"{name:28s} old {res[0][2]:7.3f} ms   p/c {res[1][2]:7.3f} ms   x{res[0][2] / res[1][2]:5.2f}{msg}", flush=True)
L.nef_set_option(_lib.OPT_H2_FORM, 1)
print("FAILED" if bad else "all equal")
sys.exit(1 if bad else 0)
