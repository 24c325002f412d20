"""GPU-side screening of candidate seeds for TIE-FREE train fixtures (run by hand on the GPU box, not collected by pytest):

    python tests/screen_tie_free.py  kind:V:B:L:reg:masked:seed  ...        (kind = train | nefnet2)

oracle/tie_search.py (build container) ranks seeds by how far every ReLU / L1 argument of the fp64 oracle is from its switch in
units of its own fp32 error.  The arbiter is the HIP path: for each candidate this script runs the train step on BOTH conv paths
(split-fp16 forced / the product's own choice, as tests/test_model_gpu.py's `conv_path` fixture does), lets the fp64 oracle replay
the decisions the HIP path took (tests/decisions.py) and prints how many of them the oracle would have taken differently.  Seeds with
0 on both paths are the ones oracle/make_golden.py turns into fixtures from the REFERENCE's own run (`tie_free` = 1): on those,
tests compare HIP gradients with the reference's at the plain bars, no measured allowance."""
import json
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import test_model_gpu as T        # noqa: E402


def run(kind, V, B, L, reg, masked, seed):
    from electrocardio_panorama_amd import ops as o
    from electrocardio_panorama_amd.network import build_loss
    from oracle import hashweights as hw
    out = {}
    for path in ("h2", "auto"):
        saved = o._H2_MIN_WGS
        o._H2_MIN_WGS = 0 if path == "h2" else saved
        try:
            if kind == "train":
                m, outs, losses = T._train_once(V, B, L, seed, reg, masked)
                masks = hw.hashed_masks(V, B, L // 4) if masked else None
                _, _, dec, flat = T.oracle_replaying(m, outs, T.batch_t(B, V, L, seed, dev="cpu"), V, seed, masks=masks, reg=reg, dt=torch.float64)
            else:
                b = T.batch_t(B, V, L, seed, 3)
                cfg = T.make_cfg(V, reg)
                m = T.hashed_model2(V).train()
                m.dropout_p = 0.0
                m.keep_saved = True
                random.seed(seed)
                outs = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
                losses = build_loss(cfg)(outs[0], outs[1], outs[2], b["target_view"].unsqueeze(1), cfg)
                losses[0].backward()
                _, _, dec, flat = T.oracle_replaying(m, outs, {k: v.cpu() for k, v in b.items()}, V, seed, reg=reg, model2=True, fold=(B, V), dt=torch.float64)
            out[path] = {"flips": int(dec.total_flips()), "flat_vs_fp64": float(flat)}
        except AssertionError as exc:
            out[path] = {"error": str(exc)[:200]}
        finally:
            o._H2_MIN_WGS = saved
    return out


if __name__ == "__main__":
    res = {}
    for spec in sys.argv[1:]:
        kind, V, B, L, reg, masked, seed = spec.split(":")
        res[spec] = run(kind, int(V), int(B), int(L), reg, bool(int(masked)), int(seed))
        print(spec, json.dumps(res[spec]), flush=True)
    good = [s for s, r in res.items() if all(v.get("flips") == 0 for v in r.values())]
    print("tie-free on both conv paths:", good)
