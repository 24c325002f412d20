"""Prints the fp16-vs-fp32 sweep error (rel-L2 on outputs and on pre-sigmoid logits) for a few shapes."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_model_gpu import batch_t, hashed_model
from test_pano_gpu import _logit3
from util import rel
from electrocardio_panorama_amd.network import build_model
import bench
for B, V, L, Q, init in [(2, 3, 512, 5, "hash"), (3, 1, 1000, 7, "hash"), (4, 3, 5000, 12, "hash"), (8, 1, 512, 36, "ref"), (4, 3, 5000, 12, "ref")]:
    if init == "hash":
        m = hashed_model(V).eval()
    else:
        torch.manual_seed(123); m = build_model(bench.make_cfg(V)).float().cuda().eval()
    b = batch_t(B, V, L, 11, Q)
    random.seed(0); ref = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], rest_theta=b["rest_theta"], phase="test")[3]
    m.panorama_dtype = "fp16"
    random.seed(0); got = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], rest_theta=b["rest_theta"], phase="test")[3]
    print(f"B={B} V={V} L={L} Q={Q} init={init}: out rel {rel(got, ref):.2e}  logit rel {rel(_logit3(got), _logit3(ref)):.2e}  "
          f"centered rel {rel(got - ref.mean(), ref - ref.mean()):.2e}  out std {float(ref.std()):.3e}")
